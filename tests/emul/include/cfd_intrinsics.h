// Emulator twin of cfdbench_amd/csrc/cfd_intrinsics.h (same function names, host semantics).
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <cstring>
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: a = A[i=lane&15][k=lane>>4], b = B[k=lane>>4][j=lane&15],
// c/d[r] = C[row=(lane>>4)*4+r][col=lane&15]; k-ordered fmaf chain (cdna_hip_programming.md section 3).
inline f32x4 cfd_mfma16x16x4(float a, float b, f32x4 c) {
    auto& w = cfd_emul::wave();
    const int l = cfd_emul::lane();
    w.fa[l] = a;
    w.fb[l] = b;
    cfd_emul::wave_sync();
    f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[k * 16 + row], w.fb[k * 16 + col], acc);
        d[r] = acc;
    }
    cfd_emul::wave_sync();
    return d;
}

// v_mfma_f32_16x16x32_bf16: a = A[i=lane&15][k=8*(lane>>4)+v], b = B[k=8*(lane>>4)+v][j=lane&15], v = 0..7;
// c/d as the 16x16x4 form.  bf16 products are exact in fp32; the hardware's internal summation order is not
// architected, so the emulator sums in k order in fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
inline f32x4 cfd_mfma16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    auto& w = cfd_emul::wave();
    const int l = cfd_emul::lane();
    for (int v = 0; v < 8; ++v) { w.fa8[l][v] = (float)a[v]; w.fb8[l][v] = (float)b[v]; }
    cfd_emul::wave_sync();
    f32x4 d = c;
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) acc += w.fa8[(k >> 3) * 16 + row][k & 7] * w.fb8[(k >> 3) * 16 + col][k & 7];
        d[r] = acc;
    }
    cfd_emul::wave_sync();
    return d;
}

inline float cfd_shfl_xor(float v, int mask) {
    auto& w = cfd_emul::wave();
    const int l = cfd_emul::lane();
    w.fa[l] = v;
    cfd_emul::wave_sync();
    float r = w.fa[(l ^ mask) & 63];
    cfd_emul::wave_sync();
    return r;
}

inline float cfd_row_sum4(float v) {  // same association as the device form: (r0 + r1) + (r2 + r3) per lane & 15
    const float s = v + cfd_shfl_xor(v, 16);
    return s + cfd_shfl_xor(s, 32);
}

inline float cfd_wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += cfd_shfl_xor(v, m);
    return v;
}

inline void cfd_wave_lds_sync() { cfd_emul::wave_sync(); }

// global -> LDS copy of 16 bytes per lane: destination = wave-uniform base + 16 * lane (global_load_lds_dwordx4)
inline void cfd_glds16(const void* g_lane, void* lds_wave_base) { std::memcpy((char*)lds_wave_base + 16 * cfd_emul::lane(), g_lane, 16); }

inline int cfd_opaque(int x) { return x; }
inline float cfd_opaque_f(float x) { return x; }
inline f32x4 cfd_opaque_f4(f32x4 x) { return x; }
inline unsigned cfd_pack_hi16(unsigned hi, unsigned lo) { return (hi & 0xffff0000u) | (lo >> 16); }

inline void cfd_sched_fence() {}

inline int cfd_uniform(int x) { return x; }

// dynamic LDS: the emulator's launch() sizes one buffer per launch
#define CFD_DYN_SHARED(T, name) T* name = reinterpret_cast<T*>(cfd_emul::g_dyn_shared.data())

inline float cfd_erff(float x) { return erff(x); }
inline float cfd_expf(float x) { return expf(x); }
inline float cfd_rcpf(float x) { return 1.0f / x; }
inline unsigned cfd_umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline float cfd_exp2f(float x) { return exp2f(x); }

// raw buffer resource: the hardware drops lanes whose voffset is out of range (loads return 0); a LIVE lane that reaches beyond the
// resource is a kernel bug the hardware would not report -- the emulator aborts on it
struct CfdBuf { char* base; unsigned bytes; };
#define CFD_BUF_OOB 0x80000000u
inline CfdBuf cfd_buf(const void* base, unsigned bytes) { return CfdBuf{const_cast<char*>(static_cast<const char*>(base)), bytes}; }
inline void cfd_buf_check(const CfdBuf& b, unsigned voff, unsigned soff) {
    if ((unsigned long long)voff + soff + 4ull > b.bytes) {
        fprintf(stderr, "emulator: buffer access at voffset %u + soffset %u beyond a %u-byte resource\n", voff, soff, b.bytes);
        abort();
    }
}
inline float cfd_buf_ld(CfdBuf b, unsigned voff, unsigned soff) {
    if (voff >= b.bytes) return 0.f;
    cfd_buf_check(b, voff, soff);
    float v;
    memcpy(&v, b.base + voff + soff, 4);
    return v;
}
inline f32x4 cfd_buf_ld4(CfdBuf b, unsigned voff, unsigned soff) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (voff >= b.bytes) return v;
    cfd_buf_check(b, voff + 12, soff);  // a live access lies inside the resource as a whole
    memcpy(&v, b.base + voff + soff, 16);
    return v;
}
inline void cfd_buf_st(CfdBuf b, unsigned voff, unsigned soff, float v) {
    if (voff >= b.bytes) return;
    cfd_buf_check(b, voff, soff);
    memcpy(b.base + voff + soff, &v, 4);
}
inline bool cfd_wave_any(bool p) {
    auto& w = cfd_emul::wave();
    const int l = cfd_emul::lane();
    w.fa[l] = p ? 1.f : 0.f;
    cfd_emul::wave_sync();
    bool any = false;
    for (int i = 0; i < 64; ++i) any = any || w.fa[i] != 0.f;
    cfd_emul::wave_sync();
    return any;
}

typedef float cfd_f2 __attribute__((ext_vector_type(2)));
inline cfd_f2 cfd_fma2(cfd_f2 a, cfd_f2 b, cfd_f2 c) { return __builtin_elementwise_fma(a, b, c); }
inline cfd_f2 cfd_abs2(cfd_f2 a) { return __builtin_elementwise_abs(a); }
inline float cfd_hsum2(cfd_f2 v) { return v.x + v.y; }

// complex multiply-accumulate, same rounding sequence as the device form (two fused steps per component)
inline cfd_f2 cfd_cmla(cfd_f2 acc, cfd_f2 x, cfd_f2 w) {
    cfd_f2 r = {fmaf(x.x, w.x, acc.x), fmaf(x.x, w.y, acc.y)};
    return cfd_f2{fmaf(-x.y, w.y, r.x), fmaf(x.y, w.x, r.y)};
}
inline cfd_f2 cfd_cmla_conj(cfd_f2 acc, cfd_f2 x, cfd_f2 w) {
    cfd_f2 r = {fmaf(x.x, w.x, acc.x), fmaf(x.x, w.y, acc.y)};
    return cfd_f2{fmaf(x.y, w.y, r.x), fmaf(-x.y, w.x, r.y)};
}
