// Shared host/device helpers for the cfdbench_amd HIP sources.
#pragma once
#include <cfd_intrinsics.h>
#include <hip/hip_runtime.h>

#include "../../include/cfdbench_amd.h"

void cfd_set_error(const char* fmt, ...);

// dispatch overrides (tune.cpp): environment read once per process, cfd_tune_set() afterwards; -1 = built-in choice
enum { CFD_TUNE_MIX_NWV = 0, CFD_TUNE_WGRAD_WG, CFD_TUNE_FUSED_VARIANT, CFD_TUNE_BLOCK_FUSE, CFD_TUNE_GENERAL_B3, CFD_TUNE_HEAD_BLOCKS, CFD_TUNE_EXACT_FP32, CFD_TUNE_CONV6_GRID, CFD_TUNE_CONV6_WGRAD_MUL, CFD_TUNE_CONVT_MFMA, CFD_TUNE_CONV1_MFMA, CFD_TUNE_SIDE_STREAM, CFD_TUNE_ACT_PIECES, CFD_TUNE_BLOCK_GEN,
    CFD_TUNE_GEMM_TILE, CFD_TUNE_GEMM_SPLITS, CFD_TUNE_BLOCK_WIDE, CFD_TUNE_HEAD_WAVES, CFD_TUNE_STEM_FUSE, CFD_TUNE_MODE_MFMA, CFD_TUNE_MODE_BC, CFD_TUNE_STEM_DFT, CFD_TUNE_GEMM_B3, CFD_TUNE_COUNT };
int cfd_tune_get(int which);

#define CFD_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            cfd_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define CFD_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            cfd_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e_));  \
            return CFD_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

#define CFD_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != CFD_OK) return rc_; \
    } while (0)

// side.cpp: fork / join onto the library's per-device side stream (independent kernels of one call beside each other)
hipStream_t cfd_side_fork(hipStream_t main, int use);
int cfd_side_join(hipStream_t main, hipStream_t side);
bool cfd_prof_active();  // prof.cpp: the per-kernel event profiler is recording

// Per-kernel event timing (prof.cpp); a no-op unless cfd_prof_begin() was called.
struct CfdProfScope {
    hipStream_t st;
    int idx;
    CfdProfScope(const char* name, hipStream_t s, double bytes = 0.0, double flops = 0.0);
    ~CfdProfScope();
};
#define CFD_PROF_CAT2(a, b) a##b
#define CFD_PROF_CAT(a, b) CFD_PROF_CAT2(a, b)
#define CFD_PROF(name, st) CfdProfScope CFD_PROF_CAT(cfd_prof_scope_, __LINE__)(name, st)
// the same with the launch's ALGORITHMIC HBM bytes (what must cross HBM once: inputs read + outputs written, DESIGN.md
// section 4) and flops, so that bench.py's roofline of a kernel whose launches differ in shape is the sum over its launches
#define CFD_PROF_W(name, st, bytes, flops) \
    CfdProfScope CFD_PROF_CAT(cfd_prof_scope_, __LINE__)(name, st, (double)(bytes), (double)(flops))

#define CFD_KB_TMAX 5  // row tiles of 16 the 64-column kernels walk at most (H <= 80)
// Operator tables of one (H,W,m1,m2); all d_* pointers are device memory, fragment-major ([step][lane]).
struct cfd_plan {
    int H, W, m1, m2;
    int NJ;  // columns per lane = ceil(W/16)
    int KX;  // forward stage-1 k-steps over folded rows: ceil((H/2+1)/4)
    int T;   // inverse x tiles: ceil(H/16)
    int SA;  // inverse stage-A k-steps: 4 (cos, kappa 0..15) + ceil(m1/4) (sin, kappa 1..m1)
    int SB;  // inverse stage-B k-steps: ceil(2*m2/4)
    int n_fwd, n_inv;
    float* d_fwd;   // t1c[KX][64] | t1s[KX][64] | t2c[4*NJ][64] | t2s[4*NJ][64]
    float* d_inv;   // ta[T][SA][64] | tb[SB][NJ][64]
    void* d_fwd_b3; // split-bf16 forward tables for the K = 32 MFMA (NULL unless H == 64, W == 64): 16-byte vectors
                    // [table][hi|lo][64 lanes], tables T1C, T1S, T1N (stage 1: rows 4v+q | Nyquist row), T2C0, T2C1,
                    // T2S0, T2S1 (stage 2: columns 4(4q+r)+2h+jj, v = 4jj+r)
    void* d_inv_b3; // split-bf16 form of the inverse tables for the K = 32 MFMA with the column map y = 4 n + j (NULL unless T <=
                    // CFD_KB_TMAX, SA <= 8, SB <= 8, 64 <= W <= 68): 16-byte vectors ta3[T][piece][64 lanes] | tb3[4][piece][64
                    // lanes], element v of lane vector = table value of k-step v (zero beyond SA / SB, beyond H / W)
    int E;          // tail columns W - 64 of the 64-column kernels (0 when W == 64 or d_inv_b3 == NULL)
    float* d_tail;  // [4][32] stage-B factors of the tail columns (fp32; NULL when d_inv_b3 == NULL)
    // General-width split-bf16 tables (NULL unless H <= 70 and W <= 80): columns are dealt to lanes as y = 16 j + n (tile j of
    // NJG = ceil(W/16), lane n) instead of the 64-wide kernels' y = 4 n + j, so any W works with coalesced scalar accesses.
    void* d_fwd_g;  // 16-byte vectors [table][hi|lo][64 lanes]: T1C0, T1S0 (folded rows 4v+q), T1C1, T1S1 (rows 32+4v+q), then
                    // T2C[h], T2S[h] for the NHG = ceil(NJG/2) column halves (k-slot (q, v = 4jj+r) = column 16(2h+jj)+4q+r)
    void* d_inv_g;  // ta3[T][hi|lo][64] | tb3[NJG][hi|lo][64] with y = 16 j + n
    int NJG, NHG, n_fwd_gv, n_inv_gv;  // vector counts of the two tables
    float* d_clhw;  // [m2]  c_l / (H*W)
    float* d_gx;    // [H]  np.linspace(0,1,H) as float32   (fno2d.py:251)
    float* d_gy;    // [W]
};

static inline size_t cfd_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Storage type of the activations between kernels: float everywhere, or __bf16 on the bf16-storage inference path
// (BASELINE configs[4]; fp32 arithmetic and accumulation either way).  CFD_DT_* select it at the internal entry points.
#define CFD_DT_F32 0
#define CFD_DT_BF16 1
template <typename TA>
__device__ __forceinline__ float cfd_ld(const TA* p) { return (float)*p; }
template <typename TA>
__device__ __forceinline__ void cfd_st(TA* p, float v) { *p = (TA)v; }
static inline size_t cfd_dt_size(int dt) { return dt == CFD_DT_BF16 ? 2 : 4; }
// four consecutive stored activations (16-byte aligned fp32 / 8-byte aligned bf16) as one vector load
__device__ __forceinline__ float4 cfd_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 cfd_ld4(const __bf16* p) {
    typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
    const bf16x4_ t = *reinterpret_cast<const bf16x4_*>(p);
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
}

// Four consecutive floats from / to an address that is only 4-byte aligned, as ONE 16-byte access (global_load / store_dwordx4; gfx950
// runs them at 94 % of the aligned rate: tools/exp/unaligned_x4.hip).  The planes of the 66 x 65 grids (4290 floats) start on 8-byte
// boundaries and their rows on 4-byte ones; rounds 1-3 read them with 8- or 4-byte accesses.
struct __attribute__((packed, aligned(4))) cfd_f4u { float x, y, z, w; };
__device__ __forceinline__ float4 cfd_ld4u(const float* p) {
    const cfd_f4u v = *reinterpret_cast<const cfd_f4u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void cfd_st4u(float* p, float4 v) { *reinterpret_cast<cfd_f4u*>(p) = cfd_f4u{v.x, v.y, v.z, v.w}; }

// Internal (not exported) forms of the forward-path entry points with the activation storage type as an argument; dt =
// CFD_DT_F32 is exactly the public function.  x / out / aprev / a point at activations of that type.  With bf16 storage the
// 1x1 conv writes its (unrounded) result as fp32 and the inverse transform takes that fp32 addend, so a stored pre-activation
// is rounded exactly once.
int cfd_int_spectral_dft(const cfd_plan* p, const void* x, float* xh, int nimg, int act_in, int dt, void* stream);
bool cfd_int_dft_stem_ok(const cfd_plan* p, int B, int in_chan, int P, int C, const void* inputs, const void* mask, const void* a0);
int cfd_int_spectral_dft_stem(const cfd_plan* p, const float* inputs, const float* mask, const float* cp, const float* w, const float* bias,
                              float* a0, float* xh, int B, int P, int C, void* stream);
int cfd_int_spectral_idft(const cfd_plan* p, const float* z, const void* addend, const void* aprev, void* out, int nimg, int epi,
                          int dt, void* stream);
int cfd_int_spectral_idft_grad(const cfd_plan* p, const float* z, const float* addend, const void* aprev, float* out, int nimg,
                               int dt, void* stream);
int cfd_int_chanmix(const void* in, const float* w, const float* bias, void* out, int B, int Ci, int Co, int HW, int act_in,
                    int transpose, int dt, void* stream);
int cfd_int_fno_stem_fwd(const cfd_plan* p, const float* inputs, const float* mask, const float* case_params, const float* w,
                         const float* bias, void* out, int B, int in_chan, int P, int C, int dt, void* stream);
int cfd_int_fno_head_fwd(const void* a, const float* mask, const float* label, const float* w1, const float* b1, const float* w2,
                         const float* b2, float* preds, float* sums, void* ws, int B, int C, int Hd, int Co, int HW, int act_in,
                         int dt, void* stream);

int cfd_int_fno_head_train(const void* a, const float* mask, const float* label, const float* coef, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* preds, float* sums, float* ga, float* gw1, float* gb1,
                           float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co, int HW, int act_in, int dt, void* stream);

#define CFD_SQRT1_2 0.70710678118654752440f
#define CFD_INV_SQRT_2PI 0.39894228040143267794f

// nn.GELU() exact-erf form (fno2d.py:147) and its derivative Phi(x) + x phi(x).
// Phi(x) = (1 + erf(x/sqrt 2))/2 with erf from Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7, i.e. fp32 round-off
// class: nMSE vs the fp64 GELU on N(0,1) inputs 8e-15, ATen's own fp32 erff-based GELU 4e-15).  GELU is recomputed
// on load by every consumer of an activation and is the VALU bottleneck of the head kernels, so it is evaluated on
// PAIRS of values with packed fp32 math: 12 packed ops + 2 rcp + 2 exp2 per PAIR instead of ~36 scalar VALU
// instructions per value of libdevice erff.  The same exp(-x^2/2) serves erf and the density phi (derivative = 2 more ops).
// hc = erfc(|x| / sqrt 2) / 2 = Phi(-|x|) and e = exp(-x^2 / 2): the A&S polynomial with the 1/2 folded into its coefficients
__device__ __forceinline__ void cfd_gelu_hc2(cfd_f2 x, cfd_f2& hc, cfd_f2& e) {
    const cfd_f2 d = cfd_fma2(cfd_abs2(x), (cfd_f2)(0.3275911f * CFD_SQRT1_2), (cfd_f2)(1.0f));
    cfd_f2 t;
    t.x = cfd_rcpf(d.x);
    t.y = cfd_rcpf(d.y);
    const cfd_f2 arg = (x * x) * (cfd_f2)(-0.72134752044448170368f);  // -x^2/2 * log2(e)
    e.x = cfd_exp2f(arg.x);  // exp(-x^2/2)
    e.y = cfd_exp2f(arg.y);
    cfd_f2 p = cfd_fma2((cfd_f2)(0.5f * 1.061405429f), t, (cfd_f2)(0.5f * -1.453152027f));
    p = cfd_fma2(p, t, (cfd_f2)(0.5f * 1.421413741f));
    p = cfd_fma2(p, t, (cfd_f2)(0.5f * -0.284496736f));
    p = cfd_fma2(p, t, (cfd_f2)(0.5f * 0.254829592f));
    hc = (p * t) * e;
}
__device__ __forceinline__ void cfd_gelu_terms2(cfd_f2 x, cfd_f2& Phi, cfd_f2& e) {
    cfd_f2 hc;
    cfd_gelu_hc2(x, hc, e);
    cfd_f2 s = (cfd_f2)(0.5f) - hc;  // Phi(|x|) - 1/2 >= 0
    s.x = copysignf(s.x, x.x);
    s.y = copysignf(s.y, x.y);
    Phi = (cfd_f2)(0.5f) + s;
}
// gelu(x) = x Phi(x) = x/2 + |x| (1/2 - hc): no sign fix-up at all
__device__ __forceinline__ cfd_f2 cfd_gelu2(cfd_f2 x) {
    cfd_f2 hc, e;
    cfd_gelu_hc2(x, hc, e);
    return cfd_fma2(cfd_abs2(x), (cfd_f2)(0.5f) - hc, x * (cfd_f2)(0.5f));
}
__device__ __forceinline__ cfd_f2 cfd_gelu_grad2(cfd_f2 x) {
    cfd_f2 Phi, e;
    cfd_gelu_terms2(x, Phi, e);
    return cfd_fma2(x * (cfd_f2)(CFD_INV_SQRT_2PI), e, Phi);
}
// scalar forms: the same operation sequence on one value
__device__ __forceinline__ void cfd_gelu_terms(float x, float& Phi, float& e) {
    cfd_f2 P2, e2;
    cfd_gelu_terms2((cfd_f2)(x), P2, e2);
    Phi = P2.x;
    e = e2.x;
}
__device__ __forceinline__ float cfd_gelu(float x) {
    float Phi, e;
    cfd_gelu_terms(x, Phi, e);
    return x * Phi;
}
// in-place on four consecutive values (a float4 load)
__device__ __forceinline__ void cfd_gelu4(float& a, float& b, float& c, float& d) {
    const cfd_f2 u = cfd_gelu2(cfd_f2{a, b}), v = cfd_gelu2(cfd_f2{c, d});
    a = u.x; b = u.y; c = v.x; d = v.y;
}
__device__ __forceinline__ void cfd_gelu_grad4(float& a, float& b, float& c, float& d) {
    const cfd_f2 u = cfd_gelu_grad2(cfd_f2{a, b}), v = cfd_gelu_grad2(cfd_f2{c, d});
    a = u.x; b = u.y; c = v.x; d = v.y;
}
__device__ __forceinline__ float cfd_gelu_grad(float x) {
    float Phi, e;
    cfd_gelu_terms(x, Phi, e);
    return fmaf(x * CFD_INV_SQRT_2PI, e, Phi);
}

// ---- split-bf16 operands ("bf16x3") ---------------------------------------------------------------------------
// fp32 MFMA runs at the VALU's fp32 rate on gfx950 (157 TF, and it does not overlap with VALU work), the bf16 form at
// 16x that.  A value x is carried as hi = bf16(x), lo = bf16(x - hi) (16 significant bits together) and a product
// a*b as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi in three bf16 MFMAs with fp32 accumulation: relative error <= ~2^-16 per
// product (the dropped a_lo*b_lo term and the rounding of lo), i.e. an nMSE of ~1e-10 against exact fp32 -- five
// orders inside the 1e-5 parity budget -- at < 1/5 of the fp32 MFMA cost.
struct CfdSplit8 {
    bf16x8 hi, lo;
};
// FIXED operands -- the twiddle tables of the transforms -- are carried in CFD_TW = 3 bf16 pieces (hi + lo + lo2 = the fp32 value
// exactly): the rounding of a fixed operand is the same in every step of a rollout, i.e. a coherent perturbation of the
// operator that accumulates (200 steps through the spectral branch: nMSE 7e-7 with two pieces, 1e-9 with three --
// tests/test_gpu_fullsize.py), while the activations' rounding averages out.  A product then takes FOUR bf16 MFMAs:
// x_lo t_hi + x_hi t_lo + x_hi t_lo2 + x_hi t_hi.  Tables are stored piece-major: [table][piece][64 lanes] 16-byte vectors.
#define CFD_TW 3
// hi = bf16(x), lo = bf16(x - hi), two values per v_cvt_pk_bf16_f32: the float value of hi is its bit pattern shifted
// into the upper half, so a pair costs 2 conversions + 1 shift + 1 mask + 2 subtractions (written element by element the
// compiler converted every value twice, once for the packed operand and once for 16-bit stores: 40 instead of 24 VALU
// instructions per 8 values).
typedef __bf16 cfd_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned cfd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cfd_pack_bf16(float a, float b) {
    const cfd_bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ CfdSplit8 cfd_split8(const float (&x)[8]) {
    cfd_u32x4 h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned hk = cfd_pack_bf16(x[2 * k], x[2 * k + 1]);
        const float h0 = __builtin_bit_cast(float, hk << 16), h1 = __builtin_bit_cast(float, hk & 0xffff0000u);
        // opaque: a later 16-bit element extraction reads these packed words instead of re-converting the float
        h[k] = (unsigned)cfd_opaque((int)hk);
        l[k] = (unsigned)cfd_opaque((int)cfd_pack_bf16(x[2 * k] - h0, x[2 * k + 1] - h1));
    }
    CfdSplit8 s;
    s.hi = __builtin_bit_cast(bf16x8, h);
    s.lo = __builtin_bit_cast(bf16x8, l);
    return s;
}
// Three pieces, p[0] + p[1] + p[2] == x exactly (8 + 8 + 8 significant bits; both residuals are exact in fp32): the operand format
// of the convolution stack (conv6.hip), whose ReLU / max-pool networks need fp32-exact products (a two-piece
// product moves pre-activations across the ReLU kink).
struct CfdSplit8x3 {
    bf16x8 p[3];
};
__device__ __forceinline__ CfdSplit8x3 cfd_split8x3(const float (&x)[8]) {
    // Truncating pieces (hi = the upper 16 bits of x, lo = the upper 16 bits of x - hi, lo2 = x - hi - lo, 8 bits left: exact); a
    // PAIR of values costs 4 masks + 2 packed subtractions + 3 byte permutes = 9 VALU instructions (the round-to-nearest form
    // with v_cvt_pk_bf16_f32 took 13; the split is what the convolution kernels spend most of their VALU time on).
    cfd_u32x4 h, l, m;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // (scalars are bit-cast, never vector elements: __builtin_bit_cast of `v.y` read element 0 with this clang)
        const float x0 = x[2 * k], x1 = x[2 * k + 1];
        const unsigned h0 = __builtin_bit_cast(unsigned, x0) & 0xffff0000u, h1 = __builtin_bit_cast(unsigned, x1) & 0xffff0000u;
        const cfd_f2 xv = {x0, x1}, hv = {__builtin_bit_cast(float, h0), __builtin_bit_cast(float, h1)};
        const cfd_f2 r = xv - hv;
        const float r0 = r.x, r1 = r.y;
        const unsigned l0 = __builtin_bit_cast(unsigned, r0) & 0xffff0000u, l1 = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
        const cfd_f2 lv = {__builtin_bit_cast(float, l0), __builtin_bit_cast(float, l1)};
        const cfd_f2 t = r - lv;
        const float t0 = t.x, t1 = t.y;
        h[k] = cfd_pack_hi16(h1, h0);
        l[k] = cfd_pack_hi16(l1, l0);
        m[k] = cfd_pack_hi16(__builtin_bit_cast(unsigned, t1), __builtin_bit_cast(unsigned, t0));
    }
    CfdSplit8x3 s;
    s.p[0] = __builtin_bit_cast(bf16x8, h);
    s.p[1] = __builtin_bit_cast(bf16x8, l);
    s.p[2] = __builtin_bit_cast(bf16x8, m);
    return s;
}
// Workgroup copy of n elements (16-byte vectors or floats) from global memory into LDS with up to EIGHT loads per thread in flight before
// the first store.  Written as `for (i = tid; i < n; i += blockDim.x) dst[i] = src[i]` with a run-time n, hipcc emits load / s_waitcnt
// vmcnt(0) / ds_write per trip: one exposed L2 round trip per 4 KB (256 threads) -- 6-8 of them, 4-5 us, in front of every workgroup of
// the general-grid transforms and of k_idft64 (round 6; found in the ISA of k_idft_g).
template <typename V>
__device__ __forceinline__ void cfd_stage_lds(V* dst, const V* __restrict__ src, int n) {
    const int step = (int)blockDim.x;
    for (int i0 = (int)threadIdx.x; i0 < n; i0 += 8 * step) {
        V r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * step;
            r[k] = src[i < n ? i : n - 1];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * step;
            if (i < n) dst[i] = r[k];
        }
    }
}
// counter-based hash of (seed, element index): the keep mask of nn.Dropout (conv.hip k_dropout, pointwise.hip k_dropout_gelu)
__device__ __forceinline__ unsigned cfd_hash32(unsigned long long v) {
    v ^= v >> 33; v *= 0xff51afd7ed558ccdULL;
    v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL;
    v ^= v >> 33;
    return (unsigned)v;
}
// base[byte offset]: a wave-uniform base pointer plus a 32-bit per-lane BYTE offset compiles to the saddr + voffset addressing
// form (no 64-bit address arithmetic in the VALU); the caller guarantees the tensor is smaller than 4 GB
__device__ __forceinline__ float cfd_ldg_off(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void cfd_stg_off(float* base, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// D += A*B for two three-piece operands: the six products down to 2^-24 relative (a0 b0, a0 b1, a1 b0, a1 b1, a0 b2, a2 b0);
// the dropped ones are <= 2^-32 of the product, below the rounding of the fp32 accumulation itself.  Small terms first.
__device__ __forceinline__ f32x4 cfd_mfma_bf16x6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
    c = cfd_mfma16x16x32_bf16(a[2], b[0], c);
    c = cfd_mfma16x16x32_bf16(a[0], b[2], c);
    c = cfd_mfma16x16x32_bf16(a[1], b[1], c);
    c = cfd_mfma16x16x32_bf16(a[1], b[0], c);
    c = cfd_mfma16x16x32_bf16(a[0], b[1], c);
    return cfd_mfma16x16x32_bf16(a[0], b[0], c);
}
// The same for N independent accumulators sharing the A operand, product-major: consecutive MFMAs never depend on each other (a
// chain of six dependent 16x16x32 MFMAs issues at about half rate: each waits for the previous result).
template <int N>
__device__ __forceinline__ void cfd_mfma_bf16x6_n(const bf16x8 (&a)[3], const bf16x8 (&b)[N][3], f32x4 (&c)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[2], b[i][0], c[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[0], b[i][2], c[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[1], b[i][1], c[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[1], b[i][0], c[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[0], b[i][1], c[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) c[i] = cfd_mfma16x16x32_bf16(a[0], b[i][0], c[i]);
}
// ---- activation operands of the FNO contractions (transforms, 1x1 conv, 1x1 weight gradient) ---------------------------------
// AP = pieces the ACTIVATION operand is carried in (template parameter of the kernels, chosen per launch by the act_pieces knob):
//   AP = 3 (the default since round 5, tune.cpp: cfd_act_pieces()): three truncating pieces (cfd_split8x3: p[0] + p[1] + p[2] == x
//           exactly) and the six MFMAs of cfd_mfma_bf16x6 -- every term down to 2^-24 of the product, i.e. fp32-exact class like the
//           convolution stack: measured nMSE 6e-15 .. 5e-14 on every transform / FnoBlock / 1x1 kernel (profiles/r04d_err_act3.json).
//           Against two pieces it costs 3 more VALU instructions per pair of values and two more MFMAs per product: +8 % on the step.
//   AP = 2 (`cfd_tune_set("act_pieces", 2)`, the default of rounds 1-4, bench.py's `split2` leg): hi = bf16(x), lo = bf16(x - hi); a
//           product against a three-piece fixed operand takes four bf16 MFMAs (x_lo t_hi + x_hi t_lo2 + x_hi t_lo + x_hi t_hi):
//           relative error <= ~2^-16 per product, measured nMSE against the fp64 oracle 2e-11 .. 5e-11.
template <int AP>
struct CfdAct8 {
    bf16x8 p[AP];
};
template <int AP>
__device__ __forceinline__ CfdAct8<AP> cfd_act_split8(const float (&x)[8]) {
    CfdAct8<AP> r;
    if constexpr (AP == 3) {
        const CfdSplit8x3 s = cfd_split8x3(x);
        r.p[0] = s.p[0]; r.p[1] = s.p[1]; r.p[2] = s.p[2];
    } else {
        const CfdSplit8 s = cfd_split8(x);
        r.p[0] = s.hi; r.p[1] = s.lo;
    }
    return r;
}
// the same for a vector whose ONLY non-zero element is element 0 (the Nyquist row of the 64-wide forward transform): a handful of
// scalar instructions instead of a full eight-value split
template <int AP>
__device__ __forceinline__ CfdAct8<AP> cfd_act_split1(float x) {
    if constexpr (AP == 3) {
        const unsigned h = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
        const float r = x - __builtin_bit_cast(float, h);
        const unsigned l = __builtin_bit_cast(unsigned, r) & 0xffff0000u;
        const float t = r - __builtin_bit_cast(float, l);
        CfdAct8<AP> o;
        o.p[0] = __builtin_bit_cast(bf16x8, cfd_u32x4{h >> 16, 0u, 0u, 0u});
        o.p[1] = __builtin_bit_cast(bf16x8, cfd_u32x4{l >> 16, 0u, 0u, 0u});
        o.p[2] = __builtin_bit_cast(bf16x8, cfd_u32x4{__builtin_bit_cast(unsigned, t) >> 16, 0u, 0u, 0u});
        return o;
    } else {
        const float v[8] = {x, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        return cfd_act_split8<AP>(v);
    }
}
// the (activation piece, table piece) pairs of a product against a three-piece fixed operand, smallest terms first
__host__ __device__ constexpr int cfd_nterm(int ap) { return ap == 3 ? 6 : 4; }
__host__ __device__ constexpr int cfd_term_a(int ap, int k) { return ap == 3 ? (k == 0 ? 2 : (k == 2 || k == 3) ? 1 : 0) : (k == 0 ? 1 : 0); }
__host__ __device__ constexpr int cfd_term_t(int ap, int k) { return ap == 3 ? (k == 1 ? 2 : (k == 2 || k == 4) ? 1 : 0) : (k == 1 ? 2 : k == 2 ? 1 : 0); }
// activation x activation (1x1 weight gradient): 3 pieces: the six of cfd_mfma_bf16x6; 2 pieces: lo*hi, hi*lo, hi*hi
__host__ __device__ constexpr int cfd_nterm_aa(int ap) { return ap == 3 ? 6 : 3; }
__host__ __device__ constexpr int cfd_term_aa_a(int ap, int k) { return ap == 3 ? cfd_term_a(3, k) : (k == 0 ? 1 : 0); }
__host__ __device__ constexpr int cfd_term_aa_b(int ap, int k) { return ap == 3 ? cfd_term_t(3, k) : (k == 1 ? 1 : 0); }
// three pieces of a fixed operand (piece-major tables: [piece][64 lanes] vectors behind `t`)
struct CfdTab3 {
    bf16x8 p[3];
};
__device__ __forceinline__ CfdTab3 cfd_tab3(const bf16x8* t, int lane) {
    CfdTab3 r;
    r.p[0] = t[lane]; r.p[1] = t[64 + lane]; r.p[2] = t[128 + lane];
    return r;
}
// term k of  acc += activation x table  with the activation as the A operand (ACTA) or as the B operand.  With the table as
// the A operand the two-piece route walks its four terms in the order the round-3 kernels used (hi*lo2, hi*lo, lo*hi, hi*hi), so
// that the default route's results stay bit-identical to round 3.
template <bool ACTA, int AP>
__device__ __forceinline__ f32x4 cfd_term(const CfdAct8<AP>& x, const CfdTab3& t, int k, f32x4 c) {
    if constexpr (ACTA) return cfd_mfma16x16x32_bf16(x.p[cfd_term_a(AP, k)], t.p[cfd_term_t(AP, k)], c);
    else if constexpr (AP == 2) return cfd_mfma16x16x32_bf16(t.p[k == 0 ? 2 : k == 1 ? 1 : 0], x.p[k == 2 ? 1 : 0], c);
    else return cfd_mfma16x16x32_bf16(t.p[cfd_term_t(AP, k)], x.p[cfd_term_a(AP, k)], c);
}
int cfd_act_pieces();  // tune.cpp: 2 or 3 (act_pieces knob)
// D += A*B with both operands split
__device__ __forceinline__ f32x4 cfd_mfma_bf16x3(const CfdSplit8& a, const CfdSplit8& b, f32x4 c) {
    c = cfd_mfma16x16x32_bf16(a.lo, b.hi, c);
    c = cfd_mfma16x16x32_bf16(a.hi, b.lo, c);
    return cfd_mfma16x16x32_bf16(a.hi, b.hi, c);
}

// Sum of a contiguous row of n floats by one wave (all lanes get the result): float4 loads when the row is 16-byte
// aligned, all of a lane's loads issued before the adds, fixed order.
__device__ __forceinline__ float cfd_row_sum(const float* __restrict__ row, int n, int lane) {
    float s = 0.f;
    if ((n & 3) == 0 && ((size_t)row & 15) == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
        const int n4 = n >> 2;
        int k = lane;
        for (; k + 192 < n4; k += 256) {  // four independent 1-KiB wave loads in flight
            const float4 a = r4[k], b = r4[k + 64], c = r4[k + 128], d = r4[k + 192];
            s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
        }
        for (; k < n4; k += 64) {
            const float4 a = r4[k];
            s += (a.x + a.y) + (a.z + a.w);
        }
    } else {
        for (int k = lane; k < n; k += 64) s += row[k];
    }
    return cfd_wave_sum(s);
}

// Sum over `nrec` BLOCK-MAJOR partial records (part[rec * stride + e]: what a producer workgroup writes with coalesced stores) of the 16
// consecutive elements e0 .. e0 + 15.  Every thread of the workgroup takes part (blockDim.x a multiple of 64, <= 1024).  lane = (record
// subset bq = lane >> 4, element el = lane & 15): a wave's load instruction reads four 64-byte runs; wave w adds the records
// (4 w + bq) + k (4 waves) in that order, eight loads in flight; the four subsets of a wave meet in cfd_row_sum4, the waves through
// `scratch` (>= 16 * waves floats of LDS) in wave order -- the result depends on nothing but the records and the workgroup size.
// Returns the total of element e0 + threadIdx.x in the threads with threadIdx.x < 16.  (Rounds 1-4 kept such partials ELEMENT-major --
// contiguous rows for the reduction, but every producer store a 4-byte transaction of its own: 1.5 M per head launch, 0.43 M per 1x1
// weight-gradient launch; the head's counter-based write traffic was 1.58x its algorithmic bytes and the kernel 13 us slower.)
__device__ __forceinline__ float cfd_record_sum16(const float* __restrict__ part, int nrec, int stride, int e0, int nelem, float* scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int bq = lane >> 4, el = lane & 15;
    const float* src = part + (e0 + el < nelem ? e0 + el : nelem - 1);
    const int step = 4 * nw;
    float s = 0.f;
    int r = 4 * wave + bq;
    for (; r + 7 * step < nrec; r += 8 * step) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(r + u * step) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; r < nrec; r += step) s += src[(size_t)r * stride];
    s = cfd_row_sum4(s);
    __syncthreads();  // (the readers of a previous call are done with `scratch`)
    if (lane < 16) scratch[wave * 16 + lane] = s;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x < 16)
        for (int w = 0; w < nw; ++w) tot += scratch[w * 16 + threadIdx.x];
    return tot;
}

// ---- division of 31-bit indices by a launch-invariant divisor --------------------------------------------------
// Element-wise kernels turn a flat index into (image, channel, pixel).  A 64-bit integer division costs ~100 VALU
// instructions on gfx950 and the first versions of those kernels did several per element; with the magic number made on
// the host (round-up method for 31-bit numerators: s = ceil(log2 d), m = ceil(2^(31+s) / d) < 2^32) it is one v_mul_hi
// and one shift.  Callers guarantee n < 2^31.
struct CfdDiv {
    unsigned d, m, sh;
};
static inline CfdDiv cfd_div_make(unsigned d) {
    CfdDiv r;
    r.d = d;
    r.m = 0;
    r.sh = 0;
    if (d <= 1) return r;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    r.m = (unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d);
    r.sh = s - 1;
    return r;
}
__device__ __forceinline__ unsigned cfd_div(unsigned n, const CfdDiv& dv) {
    return dv.d <= 1 ? n : (cfd_umulhi(n, dv.m) >> dv.sh);
}
#define CFD_REQUIRE_I31(total, fn) \
    CFD_REQUIRE((total) < (1L << 31), CFD_ERR_UNSUPPORTED, "%s: %ld elements exceed the 2^31 index range of this kernel", fn, (long)(total))
