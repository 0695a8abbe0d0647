// Fno2d projection head fused with the mask and the MseLoss reductions (src/models/fno/fno2d.py:228-237,
// src/models/loss.py:22-37):   preds = mask * fc2(gelu(fc1(h)))   with h = f(a), fc1: C -> 128, fc2: 128 -> out_chan.
//
// The (B,128,H,W) hidden tensor (537 MB at B=256, 64x64) never exists: each wave pushes 64-pixel tiles through
// v_mfma_f32_16x16x4_f32 with the hidden units on the M axis (128 = 8 tiles, K = C channels), applies GELU on the
// accumulator registers and folds fc2 + the loss sums in place.  The backward kernel recomputes the hidden tile and
// produces d/da plus all four parameter gradients; the two contractions over PIXELS (weight gradients) need the tile
// transposed, which goes through a wave-private LDS region (no workgroup barrier inside the tile loop).
#include "cfd_common.h"

#define HEAD_HD 128
#define HEAD_MT 8
#define HEAD_LD 17  // LDS row stride of the transposed tiles (16 pixels + 1 pad -> conflict-free column reads)

static int head_blocks(int B, int HW) {
    const long tiles = (long)B * ((HW + 63) / 64);
    long blocks = (tiles + 7) / 8;  // >= 2 tiles per wave
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// backward: the four waves of a block share each tile (see k_head_bwd)
static int head_bwd_blocks(int B, int HW) {
    const long tiles = (long)B * ((HW + 63) / 64);
    long blocks = tiles < 512 ? tiles : 512;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int KS, bool VEC4, bool ACT>
__device__ __forceinline__ void head_load_h(const float* __restrict__ a, int b, int C, int HW, int px, int q,
                                            float (&h)[KS][4]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int i = 4 * s + q;
        const float* src = a + ((size_t)b * C + i) * HW + px;
        if constexpr (VEC4) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < C && px < HW) t = *reinterpret_cast<const float4*>(src);
            h[s][0] = t.x; h[s][1] = t.y; h[s][2] = t.z; h[s][3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) h[s][j] = (i < C && px + j < HW) ? src[j] : 0.f;
        }
        if constexpr (ACT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) h[s][j] = cfd_gelu(h[s][j]);
        }
    }
}

// A-operand fragments of fc1 for z = W1 h:  frag[mt][s][lane=(q,i)] = w1[16mt+i][4s+q]
template <int KS>
__device__ __forceinline__ void head_build_w1f(float* s_w1f, const float* __restrict__ w1, int C) {
    for (int idx = threadIdx.x; idx < HEAD_MT * KS * 64; idx += blockDim.x) {
        const int ln = idx & 63, s = (idx >> 6) % KS, mt = idx / (64 * KS);
        const int jh = 16 * mt + (ln & 15), i = 4 * s + (ln >> 4);
        s_w1f[idx] = i < C ? w1[jh * C + i] : 0.f;
    }
}

template <int KS, bool VEC4, bool ACT>
__global__ __launch_bounds__(256, 2) void k_head_fwd(const float* __restrict__ a, const float* __restrict__ mask,
                                                  const float* __restrict__ label, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ preds,
                                                  float* __restrict__ part, int B, int C, int Co, int HW) {
    __shared__ float s_w1f[HEAD_MT * KS * 64];
    __shared__ float s_b1[HEAD_HD];
    __shared__ float s_w2[2 * HEAD_HD];
    __shared__ float s_red[12];
    head_build_w1f<KS>(s_w1f, w1, C);
    for (int i = threadIdx.x; i < HEAD_HD; i += blockDim.x) s_b1[i] = b1[i];
    for (int i = threadIdx.x; i < 2 * HEAD_HD; i += blockDim.x) s_w2[i] = (i / HEAD_HD) < Co ? w2[i] : 0.f;
    __syncthreads();
    const float b2v0 = b2[0], b2v1 = Co > 1 ? b2[1] : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    float lsq = 0.f, labs = 0.f, ll2 = 0.f;
    const int tpb = (HW + 63) / 64;
    const long total = (long)B * tpb;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < total; tile += (long)gridDim.x * 4) {
        const int b = (int)(tile / tpb);
        const int px = (int)(tile - (long)b * tpb) * 64 + 4 * n;
        float h[KS][4];
        head_load_h<KS, VEC4, ACT>(a, b, C, HW, px, q, h);
        // The 4 pixel phases j run in a ROLLED loop (one 16-pixel sub-tile per trip keeps the live set at one
        // z tile); the phase being processed always sits in h[s][0] / lands in out*[3], registers rotate each trip.
        float out0[4] = {0.f, 0.f, 0.f, 0.f}, out1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int lo = cfd_opaque(lane), q4 = cfd_opaque(4 * q);  // keep the LDS table reads inside the loop
            f32x4 z[HEAD_MT];
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt) {
                const int jb = 16 * mt + q4;
                z[mt] = f32x4{s_b1[jb], s_b1[jb + 1], s_b1[jb + 2], s_b1[jb + 3]};
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int mt = 0; mt < HEAD_MT; ++mt)
                    z[mt] = cfd_mfma16x16x4(s_w1f[(mt * KS + s) * 64 + lo], h[s][0], z[mt]);
            float o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jh = 16 * mt + q4 + r;
                    const float gl = cfd_gelu(z[mt][r]);
                    o0 = fmaf(s_w2[jh], gl, o0);
                    o1 = fmaf(s_w2[HEAD_HD + jh], gl, o1);
                }
            o0 += cfd_shfl_xor(o0, 16); o0 += cfd_shfl_xor(o0, 32);
            o1 += cfd_shfl_xor(o1, 16); o1 += cfd_shfl_xor(o1, 32);
            out0[0] = out0[1]; out0[1] = out0[2]; out0[2] = out0[3]; out0[3] = o0;
            out1[0] = out1[1]; out1[1] = out1[2]; out1[2] = out1[3]; out1[3] = o1;
#pragma unroll
            for (int s = 0; s < KS; ++s) { h[s][0] = h[s][1]; h[s][1] = h[s][2]; h[s][2] = h[s][3]; }
        }
        if (q < Co) {  // lane group q stores output channel q
            const int c = q;
            const float bias = c == 0 ? b2v0 : b2v1;
            float pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = px + j;
                const bool ok = pix < HW;
                const float mk = (ok && mask) ? mask[(size_t)b * HW + pix] : 1.f;
                pv[j] = ((c == 0 ? out0[j] : out1[j]) + bias) * mk;  // fno2d.py:233
                if (label && ok) {
                    const float lab = label[((size_t)b * Co + c) * HW + pix] * mk;  // fno2d.py:236
                    const float d = pv[j] - lab;
                    lsq = fmaf(d, d, lsq);
                    labs += fabsf(d);
                    ll2 = fmaf(lab, lab, ll2);
                }
            }
            float* dst = preds + ((size_t)b * Co + c) * HW + px;
            if constexpr (VEC4) {
                if (px < HW) *reinterpret_cast<float4*>(dst) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (px + j < HW) dst[j] = pv[j];
            }
        }
    }
    if (part) {
        lsq = cfd_wave_sum(lsq); labs = cfd_wave_sum(labs); ll2 = cfd_wave_sum(ll2);
        if (lane == 0) { s_red[wave] = lsq; s_red[4 + wave] = labs; s_red[8 + wave] = ll2; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const float* s = s_red + 4 * threadIdx.x;
            part[blockIdx.x * 3 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
        }
    }
}

__global__ __launch_bounds__(64) void k_head_loss_final(const float* __restrict__ part, int nblk, float count,
                                                        float* __restrict__ sums) {
    const int lane = threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = lane; k < nblk; k += 64) { a += part[k * 3]; b += part[k * 3 + 1]; c += part[k * 3 + 2]; }
    a = cfd_wave_sum(a); b = cfd_wave_sum(b); c = cfd_wave_sum(c);
    if (lane == 0) { sums[0] = a; sums[1] = b; sums[2] = c; sums[3] = count; }
}

static size_t head_part_floats(int C, int Co) { return (size_t)HEAD_HD * C + HEAD_HD + (size_t)Co * HEAD_HD + Co; }

extern "C" size_t cfd_fno_head_workspace_bytes(int B, int C, int Hd, int Co, int HW) {
    (void)Hd;
    if (B <= 0) return 0;
    const size_t fwd = (size_t)head_blocks(B, HW) * 3 * sizeof(float);
    const size_t bwd = (size_t)head_bwd_blocks(B, HW) * head_part_floats(C, Co) * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

static int head_check(const char* fn, int B, int C, int Hd, int Co, int HW) {
    CFD_REQUIRE(B >= 0 && C >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "%s: bad sizes", fn);
    CFD_REQUIRE(Hd == HEAD_HD, CFD_ERR_UNSUPPORTED, "%s: head width %d unsupported (the reference hard-codes 128, fno2d.py:175)", fn, Hd);
    CFD_REQUIRE(C <= 32, CFD_ERR_UNSUPPORTED, "%s: hidden=%d (max 32) unsupported", fn, C);
    CFD_REQUIRE(Co >= 1 && Co <= 2, CFD_ERR_UNSUPPORTED, "%s: out_chan=%d (max 2) unsupported", fn, Co);
    return CFD_OK;
}

extern "C" int cfd_fno_head_fwd(const float* a, const float* mask, const float* label, const float* w1, const float* b1,
                                const float* w2, const float* b2, float* preds, float* sums, void* ws, int B, int C,
                                int Hd, int Co, int HW, int act_in, void* stream) {
    CFD_REQUIRE(a && w1 && b1 && w2 && b2 && preds, CFD_ERR_INVALID_ARG, "cfd_fno_head_fwd: NULL pointer");
    CFD_REQUIRE(!label || (sums && ws), CFD_ERR_INVALID_ARG, "cfd_fno_head_fwd: label given without sums/workspace");
    CFD_TRY(head_check("cfd_fno_head_fwd", B, C, Hd, Co, HW));
    if (B == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = head_blocks(B, HW);
    float* part = label ? (float*)ws : nullptr;
    const bool v4 = HW % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)preds % 16) == 0;
    {
    CFD_PROF("k_head_fwd", st);
#define CFD_HF(K_, V_, A_)                                                                                      \
    hipLaunchKernelGGL((k_head_fwd<K_, V_, A_>), dim3(blocks), dim3(256), 0, st, a, mask, label, w1, b1, w2, b2, \
                       preds, part, B, C, Co, HW)
#define CFD_HF_VA(K_)                              \
    do {                                           \
        if (v4 && act_in) CFD_HF(K_, true, true);  \
        else if (v4) CFD_HF(K_, true, false);      \
        else if (act_in) CFD_HF(K_, false, true);  \
        else CFD_HF(K_, false, false);             \
    } while (0)
    if (C <= 8) CFD_HF_VA(2);
    else if (C <= 20) CFD_HF_VA(5);
    else CFD_HF_VA(8);
#undef CFD_HF_VA
#undef CFD_HF
    }
    CFD_LAUNCH_CHECK("cfd_fno_head_fwd");
    if (label) {
        hipLaunchKernelGGL(k_head_loss_final, dim3(1), dim3(64), 0, st, (const float*)part, blocks,
                           (float)((double)B * Co * HW), sums);
        CFD_LAUNCH_CHECK("cfd_fno_head_fwd(loss)");
    }
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------
// The four waves of a workgroup share one 64-pixel tile and split the 128 hidden units: wave w owns hidden units
// [32w, 32w+32) (two MFMA M-tiles).  Everything indexed by hidden unit -- the recomputed pre-activation, GELU and its
// derivative, the fc2 weight gradient (per-lane VALU accumulators), the fc1 weight/bias gradient (MFMA, K = pixels,
// operands transposed through a wave-private LDS tile) -- is wave-local, and all weight fragments of the slice are
// loop-invariant registers.  Only d/dh = W1^T gz sums over hidden units: each wave contributes its partial through
// LDS once per tile and the workgroup finishes ga = d/dh * f'(a) with coalesced float4 stores.
// A wave issues 42 MFMAs per 16-pixel phase (10 recompute + 16 d/dh + 16 weight gradient) on ~150 registers, so
// several waves per SIMD overlap MFMA with GELU/LDS work (the one-wave-owns-128-units version needed 450 registers).
template <int KS, bool VEC4, bool ACT>
__global__ __launch_bounds__(256, 2) void k_head_bwd(
    const float* __restrict__ a, const float* __restrict__ mask, const float* __restrict__ label,
    const float* __restrict__ preds, const float* __restrict__ gext, const float* __restrict__ coef,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float* __restrict__ ga,
    float* __restrict__ part, int B, int C, int Co, int HW) {
    constexpr int CP = 4 * KS;                   // padded channel count
    constexpr int MU = (CP + 15) / 16;           // M tiles of the channel axis for d/dh
    constexpr int NTI = (CP + 1 + 15) / 16;      // N tiles of [channels | ones] for the fc1 weight/bias gradient
    constexpr int XR = 32 * HEAD_LD;             // transposed gz tile of one wave   [32 hidden][17]
    constexpr int HR = 16 * NTI * HEAD_LD;       // transposed input tile of one wave [16*NTI][17]
    __shared__ float s_t[4 * (XR + HR)];
    __shared__ float4 s_red[4 * CP * 16];        // [wave][channel][16 x float4 = 64 pixels] partial d/dh
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    for (int i = threadIdx.x; i < 4 * (XR + HR); i += blockDim.x) s_t[i] = 0.f;
    float* s_x = s_t + wave * (XR + HR);
    float* s_h = s_x + XR;
    // ---- loop-invariant fragments of this wave's hidden slice (global hidden tile mt = 2*wave + t) ----
    float w1f[2][KS];   // A operand of z = W1 h:        w1[16mt + n][4s + q]
    float w1t[MU][8];   // A operand of d/dh = W1^T gz:  w1[16mt + 4q + r][16mu + n], k-step 4t + r
    float bz[2][4], w2a[2][4], w2b[2][4];  // b1, w2[0], w2[1] at hidden unit 16mt + 4q + r
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int mt = 2 * wave + t;
#pragma unroll
        for (int s = 0; s < KS; ++s) w1f[t][s] = (4 * s + q < C) ? w1[(16 * mt + n) * C + 4 * s + q] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jh = 16 * mt + 4 * q + r;
            bz[t][r] = b1[jh];
            w2a[t][r] = w2[jh];
            w2b[t][r] = Co > 1 ? w2[HEAD_HD + jh] : 0.f;
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) w1t[mu][4 * t + r] = (16 * mu + n < C) ? w1[jh * C + 16 * mu + n] : 0.f;
        }
    }
    const float c0 = label ? coef[0] : 0.f, c1 = label ? coef[1] : 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 aw1[2][NTI];
    float acc2a[2][4], acc2b[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int v = 0; v < NTI; ++v) aw1[t][v] = zero;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc2a[t][r] = 0.f; acc2b[t][r] = 0.f; }
    }
    float gb2a0 = 0.f, gb2a1 = 0.f;
    __syncthreads();
    const int tpb = (HW + 63) / 64;
    const long total = (long)B * tpb;
    for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {  // all four waves walk the same tiles
        const int b = (int)(tile / tpb);
        const int px0 = (int)(tile - (long)b * tpb) * 64;
        const int px = px0 + 4 * n;
        float h[KS][4];
        head_load_h<KS, VEC4, ACT>(a, b, C, HW, px, q, h);
        // upstream gradient on the raw head output for this lane's 4 pixels (same for every q)
        float gr0[4], gr1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = px + j;
            gr0[j] = 0.f;
            gr1[j] = 0.f;
            if (pix < HW) {
                const float mk = mask ? mask[(size_t)b * HW + pix] : 1.f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c < Co) {
                        const size_t off = ((size_t)b * Co + c) * HW + pix;
                        float gp = gext ? gext[off] : 0.f;
                        if (label) {
                            const float d = preds[off] - label[off] * mk;
                            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                            gp += c0 * 2.f * d + c1 * sg;
                        }
                        if (c == 0) gr0[j] = gp * mk; else gr1[j] = gp * mk;
                    }
                }
            }
        }
        if (wave == 0 && q == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { gb2a0 += gr0[j]; gb2a1 += gr1[j]; }
        }
        // The 4 pixel phases, unrolled (static register indices: no rotation moves).
        f32x4 gh[MU][4];
#pragma unroll
        for (int mu = 0; mu < MU; ++mu)
#pragma unroll
            for (int j = 0; j < 4; ++j) gh[mu][j] = zero;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // 1. recompute this wave's slice of the hidden pre-activation z[hidden][pixel n]
            f32x4 z[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) z[t] = f32x4{bz[t][0], bz[t][1], bz[t][2], bz[t][3]};
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) z[t] = cfd_mfma16x16x4(w1f[t][s], h[s][j], z[t]);
            // 2. a1 = gelu(z) feeds the fc2 weight gradient; gz = (W2^T graw) * gelu'(z) replaces z
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float Phi, e;
                    cfd_gelu_terms(z[t][r], Phi, e);
                    const float a1 = z[t][r] * Phi;
                    acc2a[t][r] = fmaf(gr0[j], a1, acc2a[t][r]);
                    acc2b[t][r] = fmaf(gr1[j], a1, acc2b[t][r]);
                    const float ga1 = fmaf(w2a[t][r], gr0[j], w2b[t][r] * gr1[j]);
                    const float gz = ga1 * fmaf(z[t][r] * CFD_INV_SQRT_2PI, e, Phi);
                    z[t][r] = gz;
                    s_x[(16 * t + 4 * q + r) * HEAD_LD + n] = gz;
                }
            // 3. transposed input tile [channel | ones][pixel]
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (4 * s + q < C) s_h[(4 * s + q) * HEAD_LD + n] = h[s][j];
            if (q == 0) s_h[C * HEAD_LD + n] = (px + j < HW) ? 1.f : 0.f;
            cfd_wave_lds_sync();
            // 4. partial d/dh[i][pixel] = sum over this wave's hidden units w1[jh][i] gz[jh][pixel]
            f32x4 ghc[MU];
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) ghc[mu] = zero;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mu = 0; mu < MU; ++mu) ghc[mu] = cfd_mfma16x16x4(w1t[mu][4 * t + r], z[t][r], ghc[mu]);
            // 5. gw1[jh][i] (+ gb1 through the ones column) += sum_pixel gz[jh][pixel] h[i][pixel]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float bh[NTI];
#pragma unroll
                for (int v = 0; v < NTI; ++v) bh[v] = s_h[(16 * v + n) * HEAD_LD + 4 * s4 + q];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float ax = s_x[(16 * t + n) * HEAD_LD + 4 * s4 + q];
#pragma unroll
                    for (int v = 0; v < NTI; ++v) aw1[t][v] = cfd_mfma16x16x4(ax, bh[v], aw1[t][v]);
                }
            }
            cfd_wave_lds_sync();
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) gh[mu][j] = ghc[mu];
        }
        // this wave's partial d/dh of the tile: [channel][64 pixels], pixel = 4n + phase
#pragma unroll
        for (int mu = 0; mu < MU; ++mu)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * mu + 4 * q + r;
                if (i < CP) s_red[(wave * CP + i) * 16 + n] = make_float4(gh[mu][0][r], gh[mu][1][r], gh[mu][2][r], gh[mu][3][r]);
            }
        __syncthreads();
        // ga[b][i][px0 .. px0+63] = (sum over the four hidden slices) * f'(a)
        for (int e = threadIdx.x; e < C * 16; e += blockDim.x) {
            const int i = e >> 4, n4 = e & 15;
            float4 v = s_red[i * 16 + n4];
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const float4 u = s_red[(wv * CP + i) * 16 + n4];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const int p4 = px0 + 4 * n4;
            const size_t off = ((size_t)b * C + i) * HW + p4;
            if constexpr (VEC4) {
                if (p4 < HW) {
                    if constexpr (ACT) {
                        const float4 av = *reinterpret_cast<const float4*>(a + off);
                        v.x *= cfd_gelu_grad(av.x); v.y *= cfd_gelu_grad(av.y);
                        v.z *= cfd_gelu_grad(av.z); v.w *= cfd_gelu_grad(av.w);
                    }
                    *reinterpret_cast<float4*>(ga + off) = v;
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p4 + j < HW) {
                        float x = vv[j];
                        if constexpr (ACT) x *= cfd_gelu_grad(a[off + j]);
                        ga[off + j] = x;
                    }
            }
        }
        __syncthreads();
    }
    // ---- this block's partial parameter gradients: [gw1 128*C | gb1 128 | gw2 Co*128 | gb2 Co] ----
    const int o_gb1 = HEAD_HD * C, o_gw2 = o_gb1 + HEAD_HD, o_gb2 = o_gw2 + Co * HEAD_HD;
    float* dst = part + (size_t)blockIdx.x * (o_gb2 + Co);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jh = 16 * (2 * wave + t) + 4 * q + r;
#pragma unroll
            for (int v = 0; v < NTI; ++v) {
                const int i = 16 * v + n;
                if (i < C) dst[jh * C + i] = aw1[t][v][r];
                else if (i == C) dst[o_gb1 + jh] = aw1[t][v][r];
            }
            float sa = acc2a[t][r], sb = acc2b[t][r];  // sum over the 16 pixel lanes of this q group
#pragma unroll
            for (int m = 1; m <= 8; m <<= 1) { sa += cfd_shfl_xor(sa, m); sb += cfd_shfl_xor(sb, m); }
            if (n == 0) {
                dst[o_gw2 + jh] = sa;
                if (Co > 1) dst[o_gw2 + HEAD_HD + jh] = sb;
            }
        }
    if (wave == 0) {
        gb2a0 = cfd_wave_sum(gb2a0);
        gb2a1 = cfd_wave_sum(gb2a1);
        if (lane == 0) {
            dst[o_gb2] = gb2a0;
            if (Co > 1) dst[o_gb2 + 1] = gb2a1;
        }
    }
}

// One wave per output element (see k_wgrad_reduce).
__global__ __launch_bounds__(256) void k_head_reduce(const float* __restrict__ part, int nblk, int PS,
                                                     float* __restrict__ gw1, float* __restrict__ gb1,
                                                     float* __restrict__ gw2, float* __restrict__ gb2, int C, int Co) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= PS) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += part[(size_t)k * PS + e];
    s = cfd_wave_sum(s);
    if (lane == 0) {
        const int o_gb1 = HEAD_HD * C, o_gw2 = o_gb1 + HEAD_HD, o_gb2 = o_gw2 + Co * HEAD_HD;
        if (e < o_gb1) gw1[e] = s;
        else if (e < o_gw2) gb1[e - o_gb1] = s;
        else if (e < o_gb2) gw2[e - o_gw2] = s;
        else gb2[e - o_gb2] = s;
    }
}

extern "C" int cfd_fno_head_bwd(const float* a, const float* mask, const float* label, const float* preds,
                                const float* gpreds_ext, const float* coef, const float* w1, const float* b1,
                                const float* w2, float* ga, float* gw1, float* gb1, float* gw2, float* gb2, void* ws,
                                int B, int C, int Hd, int Co, int HW, int act_in, void* stream) {
    CFD_REQUIRE(a && w1 && b1 && w2 && ga && gw1 && gb1 && gw2 && gb2 && ws, CFD_ERR_INVALID_ARG, "cfd_fno_head_bwd: NULL pointer");
    CFD_REQUIRE(!label || (preds && coef), CFD_ERR_INVALID_ARG, "cfd_fno_head_bwd: label given without preds/coef");
    CFD_REQUIRE(label || gpreds_ext, CFD_ERR_INVALID_ARG, "cfd_fno_head_bwd: neither a loss (label) nor an upstream gradient given");
    CFD_TRY(head_check("cfd_fno_head_bwd", B, C, Hd, Co, HW));
    CFD_REQUIRE(B >= 1, CFD_ERR_INVALID_ARG, "cfd_fno_head_bwd: empty batch");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = head_bwd_blocks(B, HW);
    float* part = (float*)ws;
    const bool v4 = HW % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)ga % 16) == 0;
    {
    CFD_PROF("k_head_bwd", st);
#define CFD_HB(K_, V_, A_)                                                                                         \
    hipLaunchKernelGGL((k_head_bwd<K_, V_, A_>), dim3(blocks), dim3(256), 0, st, a, mask, label, preds, gpreds_ext, \
                       coef, w1, b1, w2, ga, part, B, C, Co, HW)
#define CFD_HB_VA(K_)                              \
    do {                                           \
        if (v4 && act_in) CFD_HB(K_, true, true);  \
        else if (v4) CFD_HB(K_, true, false);      \
        else if (act_in) CFD_HB(K_, false, true);  \
        else CFD_HB(K_, false, false);             \
    } while (0)
    if (C <= 8) CFD_HB_VA(2);
    else if (C <= 20) CFD_HB_VA(5);
    else CFD_HB_VA(8);
#undef CFD_HB_VA
#undef CFD_HB
    }
    CFD_LAUNCH_CHECK("cfd_fno_head_bwd");
    const int PS = (int)head_part_floats(C, Co);
    CFD_PROF("k_head_reduce", st);
    hipLaunchKernelGGL(k_head_reduce, dim3((PS + 3) / 4), dim3(256), 0, st, (const float*)part, blocks, PS, gw1,
                       gb1, gw2, gb2, C, Co);
    CFD_LAUNCH_CHECK("cfd_fno_head_bwd(reduce)");
    return CFD_OK;
}
