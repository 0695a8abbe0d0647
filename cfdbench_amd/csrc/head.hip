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
    const size_t nb = (size_t)head_blocks(B, HW);
    const size_t fwd = nb * 3 * sizeof(float);
    const size_t bwd = nb * head_part_floats(C, Co) * sizeof(float);
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
template <int KS, bool VEC4, bool ACT>
__global__ __launch_bounds__(256, 1) void k_head_bwd(
    const float* __restrict__ a, const float* __restrict__ mask, const float* __restrict__ label,
    const float* __restrict__ preds, const float* __restrict__ gext, const float* __restrict__ coef,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float* __restrict__ ga,
    float* __restrict__ part, int B, int C, int Co, int HW) {
    constexpr int MU = (4 * KS + 15) / 16;       // M tiles of the channel axis for d/dh
    constexpr int NTI = (4 * KS + 1 + 15) / 16;  // N tiles of [channels | ones] for the fc1 weight/bias gradient
    constexpr int XR = HEAD_HD * HEAD_LD;        // transposed hidden tile  [128][17]
    constexpr int HR = 16 * NTI * HEAD_LD;       // transposed input tile   [16*NTI][17]
    constexpr int GR = 16 * HEAD_LD;             // transposed output-grad tile [16][17]
    constexpr int WAVE_LDS = XR + HR + GR;
    __shared__ float s_w1f[HEAD_MT * KS * 64];
    __shared__ float s_w1t[MU * 32 * 64];
    __shared__ float s_b1[HEAD_HD];
    __shared__ float s_w2[2 * HEAD_HD];
    __shared__ float s_tile[4 * WAVE_LDS];
    head_build_w1f<KS>(s_w1f, w1, C);
    // A-operand fragments for d/dh = W1^T gz:  frag[mu][step=(mt,r)][lane=(q,i)] = w1[16mt+4q+r][16mu+i]
    for (int idx = threadIdx.x; idx < MU * 32 * 64; idx += blockDim.x) {
        const int ln = idx & 63, step = (idx >> 6) & 31, mu = idx >> 11;
        const int jh = 16 * (step >> 2) + 4 * (ln >> 4) + (step & 3), i = 16 * mu + (ln & 15);
        s_w1t[idx] = i < C ? w1[jh * C + i] : 0.f;
    }
    for (int i = threadIdx.x; i < HEAD_HD; i += blockDim.x) s_b1[i] = b1[i];
    for (int i = threadIdx.x; i < 2 * HEAD_HD; i += blockDim.x) s_w2[i] = (i / HEAD_HD) < Co ? w2[i] : 0.f;
    for (int i = threadIdx.x; i < 4 * WAVE_LDS; i += blockDim.x) s_tile[i] = 0.f;
    __syncthreads();
    const float c0 = label ? coef[0] : 0.f, c1 = label ? coef[1] : 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    float* s_x = s_tile + wave * WAVE_LDS;
    float* s_h = s_x + XR;
    float* s_g = s_h + HR;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 aw1[HEAD_MT][NTI], aw2[HEAD_MT];
#pragma unroll
    for (int mt = 0; mt < HEAD_MT; ++mt) {
        aw2[mt] = zero;
#pragma unroll
        for (int v = 0; v < NTI; ++v) aw1[mt][v] = zero;
    }
    float gb2a0 = 0.f, gb2a1 = 0.f;
    const int tpb = (HW + 63) / 64;
    const long total = (long)B * tpb;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < total; tile += (long)gridDim.x * 4) {
        const int b = (int)(tile / tpb);
        const int px = (int)(tile - (long)b * tpb) * 64 + 4 * n;
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
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { gb2a0 += gr0[j]; gb2a1 += gr1[j]; }
        }
        // Rolled loop over the 4 pixel phases (see k_head_fwd): phase data sits in h[s][0] / gr*[0], the phase
        // result is rotated into gh[mu][3].
        f32x4 gh[MU][4];
#pragma unroll
        for (int mu = 0; mu < MU; ++mu)
#pragma unroll
            for (int j = 0; j < 4; ++j) gh[mu][j] = zero;
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int lo = cfd_opaque(lane), q4 = cfd_opaque(4 * q);  // keep the LDS table reads inside the loop
            f32x4 ghc[MU];
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) ghc[mu] = zero;
            // 1. recompute the hidden pre-activation tile z[jh][pixel n]
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
            // 2. transposed tiles: a1 = gelu(z) [jh][px], h [i | ones][px], graw [c][px]
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_x[(16 * mt + 4 * q + r) * HEAD_LD + n] = cfd_gelu(z[mt][r]);
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (4 * s + q < C) s_h[(4 * s + q) * HEAD_LD + n] = h[s][0];
            if (q == 0) {
                s_h[C * HEAD_LD + n] = (px + j < HW) ? 1.f : 0.f;
                s_g[n] = gr0[0];
                s_g[HEAD_LD + n] = gr1[0];
            }
            cfd_wave_lds_sync();
            // 3. gw2[c][jh] += sum_px graw[c][px] a1[jh][px]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float bg = s_g[n * HEAD_LD + 4 * s4 + q];
#pragma unroll
                for (int mt = 0; mt < HEAD_MT; ++mt)
                    aw2[mt] = cfd_mfma16x16x4(s_x[(16 * mt + n) * HEAD_LD + 4 * s4 + q], bg, aw2[mt]);
            }
            cfd_wave_lds_sync();
            // 4. gz = (W2^T graw) * gelu'(z)   (kept in z and written transposed)
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jh = 16 * mt + q4 + r;
                    const float ga1 = fmaf(s_w2[jh], gr0[0], s_w2[HEAD_HD + jh] * gr1[0]);
                    const float gz = ga1 * cfd_gelu_grad(z[mt][r]);
                    z[mt][r] = gz;
                    s_x[jh * HEAD_LD + n] = gz;
                }
            cfd_wave_lds_sync();
            // 5. d/dh[i][px] += sum_jh w1[jh][i] gz[jh][px]   (gz registers are already the B operand)
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mu = 0; mu < MU; ++mu)
                        ghc[mu] = cfd_mfma16x16x4(s_w1t[(mu * 32 + mt * 4 + r) * 64 + lo], z[mt][r], ghc[mu]);
            // 6. gw1[jh][i] (+ gb1 through the ones column) += sum_px gz[jh][px] h[i][px]
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float bh[NTI];
#pragma unroll
                for (int v = 0; v < NTI; ++v) bh[v] = s_h[(16 * v + n) * HEAD_LD + 4 * s4 + q];
#pragma unroll
                for (int mt = 0; mt < HEAD_MT; ++mt) {
                    const float ax = s_x[(16 * mt + n) * HEAD_LD + 4 * s4 + q];
#pragma unroll
                    for (int v = 0; v < NTI; ++v) aw1[mt][v] = cfd_mfma16x16x4(ax, bh[v], aw1[mt][v]);
                }
            }
            cfd_wave_lds_sync();
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) { gh[mu][0] = gh[mu][1]; gh[mu][1] = gh[mu][2]; gh[mu][2] = gh[mu][3]; gh[mu][3] = ghc[mu]; }
#pragma unroll
            for (int s = 0; s < KS; ++s) { h[s][0] = h[s][1]; h[s][1] = h[s][2]; h[s][2] = h[s][3]; }
            gr0[0] = gr0[1]; gr0[1] = gr0[2]; gr0[2] = gr0[3];
            gr1[0] = gr1[1]; gr1[1] = gr1[2]; gr1[2] = gr1[3];
        }
        // epilogue: ga[b][i][px..px+3] = d/dh * f'(a)
#pragma unroll
        for (int mu = 0; mu < MU; ++mu)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * mu + 4 * q + r;
                if (i < C) {
                    const size_t off = ((size_t)b * C + i) * HW + px;
                    float v[4] = {gh[mu][0][r], gh[mu][1][r], gh[mu][2][r], gh[mu][3][r]};
                    if constexpr (VEC4) {
                        if (px < HW) {
                            if constexpr (ACT) {
                                const float4 av = *reinterpret_cast<const float4*>(a + off);
                                v[0] *= cfd_gelu_grad(av.x); v[1] *= cfd_gelu_grad(av.y);
                                v[2] *= cfd_gelu_grad(av.z); v[3] *= cfd_gelu_grad(av.w);
                            }
                            *reinterpret_cast<float4*>(ga + off) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (px + j < HW) {
                                float vv = v[j];
                                if constexpr (ACT) vv *= cfd_gelu_grad(a[off + j]);
                                ga[off + j] = vv;
                            }
                    }
                }
            }
    }
    // ---- block reduction of the parameter-gradient accumulators (wave after wave, fixed order) ----
    gb2a0 = cfd_wave_sum(gb2a0);
    gb2a1 = cfd_wave_sum(gb2a1);
    __syncthreads();
    float* red = s_tile;  // [gw1 128*C | gb1 128 | gw2 Co*128 | gb2 Co]
    const int o_gb1 = HEAD_HD * C, o_gw2 = o_gb1 + HEAD_HD, o_gb2 = o_gw2 + Co * HEAD_HD;
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jh = 16 * mt + 4 * q + r;
#pragma unroll
                    for (int v = 0; v < NTI; ++v) {
                        const int i = 16 * v + n;
                        if (i <= C) {
                            const int off = i < C ? jh * C + i : o_gb1 + jh;
                            red[off] = (wv == 0 ? 0.f : red[off]) + aw1[mt][v][r];
                        }
                    }
                    if (n < Co) {
                        const int off = o_gw2 + n * HEAD_HD + jh;
                        red[off] = (wv == 0 ? 0.f : red[off]) + aw2[mt][r];
                    }
                }
            if (lane == 0) {
                red[o_gb2] = (wv == 0 ? 0.f : red[o_gb2]) + gb2a0;
                if (Co > 1) red[o_gb2 + 1] = (wv == 0 ? 0.f : red[o_gb2 + 1]) + gb2a1;
            }
        }
        __syncthreads();
    }
    const int PS = o_gb2 + Co;
    float* dst = part + (size_t)blockIdx.x * PS;
    for (int e = threadIdx.x; e < PS; e += blockDim.x) dst[e] = red[e];
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
    const int blocks = head_blocks(B, HW);
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
