// Fno2d projection head fused with the mask and the MseLoss reductions (src/models/fno/fno2d.py:228-237,
// src/models/loss.py:22-37):   preds = mask * fc2(gelu(fc1(h)))   with h = f(a), fc1: C -> 128, fc2: 128 -> out_chan.
//
// The (B,128,H,W) hidden tensor (537 MB at B=256, 64x64) never exists: each wave pushes 64-pixel tiles through the
// matrix cores with the hidden units on the M axis (128 = 8 tiles, K = channels), applies GELU on the accumulator
// registers and folds fc2 + the loss sums in place.  The backward kernel recomputes the hidden tile and produces d/da
// plus all four parameter gradients.  These kernels are compute-bound (17 GFLOP of fc1-shaped GEMMs + 134 M GELUs per
// step at B=256), and fp32 MFMA on gfx950 only matches the VALU's fp32 rate without overlapping it, so the GEMMs run
// as split-bf16 3-term products on v_mfma_f32_16x16x32_bf16 (cfd_common.h, relative error ~2^-16, fp32 accumulate).
#include "cfd_common.h"
#include "cfd_tail.h"

#define HEAD_HD 128
#define HEAD_MT 8
#ifndef CFD_HEAD_FWD_BLOCKS
#define CFD_HEAD_FWD_BLOCKS 1024  // 4 resident workgroups per CU (124 VGPRs)
#endif
#ifndef CFD_HEAD_EXP
#define CFD_HEAD_EXP 0  // timing experiments on the one-pass training head (tools/build_variant.sh -DCFD_HEAD_EXP=..; results are WRONG):
                        // 1 no mid-pair barrier, 2 no tile-end barriers, 4 trivial GELU terms, 8 no fc1 MFMAs, 16 no d/dh + gw1 MFMAs,
                        // 32 one-piece split of gz, 64 no transposed gz stores
#endif
#ifndef CFD_HB_UNROLL
#define CFD_HB_UNROLL 1  // phases of k_head_bwd's rolled loop per trip (experiments: tools/build_variant.sh)
#endif
#ifndef CFD_HF_UNROLL
#define CFD_HF_UNROLL 1
#endif
#define CFD_PRAGMA_(x) _Pragma(#x)
#define CFD_UNROLL(n) CFD_PRAGMA_(unroll n)
#ifndef CFD_HF_OCC8
#define CFD_HF_OCC8 2  // workgroups per CU the 32-channel forward head is compiled for (3 = 168 VGPRs: 8-10 registers spilled)
#endif
#define HEAD_LD 17  // LDS row stride of the transposed tiles (16 pixels + 1 pad -> conflict-free column reads)

// Workgroup counts.  The `head_blocks` knob (tests: several tiles per workgroup at small sizes) can only LOWER them, and the
// workspace is sized WITHOUT the knob (knob = false), so a knob changed after a workspace was sized and cached can never make a
// launch write partial sums past its end.
static int head_blocks(int B, int HW, bool knob = true, int C = 20) {
    const long tiles = (long)B * ((HW + 63) / 64);
    long blocks = (tiles + 7) / 8;  // >= 2 tiles per wave
    if (blocks > CFD_HEAD_FWD_BLOCKS) blocks = CFD_HEAD_FWD_BLOCKS;
    // Whole rounds of resident workgroups (round 5): the waves stride over the tiles, so 536 workgroups on the 512 slots of the
    // 32-channel kernel (two per CU) ran a second, nearly empty round -- 54.6 us against 46.6 at 512 (64 rollout cases at 66 x 65);
    // 1024 on the 768 slots of the narrower kernels likewise.  (Fewer workgroups than slots: unchanged.)
    // (C <= 0: the workspace query -- no rounding, an upper bound of every rounded count)
    const long slots = 256L * (C > 20 ? CFD_HF_OCC8 : 3);
    if (C > 0 && blocks > slots) blocks = slots * (blocks / slots);
    const int k = knob ? cfd_tune_get(CFD_TUNE_HEAD_BLOCKS) : -1;
    if (k > 0 && blocks > k) blocks = k;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// backward: the four waves of a block share each tile (see k_head_bwd)
static int head_bwd_blocks(int B, int HW, bool knob = true) {
    const long tiles = (long)B * ((HW + 63) / 64);
    long blocks = tiles < 512 ? tiles : 512;
    const int k = knob ? cfd_tune_get(CFD_TUNE_HEAD_BLOCKS) : -1;  // tests reach the multi-tile loop
    if (k > 0 && blocks > k) blocks = k;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// Input tile in the K = 32 operand layout of v_mfma_f32_16x16x32_bf16: lane group q owns channels CQ q .. CQ q + CQ - 1 with
// CQ >= ceil(C / 4) a compile-time 2 / 5 / 8 (k-slot v <-> channel CQ q + v for v < CQ, zero otherwise) of the lane's four
// pixels px .. px+3.  Dealing
// the channels evenly over the four lane groups (5 each at C = 20) instead of 8 per group keeps the GELU-on-load work at
// C values per pixel: with 8 per group the groups q = 2, 3 evaluated GELU on 12 zero slots (37 % of the input GELUs, which
// are half of all GELUs of the forward head).
template <int CQ, bool VEC4, typename TA = float>
__device__ __forceinline__ void head_load_raw(const TA* __restrict__ a, int b, int C, int HW, int px, int q,
                                              float (&h)[CQ][4]) {
    if constexpr (!VEC4 && sizeof(TA) == 4) {
        // planes that are not 16-byte aligned (66 x 65): ONE branch for the lane's four pixels -- inside the plane: an unaligned
        // 16-byte load per channel, all issued back to back; across the plane's end: clamped scalars (head_mask_raw zeroes the rest)
        if (b >= 0 && px + 3 < HW) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
                const int i = CQ * q + c;
                const float4 t = cfd_ld4u(reinterpret_cast<const float*>(a) + ((size_t)b * C + (i < C ? i : C - 1)) * HW + px);
                h[c][0] = t.x; h[c][1] = t.y; h[c][2] = t.z; h[c][3] = t.w;
            }
            return;
        }
    }
#pragma unroll
    for (int c = 0; c < CQ; ++c) {
        h[c][0] = h[c][1] = h[c][2] = h[c][3] = 0.f;
        if (b < 0) continue;  // wave-uniform: past the end of the work, no load
        const int i = CQ * q + c;
        const TA* src = a + ((size_t)b * C + i) * HW + px;
        if constexpr (VEC4 && sizeof(TA) == 4) {
            if (i < C && px < HW) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                h[c][0] = t.x; h[c][1] = t.y; h[c][2] = t.z; h[c][3] = t.w;
            }
        } else {
            // Unconditional loads from clamped addresses: the caller zeroes what lies outside (head_mask_raw) when it takes the
            // values over, a tile later.  (`inside ? load : 0` compiles to a branch with a full wait per element: 32 serialized round
            // trips per tile in the bf16-storage head, which was slower than the fp32 one.)
            const TA* row = a + ((size_t)b * C + (i < C ? i : C - 1)) * HW;
            if (sizeof(TA) == 2 && (HW & 1) == 0) {  // pixel pairs: px is a multiple of 4 and the row starts on a 4-byte boundary
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int pk = px + 2 * k < HW ? px + 2 * k : (px < HW ? px : 0);
                    const unsigned t = *reinterpret_cast<const unsigned*>(row + pk);
                    h[c][2 * k] = __builtin_bit_cast(float, t << 16);
                    h[c][2 * k + 1] = __builtin_bit_cast(float, t & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) h[c][j] = cfd_ld(row + (px + j < HW ? px + j : HW - 1));
            }
        }
    }
}

// zero the slots of head_load_raw's scalar path that lie outside the tensor (b, px: the tile the values were loaded for)
template <int CQ, bool VEC4, typename TA = float>
__device__ __forceinline__ void head_mask_raw(float (&h)[CQ][4], int b, int C, int HW, int px, int q) {
    if constexpr (!(VEC4 && sizeof(TA) == 4)) {
#pragma unroll
        for (int c = 0; c < CQ; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = cfd_opaque_f(h[c][j]);
                h[c][j] = (b >= 0 && CQ * q + c < C && px + j < HW) ? v : 0.f;
            }
    }
}

template <int CQ>
__device__ __forceinline__ void head_act(float (&h)[CQ][4]) {
#pragma unroll
    for (int c = 0; c < CQ; ++c) cfd_gelu4(h[c][0], h[c][1], h[c][2], h[c][3]);
}

// split-bf16 A-operand fragments of fc1 for z = W1 h:  frag[mt][lane=(q,i)][v] = w1[16mt+i][CQ q + v]  (v < CQ)
// AP pieces per value (cfd_common.h: 2 = the default split products, 3 = fp32-exact class); piece p of fragment idx at s_w[p * HEAD_MT * 64 + idx]
template <int AP>
__device__ __forceinline__ void head_build_w1f(bf16x8* s_w, const float* __restrict__ w1, int C, int CQ) {
#pragma unroll
    for (int i0 = 0; i0 < HEAD_MT * 64; i0 += 256) {  // (workgroups of 256 threads: compile-time trips, every trip's loads issued together)
        const int idx = i0 + (int)threadIdx.x;
        if (idx >= HEAD_MT * 64) break;
        const int ln = idx & 63, mt = idx >> 6;
        const int jh = 16 * mt + (ln & 15), c0 = CQ * (ln >> 4);
        float x[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) x[v] = (v < CQ && c0 + v < C) ? w1[jh * C + c0 + v] : 0.f;
        const CfdAct8<AP> s = cfd_act_split8<AP>(x);
#pragma unroll
        for (int pc = 0; pc < AP; ++pc) s_w[pc * HEAD_MT * 64 + idx] = s.p[pc];
    }
}

template <int CQ, bool VEC4, bool ACT, typename TA, int AP>
__global__ __launch_bounds__(256, CQ == 8 ? CFD_HF_OCC8 : 3) void k_head_fwd(const TA* __restrict__ a, const float* __restrict__ mask,
                                                  const float* __restrict__ label, const float* __restrict__ w1,
                                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                                  const float* __restrict__ b2, float* __restrict__ preds,
                                                  float* __restrict__ part, int B, int C, int Co, int HW) {
    __shared__ bf16x8 s_w1[AP * HEAD_MT * 64];  // [piece][M tile][lane]
    __shared__ __attribute__((aligned(16))) float s_b1[HEAD_HD];
    // fc2 weights by PAIRS of hidden units: a packed FMA multiplies the GELU pair (g[jh], g[jh+1]) by (w2[c][jh], w2[c][jh+1]) with
    // plain operands.  (Round 2 paired the two OUTPUTS instead and broadcast one g through the instruction's operand select,
    // `v_pk_fma_f32 ... op_sel:[0,1,0]`: beside another process' k_head_bwd the low half of that form did not accumulate in lanes
    // 48-63 -- profiles/r03_det_root_cause.md.)
    __shared__ __attribute__((aligned(16))) cfd_f2 s_w2[HEAD_HD];
    __shared__ float s_red[12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    float lsq = 0.f, labs = 0.f, ll2 = 0.f;
    const int tpb = (HW + 63) / 64;
    const int total = B * tpb;  // < 2^30 (checked by the launcher)
    const int stride = (int)gridDim.x * 4;
    // a wave's tiles: the raw activations, mask and label of tile k+1 are in flight while tile k is processed (each wave
    // works alone on its tiles, so a load that is waited for right after its issue costs the wave a whole memory latency)
    int tile = (int)blockIdx.x * 4 + wave;
    int b = -1, px = 0;
    auto locate = [&](int t, int& bb, int& pp) {
        bb = -1;
        pp = 0;
        if (t < total) {
            const unsigned ub = (unsigned)t / (unsigned)tpb;
            bb = (int)ub;
            pp = (int)((unsigned)t - ub * (unsigned)tpb) * 64 + 4 * n;
        }
    };
    float hn[CQ][4];  // raw activations of the next tile
    float mkn[4], lbn[4];  // mask / label (channel q) of the next tile's four pixels
    auto fetch_io = [&](int bb, int pp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = bb >= 0 && pp + j < HW;
            mkn[j] = (ok && mask) ? mask[(size_t)bb * HW + pp + j] : 1.f;
            lbn[j] = (ok && label && q < Co) ? label[((size_t)bb * Co + q) * HW + pp + j] : 0.f;
        }
    };
    locate(tile, b, px);
    head_load_raw<CQ, VEC4, TA>(a, b, C, HW, px, q, hn);
    fetch_io(b, px);
    // The weights go to LDS AFTER the first tile has been requested, in trips of a compile-time 256 threads (round 6: with `i += blockDim.x`
    // the three copy loops stayed rolled -- load, s_waitcnt vmcnt(0), ds_write per trip -- and the first tile was requested behind them:
    // five dependent round trips in front of a wave's only tile at 64 rollout cases).
    head_build_w1f<AP>(s_w1, w1, C, CQ);
#pragma unroll
    for (int i0 = 0; i0 < HEAD_HD; i0 += 256) {
        const int i = i0 + (int)threadIdx.x;
        if (i < HEAD_HD) {
            s_b1[i] = b1[i];
            // per PAIR of hidden units (2p, 2p+1): [w2[0][2p], w2[0][2p+1]], [w2[1][2p], w2[1][2p+1]]
            s_w2[i] = (i & 1) ? (Co > 1 ? cfd_f2{w2[HEAD_HD + i - 1], w2[HEAD_HD + i]} : cfd_f2{0.f, 0.f}) : cfd_f2{w2[i], w2[i + 1]};
        }
    }
    const float b2v0 = b2[0], b2v1 = Co > 1 ? b2[1] : 0.f;
    __syncthreads();
    for (; tile < total; tile += stride) {
        float h[CQ][4], mk[4], lb[4];
#pragma unroll
        for (int c = 0; c < CQ; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[c][j] = hn[c][j];
        head_mask_raw<CQ, VEC4, TA>(h, b, C, HW, px, q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { mk[j] = mkn[j]; lb[j] = lbn[j]; }
        const int bc = b, pxc = px;
        locate(tile + stride, b, px);
        head_load_raw<CQ, VEC4, TA>(a, b, C, HW, px, q, hn);
        fetch_io(b, px);
        cfd_sched_fence();  // the prefetch stays here, ahead of this tile's arithmetic
        if constexpr (ACT) head_act<CQ>(h);
        // The 4 pixel phases j run in a ROLLED loop (one 16-pixel sub-tile per trip keeps the live set at one
        // z tile); the phase being processed always sits in h[c][0] / lands in out*[3], registers rotate each trip.
        float out0[4] = {0.f, 0.f, 0.f, 0.f}, out1[4] = {0.f, 0.f, 0.f, 0.f};
CFD_UNROLL(CFD_HF_UNROLL)
        for (int j = 0; j < 4; ++j) {
            const int lo = cfd_opaque(lane), q4 = 4 * cfd_opaque(q);  // keep the LDS table reads inside the loop (q4: a visible multiple of 4)
            float xk[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) xk[c] = c < CQ ? h[c < CQ ? c : 0][0] : 0.f;
            const CfdAct8<AP> bs = cfd_act_split8<AP>(xk);
            f32x4 z[HEAD_MT];
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt) {
                const int jb = 16 * mt + q4;
                const float4 bq = *reinterpret_cast<const float4*>(s_b1 + jb);  // one ds_read_b128 (16-byte aligned)
                z[mt] = f32x4{bq.x, bq.y, bq.z, bq.w};
            }
            // z += W1 h, both operands in AP pieces (2: w_lo*h_hi + w_hi*h_lo + w_hi*h_hi; 3: the six products of cfd_mfma_bf16x6),
            // strictly term-major: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int k = 0; k < cfd_nterm_aa(AP); ++k) {
                if (AP == 2 && k == 2) cfd_sched_fence();
#pragma unroll
                for (int mt = 0; mt < HEAD_MT; ++mt)
                    z[mt] = cfd_mfma16x16x32_bf16(s_w1[(cfd_term_aa_a(AP, k) * HEAD_MT + mt) * 64 + ((AP == 2 && k == 2) ? cfd_opaque(lo) : lo)],
                                                  bs.p[cfd_term_aa_b(AP, k)], z[mt]);
            }
            cfd_f2 ox = {0.f, 0.f}, oy = {0.f, 0.f};  // outputs 0 / 1, partial sums over the even / odd hidden units of this lane
#pragma unroll
            for (int mt = 0; mt < HEAD_MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const int jh = 16 * mt + q4 + r;
                    const cfd_f2 gl = cfd_gelu2(cfd_f2{z[mt][r], z[mt][r + 1]});
                    const float4 wq = *reinterpret_cast<const float4*>(s_w2 + jh);  // (w2[0][jh], w2[0][jh+1], w2[1][jh], w2[1][jh+1])
                    ox = cfd_fma2(cfd_f2{wq.x, wq.y}, gl, ox);
                    oy = cfd_fma2(cfd_f2{wq.z, wq.w}, gl, oy);
                }
            const float o0 = cfd_row_sum4(cfd_hsum2(ox)), o1 = cfd_row_sum4(cfd_hsum2(oy));  // sum over the four lane groups
            out0[0] = out0[1]; out0[1] = out0[2]; out0[2] = out0[3]; out0[3] = o0;
            out1[0] = out1[1]; out1[1] = out1[2]; out1[2] = out1[3]; out1[3] = o1;
#pragma unroll
            for (int c = 0; c < CQ; ++c) { h[c][0] = h[c][1]; h[c][1] = h[c][2]; h[c][2] = h[c][3]; }
        }
        if (q < Co) {  // lane group q stores output channel q
            const int c = q;
            const float bias = c == 0 ? b2v0 : b2v1;
            float pv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = pxc + j < HW;
                pv[j] = ((c == 0 ? out0[j] : out1[j]) + bias) * mk[j];  // fno2d.py:233
                if (label && ok) {
                    const float lab = lb[j] * mk[j];  // fno2d.py:236
                    const float d = pv[j] - lab;
                    lsq = fmaf(d, d, lsq);
                    labs += fabsf(d);
                    ll2 = fmaf(lab, lab, ll2);
                }
            }
            float* dst = preds + ((size_t)bc * Co + c) * HW + pxc;
            if constexpr (VEC4) {
                if (pxc < HW) *reinterpret_cast<float4*>(dst) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            } else if (pxc + 3 < HW) {
                cfd_st4u(dst, make_float4(pv[0], pv[1], pv[2], pv[3]));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (pxc + j < HW) dst[j] = pv[j];
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

#define HEAD_LOSS_ROWS 3  // rows of the one-pass head behind the parameter gradients: sum d^2, sum |d|, sum (label * mask)^2
static size_t head_part_floats(int C, int Co) { return (size_t)HEAD_HD * C + HEAD_HD + (size_t)Co * HEAD_HD + Co + HEAD_LOSS_ROWS; }
__device__ __forceinline__ size_t head_part_floats_dev(int C, int Co) { return (size_t)HEAD_HD * C + HEAD_HD + (size_t)Co * HEAD_HD + Co + HEAD_LOSS_ROWS; }
static_assert(HEAD_HD == CFD_HEAD_HD, "cfd_tail.h's head_reduce_group assumes the same hidden width");

extern "C" size_t cfd_fno_head_workspace_bytes(int B, int C, int Hd, int Co, int HW) {
    (void)Hd;
    if (B <= 0) return 0;
    const size_t fwd = (size_t)head_blocks(B, HW, false, 0) * 3 * sizeof(float);
    const size_t bwd = (size_t)head_bwd_blocks(B, HW, false) * head_part_floats(C, Co) * sizeof(float);
    return fwd > bwd ? fwd : bwd;
}

static int head_check(const char* fn, int B, int C, int Hd, int Co, int HW) {
    CFD_REQUIRE(B >= 0 && C >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "%s: bad sizes", fn);
    CFD_REQUIRE(Hd == HEAD_HD, CFD_ERR_UNSUPPORTED, "%s: head width %d unsupported (the reference hard-codes 128, fno2d.py:175)", fn, Hd);
    CFD_REQUIRE(C <= 32, CFD_ERR_UNSUPPORTED, "%s: hidden=%d (max 32) unsupported", fn, C);
    CFD_REQUIRE((long)B * ((HW + 63) / 64) < (1L << 30), CFD_ERR_UNSUPPORTED, "%s: B * ceil(HW / 64) = %ld tiles (max 2^30)", fn,
                (long)B * ((HW + 63) / 64));
    CFD_REQUIRE(Co >= 1 && Co <= 2, CFD_ERR_UNSUPPORTED, "%s: out_chan=%d (max 2) unsupported", fn, Co);
    return CFD_OK;
}

extern "C" int cfd_fno_head_fwd(const float* a, const float* mask, const float* label, const float* w1, const float* b1,
                                const float* w2, const float* b2, float* preds, float* sums, void* ws, int B, int C,
                                int Hd, int Co, int HW, int act_in, void* stream) {
    return cfd_int_fno_head_fwd(a, mask, label, w1, b1, w2, b2, preds, sums, ws, B, C, Hd, Co, HW, act_in, CFD_DT_F32, stream);
}

int cfd_int_fno_head_fwd(const void* a_, const float* mask, const float* label, const float* w1, const float* b1, const float* w2,
                         const float* b2, float* preds, float* sums, void* ws, int B, int C, int Hd, int Co, int HW, int act_in,
                         int dt, void* stream) {
    const float* a = (const float*)a_;
    CFD_REQUIRE(a && w1 && b1 && w2 && b2 && preds, CFD_ERR_INVALID_ARG, "cfd_fno_head_fwd: NULL pointer");
    CFD_REQUIRE(!label || (sums && ws), CFD_ERR_INVALID_ARG, "cfd_fno_head_fwd: label given without sums/workspace");
    CFD_TRY(head_check("cfd_fno_head_fwd", B, C, Hd, Co, HW));
    if (B == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = head_blocks(B, HW, true, C);
    float* part = label ? (float*)ws : nullptr;
    const bool v4 = dt == CFD_DT_F32 && HW % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)preds % 16) == 0;
    {
    CFD_PROF_W("k_head_fwd", st, B * HW * ((double)cfd_dt_size(dt) * C + 4.0 * (1 + (label ? 2 : 1) * Co)), 2.0 * B * HW * (double)HEAD_HD * (C + Co));
    const bool ap3 = dt == CFD_DT_F32 && cfd_act_pieces() == 3;  // fp32-exact-class route (bf16 storage keeps two pieces)
#define CFD_HF_P(Q_, V_, A_, T_, P_)                                                                                      \
    hipLaunchKernelGGL((k_head_fwd<Q_, V_, A_, T_, P_>), dim3(blocks), dim3(256), 0, st, (const T_*)a_, mask, label, w1, b1, w2, \
                       b2, preds, part, B, C, Co, HW)
#define CFD_HF(Q_, V_, A_, T_)                                                                        \
    do {                                                                                              \
        if constexpr (sizeof(T_) == 4) { if (ap3) CFD_HF_P(Q_, V_, A_, T_, 3); else CFD_HF_P(Q_, V_, A_, T_, 2); } \
        else CFD_HF_P(Q_, V_, A_, T_, 2);                                                             \
    } while (0)
#define CFD_HF_Q(Q_)                                         \
    do {                                                     \
        if (dt == CFD_DT_BF16) {                             \
            if (act_in) CFD_HF(Q_, false, true, __bf16);     \
            else CFD_HF(Q_, false, false, __bf16);           \
        } else if (v4) {                                     \
            if (act_in) CFD_HF(Q_, true, true, float);       \
            else CFD_HF(Q_, true, false, float);             \
        } else {                                             \
            if (act_in) CFD_HF(Q_, false, true, float);      \
            else CFD_HF(Q_, false, false, float);            \
        }                                                    \
    } while (0)
    if (C <= 8) CFD_HF_Q(2);  // channels per lane group of the K = 32 operand
    else if (C <= 20) CFD_HF_Q(5);
    else CFD_HF_Q(8);
#undef CFD_HF_Q
#undef CFD_HF
#undef CFD_HF_P
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
// derivative, the fc2 weight gradient and fc1 bias gradient (per-lane VALU accumulators), the fc1 weight gradient
// (MFMA, K = pixels) -- is wave-local, and all weight fragments of the slice are loop-invariant registers.
// The input tile h = f(a) is needed by every wave in two operand layouts, so the WORKGROUP stages it once per tile
// (load, GELU, bf16 hi/lo split: 1/4 of the per-wave cost, and no padding lanes) into double-buffered LDS planes one
// tile ahead, with the raw global loads issued two tiles ahead:
//   s_hk [pixel column][channel]  -> B operand of the recompute (one ds_read_b128 per hi / lo fragment)
//   s_ht [channel][pixel column]  -> B operand of the fc1 weight gradient
// (pixel 4n + j of the tile lives in column 16j + n: phase j of lane n).  All three GEMMs are split-bf16 3-term MFMAs, K = 32:
//   recompute   z[hidden][px]  = W1 h          K = channels
//   d/dh part   [channel][px]  = W1^T gz       K = this wave's 32 hidden units in ACCUMULATOR order: slot (q, v) <->
//               hidden 16(v/4) + 4q + v%4, so the gz registers feed the B operand without any data movement
//   gw1         [hidden][chan] += gz h^T       K = 32 pixels, after every second phase; gz^T goes through a
//               wave-private LDS plane pair (wave-level sync only)
// d/dh sums over hidden units: each wave contributes its partial through LDS once per tile and the workgroup
// finishes ga = d/dh * f'(a) with coalesced float4 stores.  Two workgroup barriers per tile.
// FUSE (training engine, round 2): the SAME pass also produces the forward results -- predictions and the loss sums -- so the
// hidden layer's GELU is evaluated once per step instead of twice (k_head_fwd + this kernel).  Possible because the
// derivative of mse / nmse / mae with respect to a prediction needs nothing of the forward pass but the prediction itself:
// its coefficient (1 / n or 1 / sum label^2) is known from the labels before the head runs (k_label_energy, cfd_loss_coef).
// Per pixel phase every wave adds its 32 hidden units' share of fc2 into LDS, one workgroup barrier later all four waves
// read the sum, form the prediction and its loss gradient, and continue with the backward arithmetic on the GELU terms
// they kept in registers.  `preds_w` / `b2` are used instead of `preds` / `gext`; the partial-sum block grows by
// (sum d^2, sum |d|).
// TA: storage type of the activations `a` (float; __bf16 in bf16-storage training, FUSE only).
// AP: bf16 pieces per operand value (2 = default; 3 = fp32-exact class, input planes single-buffered so that two workgroups still fit
// a CU -- the second buffer was never needed for correctness: a tile's planes are staged between the two tile-end barriers, after
// every wave has finished reading the previous tile's).
// NWV (round 5): waves per workgroup, 4 (32 hidden units = two M-tiles per wave) or 8 (16 hidden units each, eight partial d/dh planes).
// At 21 .. 32 channels the four-wave workgroup needs 97 KB of LDS -- ONE workgroup per CU, one wave per SIMD (VERDICT r4 missing #3);
// eight waves share the input planes of one 131-KB workgroup: two waves per SIMD, k_head_train 415 -> see DESIGN.md section 4.  At
// <= 20 channels two four-wave workgroups fit a CU and stay the default (the eight-wave form measured there: profiles/r05c_head_waves.txt;
// a first version that squeezed it into 80 KB by letting waves w and w + 4 ds_add_f32 into one partial plane ran 4.5x SLOWER: LDS float
// atomics serialise).
template <int KS, bool VEC4, bool ACT, bool FUSE, typename TA, int AP, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, 2) void k_head_bwd(
    const TA* __restrict__ a, const float* __restrict__ mask, const float* __restrict__ label,
    const float* __restrict__ preds, const float* __restrict__ gext, const float* __restrict__ coef,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float* __restrict__ ga,
    float* __restrict__ part, int B, int C, int Co, int HW, float* __restrict__ preds_w, const float* __restrict__ b2, float c0v, float c1v) {
    constexpr int CP = 4 * KS;          // padded channel count (rows of the d/dh exchange buffer)
    constexpr int MU = (CP + 15) / 16;  // 16-channel tiles
    constexpr int LDK = 40;             // bf16 row stride of s_hk: 32 channels + 8 pad (80 B: conflict-free b128 reads)
    constexpr int LDT = 72;             // bf16 row stride of s_ht: 64 pixel columns + 8 pad (144 B)
    constexpr int LDX = 40;             // bf16 row stride of the gz^T planes: 32 pixel columns + 8 pad
    constexpr int HT = 8 / NWV;         // 16-unit hidden tiles per wave (M-tiles of the recompute)
    constexpr int HPW = 16 * HT;        // hidden units per wave
    constexpr int NTH = 64 * NWV;       // threads
    static_assert(NWV == 4 || NWV == 8, "hidden units over four or eight waves");
    constexpr int NST = (CP * 16 + NTH - 1) / NTH;  // float4 staging loads per thread and tile
    constexpr int NBUF = AP == 3 ? 1 : 2;  // input-plane buffers (see above)
    __shared__ __attribute__((aligned(16))) __bf16 s_hk[NBUF][AP][64 * LDK];   // [buffer][piece][column][channel]
    constexpr int HTR = CP + 1;  // rows of an s_ht plane: the CP channels + ONE zero row that every padding channel of the 16-wide
                                 // MFMA tiles reads (16 MU rows would not leave room for two workgroups per CU in the fused kernel)
    __shared__ __attribute__((aligned(16))) __bf16 s_ht[NBUF][AP][HTR * LDT];  // [buffer][piece][channel][column]
    __shared__ __attribute__((aligned(16))) __bf16 s_x[NWV][AP][HPW * LDX];    // [wave][piece][hidden][32 columns]
    __shared__ float4 s_red[NWV * CP * 16];  // [wave][channel][16 x float4 = 64 pixels] partial d/dh
    __shared__ cfd_f2 s_gr[2][64];         // [buffer][column] upstream gradient on the raw head output, both channels
                                           // (FUSE: label * mask of both channels; the gradient is formed in the phase)
    __shared__ float s_mk[FUSE ? 2 : 1][64];       // FUSE: mask by column (0 past the end of the image / of the work)
    __shared__ cfd_f2 s_pp[FUSE ? 4 : 1][NWV][16]; // FUSE: [phase parity][wave][pixel lane] partial fc2 sums of both outputs
    __shared__ __attribute__((aligned(16))) float s_pv[2][FUSE ? 64 : 4];  // FUSE: the tile's predictions in pixel order (wave 0 only)
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    __bf16* s_xw = s_x[wave][0];  // piece pc of this wave's gz^T plane at s_xw + pc * 32 * LDX
    // ---- loop-invariant fragments of this wave's hidden slice ----
    CfdAct8<AP> w1f[HT];  // A operand of z = W1 h:        w1[HPW w + 16t + n][8q + v]
    CfdAct8<AP> w1t[MU];  // A operand of d/dh = W1^T gz:  w1[32w + 16(v/4) + 4q + v%4][16mu + n]
    float bz[HT][4];               // b1 at hidden unit HPW w + 16t + 4q + r
    cfd_f2 w2a[HT][2], w2b[HT][2]; // w2[0], w2[1] at hidden units HPW w + 16t + 4q + {2v, 2v+1}
    // loss-gradient coefficients: from device memory (cfd_loss_coef / cfd_label_energy_coef), or by value when coef == NULL (round 6: the
    // deferred nMSE normaliser -- the head runs with the mse coefficient and cfd_fno_adam_step rescales every gradient)
    const float c0 = label ? (coef ? coef[0] : c0v) : 0.f, c1 = label ? (coef ? coef[1] : c1v) : 0.f;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 aw1[HT][MU];
    cfd_f2 acc2a[HT][2], acc2b[HT][2], accb1[HT][2];
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
        for (int v = 0; v < MU; ++v) aw1[t][v] = zero;
#pragma unroll
        for (int v = 0; v < 2; ++v) { acc2a[t][v] = cfd_f2{0.f, 0.f}; acc2b[t][v] = cfd_f2{0.f, 0.f}; accb1[t][v] = cfd_f2{0.f, 0.f}; }
    }
    float gb2a0 = 0.f, gb2a1 = 0.f;
    float lsq = 0.f, labs = 0.f, lab2 = 0.f;                       // FUSE: loss sums and sum (label * mask)^2 (wave 0, lane group 0)
    const float b2v0 = FUSE ? b2[0] : 0.f, b2v1 = (FUSE && Co > 1) ? b2[1] : 0.f;
    const int tpb = (HW + 63) / 64;
    const int total = B * tpb;  // < 2^30 (checked by the launcher)
    // tile -> (batch entry, first pixel); b = -1 past the end of the work.  One division per trip of the tile loop: the
    // coordinates of the tile being processed / staged / fetched are carried in t0 / t1 / t2.
    struct TileAt { int b, px0; };
    auto locate = [&](int tile) -> TileAt {
        if (tile >= total) return TileAt{-1, 0};
        const unsigned ub = (unsigned)tile / (unsigned)tpb;
        return TileAt{(int)ub, (int)((unsigned)tile - ub * (unsigned)tpb) * 64};
    };
    // ---- cooperative staging: element e = (channel e/16, pixel quad e%16) of a tile ----
    float4 raw[NST];
    float4 gp_cur[NST], gp_next[NST];  // f'(a) of the tile being processed / of the tile just staged (ACT only): the GELU terms
                                       // are evaluated ONCE per activation value, for f(a) at staging and f'(a) in the epilogue
    // upstream gradient of the tile's 64 pixels: wave 1 owns it (lane = pixel), fetched / staged with the activations.
    // (The first version had every lane of every wave load mask / preds / label for its four pixels at the top of the
    // tile and wait for them: 16x redundant, ~700 instructions and one exposed memory latency per tile.)
    float rmk = 0.f, rpr[2] = {0.f, 0.f}, rlb[2] = {0.f, 0.f}, rge[2] = {0.f, 0.f};
    bool rok = false;
    auto fetch_gr = [&](const TileAt t) {
        if (wave != 1) return;
        const int pix = t.px0 + lane;
        rok = t.b >= 0 && pix < HW;
        rmk = 1.f;
        rpr[0] = rpr[1] = rlb[0] = rlb[1] = rge[0] = rge[1] = 0.f;
        if (rok) {
            if (mask) rmk = mask[(size_t)t.b * HW + pix];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < Co) {
                    const size_t off = ((size_t)t.b * Co + c) * HW + pix;
                    if constexpr (FUSE) {
                        rlb[c] = label[off];
                    } else {
                        if (gext) rge[c] = gext[off];
                        if (label) { rpr[c] = preds[off]; rlb[c] = label[off]; }
                    }
                }
        }
    };
    auto stage_gr = [&](int buf) {
        if (wave != 1) return;
        if constexpr (FUSE) {
            const int colf = 16 * (lane & 3) + (lane >> 2);
            s_gr[buf][colf] = rok ? cfd_f2{rlb[0] * rmk, rlb[1] * rmk} : cfd_f2{0.f, 0.f};  // fno2d.py:236
            s_mk[buf][colf] = rok ? rmk : 0.f;
            return;
        }
        float g[2] = {0.f, 0.f};
        if (rok) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < Co) {
                    float gp = rge[c];
                    if (label) {
                        const float d = rpr[c] - rlb[c] * rmk;
                        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                        gp += c0 * 2.f * d + c1 * sg;
                    }
                    g[c] = gp * rmk;
                }
        }
        gb2a0 += g[0];
        gb2a1 += g[1];
        s_gr[buf][16 * (lane & 3) + (lane >> 2)] = cfd_f2{g[0], g[1]};  // pixel 4n + j -> column 16j + n
    };
    auto fetch = [&](const TileAt t) {  // raw activations of the tile -> registers (zeros past the end of the work / image)
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = threadIdx.x + NTH * k;
            const int i = e >> 4, n4 = e & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t.b >= 0 && i < C) {
                const int b = t.b;
                const int p4 = t.px0 + 4 * n4;
                const TA* src = a + ((size_t)b * C + i) * HW + p4;
                if constexpr (VEC4) {
                    if (p4 < HW) v = cfd_ld4(src);
                } else if (sizeof(TA) == 4 && p4 + 3 < HW) {
                    v = cfd_ld4u(reinterpret_cast<const float*>(src));
                } else {
                    if (p4 < HW) v.x = cfd_ld(src);
                    if (p4 + 1 < HW) v.y = cfd_ld(src + 1);
                    if (p4 + 2 < HW) v.z = cfd_ld(src + 2);
                    if (p4 + 3 < HW) v.w = cfd_ld(src + 3);
                }
            }
            raw[k] = v;
        }
    };
    auto stage = [&](int buf) {  // registers -> f(a) -> bf16 hi/lo -> both operand layouts of buffer `buf`
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = threadIdx.x + NTH * k;
            const int i = e >> 4, n4 = e & 15;
            if (i < C) {
                float4 v = raw[k];
                if constexpr (ACT) {
                    cfd_f2 P0, e0, P1, e1;
                    const cfd_f2 x0 = {v.x, v.y}, x1 = {v.z, v.w};
                    cfd_gelu_terms2(x0, P0, e0);
                    cfd_gelu_terms2(x1, P1, e1);
                    const cfd_f2 g0 = cfd_fma2(x0 * (cfd_f2)(CFD_INV_SQRT_2PI), e0, P0), g1 = cfd_fma2(x1 * (cfd_f2)(CFD_INV_SQRT_2PI), e1, P1);
                    gp_next[k] = make_float4(g0.x, g0.y, g1.x, g1.y);
                    const cfd_f2 f0 = x0 * P0, f1 = x1 * P1;
                    v = make_float4(f0.x, f0.y, f1.x, f1.y);
                }
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = 16 * j + n4;
                    if constexpr (AP == 3) {  // truncating pieces: pc0 + pc1 + pc2 == the value exactly
                        float r = vv[j];
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            const unsigned hb = __builtin_bit_cast(unsigned, r) & 0xffff0000u;
                            const __bf16 pv = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
                            s_hk[buf][pc][col * LDK + i] = pv;
                            s_ht[buf][pc][i * LDT + col] = pv;
                            r -= __builtin_bit_cast(float, hb);
                        }
                    } else {
                        const __bf16 hi = (__bf16)vv[j];
                        const __bf16 lo = (__bf16)(vv[j] - (float)hi);
                        s_hk[buf][0][col * LDK + i] = hi;
                        s_hk[buf][1][col * LDK + i] = lo;
                        s_ht[buf][0][i * LDT + col] = hi;
                        s_ht[buf][1][i * LDT + col] = lo;
                    }
                }
            }
        }
    };
    // prologue: the first tile's global loads go out FIRST; the LDS planes are zeroed and the weight fragments loaded while
    // they are in flight (the first version did these one after the other: two exposed memory latencies per workgroup)
    TileAt t0 = locate((int)blockIdx.x), t1 = locate((int)blockIdx.x + (int)gridDim.x);
    fetch(t0);
    fetch_gr(t0);
    for (int i = threadIdx.x; i < NBUF * AP * 64 * LDK; i += blockDim.x) (&s_hk[0][0][0])[i] = (__bf16)0.f;
    for (int i = threadIdx.x; i < NBUF * AP * HTR * LDT; i += blockDim.x) (&s_ht[0][0][0])[i] = (__bf16)0.f;
    {
        float x[8];
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int v = 0; v < 8; ++v) x[v] = (8 * q + v < C) ? w1[(HPW * wave + 16 * t + n) * C + 8 * q + v] : 0.f;
            w1f[t] = cfd_act_split8<AP>(x);
        }
#pragma unroll
        for (int mu = 0; mu < MU; ++mu) {
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const int jh = HPW * wave + 16 * (v >> 2) + 4 * q + (v & 3);  // (NWV = 8: the k-slots v >= 4 carry zeros)
                x[v] = ((v >> 2) < HT && 16 * mu + n < C) ? w1[(jh < HEAD_HD ? jh : 0) * C + 16 * mu + n] : 0.f;
            }
            w1t[mu] = cfd_act_split8<AP>(x);
        }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jh = HPW * wave + 16 * t + 4 * q + r;
            bz[t][r] = b1[jh];
            w2a[t][r >> 1][r & 1] = w2[jh];
            w2b[t][r >> 1][r & 1] = Co > 1 ? w2[HEAD_HD + jh] : 0.f;
        }
    __syncthreads();  // planes zeroed
    stage(0);
    stage_gr(0);
#pragma unroll
    for (int k = 0; k < NST; ++k) gp_cur[k] = gp_next[k];
    fetch(t1);
    fetch_gr(t1);
    __syncthreads();
    int buf = 0;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x, buf ^= (NBUF - 1)) {  // all four waves walk the same tiles
        const int b = t0.b;
        const int px0 = t0.px0;
        // The 4 pixel phases run in a ROLLED loop, which keeps the live set at one phase.
        float* s_redf = reinterpret_cast<float*>(s_red);
        const cfd_f2* grp = s_gr[buf];
        const __bf16* hk = s_hk[buf][0];  // piece pc at + pc * 64 * LDK
        const __bf16* ht = s_ht[buf][0];  // piece pc at + pc * HTR * LDT
        auto front = [&](int j, f32x4 (&z)[HT]) {
            // 1. recompute this wave's slice of the hidden pre-activation z[hidden][pixel n]
            const int col = 16 * j + n;
            bf16x8 hp[AP];
#pragma unroll
            for (int pc = 0; pc < AP; ++pc) hp[pc] = *reinterpret_cast<const bf16x8*>(hk + pc * 64 * LDK + col * LDK + 8 * q);
#pragma unroll
            for (int t = 0; t < HT; ++t) z[t] = f32x4{bz[t][0], bz[t][1], bz[t][2], bz[t][3]};
#pragma unroll
            for (int k = 0; k < cfd_nterm_aa(AP); ++k)
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    if constexpr (CFD_HEAD_EXP & 8) z[t][k & 3] += (float)hp[cfd_term_aa_b(AP, k)][t];
                    else z[t] = cfd_mfma16x16x32_bf16(w1f[t].p[cfd_term_aa_a(AP, k)], hp[cfd_term_aa_b(AP, k)], z[t]);
                }
        };
        auto back = [&](int j, const float (&gzv)[8]) {
            const CfdAct8<AP> gs = cfd_act_split8<AP>(gzv);
            // 3. transposed gz planes of this wave: row = local hidden unit 16t + 4q + r, column 16(j&1) + n
            const int xcol = 16 * (j & 1) + n;
#pragma unroll
            for (int v = 0; v < 4 * HT; ++v) {
                const int row = 16 * (v >> 2) + 4 * q + (v & 3);
#pragma unroll
                for (int pc = 0; pc < AP; ++pc)
                    if (!(CFD_HEAD_EXP & 64) || (v == 0 && gzv[0] == 1.2345f)) s_xw[pc * HPW * LDX + row * LDX + xcol] = gs.p[pc][v];
            }
            // 4. partial d/dh[channel][pixel] = sum over this wave's hidden units w1[jh][channel] gz[jh][pixel]
            f32x4 ghc[MU];
#pragma unroll
            for (int mu = 0; mu < MU; ++mu) ghc[mu] = zero;
#pragma unroll
            for (int k = 0; k < cfd_nterm_aa(AP); ++k)
#pragma unroll
                for (int mu = 0; mu < MU; ++mu) {
                    if constexpr (CFD_HEAD_EXP & 16) ghc[mu][k & 3] += (float)gs.p[cfd_term_aa_b(AP, k)][mu];
                    else ghc[mu] = cfd_mfma16x16x32_bf16(w1t[mu].p[cfd_term_aa_a(AP, k)], gs.p[cfd_term_aa_b(AP, k)], ghc[mu]);
                }
            // this wave's partial d/dh of the tile: [channel][64 pixels], pixel = 4n + phase
#pragma unroll
            for (int mu = 0; mu < MU; ++mu)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * mu + 4 * q + r;
                    if (i < CP) s_redf[((wave * CP + i) * 16 + n) * 4 + j] = ghc[mu][r];
                }
            // 5. after each pair of phases: gw1[hidden][channel] += sum over 32 pixels gz[hidden][px] h[channel][px]
            if (j & 1) {
                cfd_wave_lds_sync();
                bf16x8 ap[HT][AP], bp[MU][AP];
#pragma unroll
                for (int t = 0; t < HT; ++t) {
                    const int o = (16 * t + n) * LDX + 8 * q;
#pragma unroll
                    for (int pc = 0; pc < AP; ++pc) ap[t][pc] = *reinterpret_cast<const bf16x8*>(s_xw + pc * HPW * LDX + o);
                }
#pragma unroll
                for (int mu = 0; mu < MU; ++mu) {
                    const int o = (16 * mu + n < CP ? 16 * mu + n : CP) * LDT + 16 * (j - 1) + 8 * q;
#pragma unroll
                    for (int pc = 0; pc < AP; ++pc) bp[mu][pc] = *reinterpret_cast<const bf16x8*>(ht + pc * HTR * LDT + o);
                }
#pragma unroll
                for (int k = 0; k < cfd_nterm_aa(AP); ++k)
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int mu = 0; mu < MU; ++mu) {
                            if constexpr (CFD_HEAD_EXP & 16) aw1[t][mu][k & 3] += (float)ap[t][cfd_term_aa_a(AP, k)][mu] * (float)bp[mu][cfd_term_aa_b(AP, k)][t];
                            else aw1[t][mu] = cfd_mfma16x16x32_bf16(ap[t][cfd_term_aa_a(AP, k)], bp[mu][cfd_term_aa_b(AP, k)], aw1[t][mu]);
                        }
                cfd_wave_lds_sync();
            }
        };
        if constexpr (FUSE) {
            // Two pixel phases per workgroup barrier: both phases' forward halves (GELU terms kept in registers, this wave's share
            // of fc2 into LDS), ONE barrier, then both backward halves.  Four slot pairs of s_pp rotate (pair parity x phase), so
            // a fast wave's next writes never meet a slow wave's reads.
            auto fwd_half = [&](int j, int slot, const f32x4 (&z)[HT], cfd_f2 (&a1k)[HT][2], cfd_f2 (&gdk)[HT][2]) {
                // forward half of the phase: GELU terms once, this wave's share of fc2, the workgroup's sum -> prediction,
                // loss terms and the gradient on the raw output; then the backward half on the kept terms
                cfd_f2 po0 = {0.f, 0.f}, po1 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const cfd_f2 zz = {z[t][2 * v], z[t][2 * v + 1]};
                        cfd_f2 Phi, e;
                        if constexpr (CFD_HEAD_EXP & 4) { Phi = zz; e = zz; } else cfd_gelu_terms2(zz, Phi, e);
                        a1k[t][v] = zz * Phi;
                        gdk[t][v] = cfd_fma2(zz * (cfd_f2)(CFD_INV_SQRT_2PI), e, Phi);
                        po0 = cfd_fma2(w2a[t][v], a1k[t][v], po0);
                        po1 = cfd_fma2(w2b[t][v], a1k[t][v], po1);
                    }
                const float p0 = cfd_row_sum4(cfd_hsum2(po0)), p1 = cfd_row_sum4(cfd_hsum2(po1));  // over the four lane groups
                if (q == 0) s_pp[slot][wave][n] = cfd_f2{p0, p1};
            };
            auto bwd_half = [&](int j, int slot, const cfd_f2 (&a1k)[HT][2], const cfd_f2 (&gdk)[HT][2], float (&gzv)[8]) {
                const int col = 16 * j + n;
                cfd_f2 sp = s_pp[slot][0][n];
#pragma unroll
                for (int wv = 1; wv < NWV; ++wv) sp = sp + s_pp[slot][wv][n];
                const float mk = s_mk[buf][col];
                const cfd_f2 lab = grp[col];
                const float pr0 = (sp.x + b2v0) * mk, pr1 = (sp.y + b2v1) * mk;  // fno2d.py:233
                const float d0 = pr0 - lab.x, d1 = pr1 - lab.y;
                float gp0 = c0 * 2.f * d0, gp1 = c0 * 2.f * d1;
                if (c1 != 0.f) {  // mae only (uniform)
                    gp0 += c1 * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
                    gp1 += c1 * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
                }
                gp0 *= mk;
                gp1 *= mk;
                if (wave == 0 && q == 0) {
                    s_pv[0][4 * n + j] = pr0;
                    s_pv[1][4 * n + j] = pr1;
                    lsq = fmaf(d0, d0, fmaf(d1, d1, lsq));
                    labs += fabsf(d0) + fabsf(d1);
                    lab2 = fmaf(lab.x, lab.x, fmaf(lab.y, lab.y, lab2));
                    gb2a0 += gp0;
                    gb2a1 += gp1;
                }
                const cfd_f2 g0 = (cfd_f2)(gp0), g1 = (cfd_f2)(gp1);
#pragma unroll
                for (int v = 4 * HT; v < 8; ++v) gzv[v] = 0.f;  // (NWV = 8: the upper k-slots of the d/dh product)
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        acc2a[t][v] = cfd_fma2(g0, a1k[t][v], acc2a[t][v]);
                        acc2b[t][v] = cfd_fma2(g1, a1k[t][v], acc2b[t][v]);
                        const cfd_f2 ga1 = cfd_fma2(w2a[t][v], g0, w2b[t][v] * g1);
                        const cfd_f2 gz = ga1 * gdk[t][v];
                        accb1[t][v] = accb1[t][v] + gz;
                        gzv[4 * t + 2 * v] = gz.x;
                        gzv[4 * t + 2 * v + 1] = gz.y;
                    }
            };
#pragma unroll 1
            for (int jp = 0; jp < 4; jp += 2) {
                cfd_f2 a1k0[HT][2], gdk0[HT][2], a1k1[HT][2], gdk1[HT][2];
                const int slot0 = jp, slot1 = jp + 1;
                {
                    f32x4 z[HT];
                    front(jp, z);
                    fwd_half(jp, slot0, z, a1k0, gdk0);
                }
                {
                    f32x4 z[HT];
                    front(jp + 1, z);
                    fwd_half(jp + 1, slot1, z, a1k1, gdk1);
                }
                if (!(CFD_HEAD_EXP & 1)) __syncthreads();
                {
                    float gzv[8];
                    bwd_half(jp, slot0, a1k0, gdk0, gzv);
                    back(jp, gzv);
                }
                {
                    float gzv[8];
                    bwd_half(jp + 1, slot1, a1k1, gdk1, gzv);
                    back(jp + 1, gzv);
                }
            }
        } else {
CFD_UNROLL(CFD_HB_UNROLL)
            for (int j = 0; j < 4; ++j) {
                f32x4 z[HT];
                front(j, z);
                const int col = 16 * j + n;
                // 2. a1 = gelu(z) feeds the fc2 weight gradient; gz = (W2^T graw) * gelu'(z), k-slot 4t + r
                float gzv[8];
            const cfd_f2 grj = grp[col];
            const cfd_f2 g0 = (cfd_f2)(grj.x), g1 = (cfd_f2)(grj.y);
#pragma unroll
            for (int v = 4 * HT; v < 8; ++v) gzv[v] = 0.f;
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int v = 0; v < 2; ++v) {  // packed pairs of hidden units r = 2v, 2v+1
                    const cfd_f2 zz = {z[t][2 * v], z[t][2 * v + 1]};
                    cfd_f2 Phi, e;
                    cfd_gelu_terms2(zz, Phi, e);
                    const cfd_f2 a1 = zz * Phi;
                    acc2a[t][v] = cfd_fma2(g0, a1, acc2a[t][v]);
                    acc2b[t][v] = cfd_fma2(g1, a1, acc2b[t][v]);
                    const cfd_f2 ga1 = cfd_fma2(w2a[t][v], g0, w2b[t][v] * g1);
                    const cfd_f2 gz = ga1 * cfd_fma2(zz * (cfd_f2)(CFD_INV_SQRT_2PI), e, Phi);
                    accb1[t][v] = accb1[t][v] + gz;
                    gzv[4 * t + 2 * v] = gz.x;
                    gzv[4 * t + 2 * v + 1] = gz.y;
                }
                back(j, gzv);
            }
        }
        if constexpr (FUSE) {
            if (wave == 0) cfd_wave_lds_sync();  // s_pv was written by this wave's lane group 0
            if (wave == 0 && q < Co) {  // predictions of the tile: lane group q stores channel q, pixels px0 + 4n .. 4n+3
                const int p4 = px0 + 4 * n;
                const float4 pv = *reinterpret_cast<const float4*>(&s_pv[q][4 * n]);
                float* dstp = preds_w + ((size_t)b * Co + q) * HW + p4;
                if constexpr (VEC4) {
                    if (p4 < HW) *reinterpret_cast<float4*>(dstp) = pv;
                } else if (p4 + 3 < HW) {
                    cfd_st4u(dstp, pv);
                } else {
                    const float pvv[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (p4 + jj < HW) dstp[jj] = pvv[jj];
                }
            }
        }
        if (!(CFD_HEAD_EXP & 2)) __syncthreads();
        // next tile's input planes (other buffer; its raw loads were issued one tile ago), then loads two tiles ahead
        stage(buf ^ (NBUF - 1));  // the other buffer -- or, single-buffered, the planes every wave has finished reading (barrier above)
        stage_gr(buf ^ (NBUF - 1));
        const TileAt t2 = locate(tile + 2 * (int)gridDim.x);
        fetch(t2);
        fetch_gr(t2);
        // ga[b][i][px0 .. px0+63] = (sum over the four hidden slices) * f'(a)
#pragma unroll
        for (int k = 0; k < NST; ++k) {
            const int e = threadIdx.x + NTH * k;
            if (e >= C * 16) continue;
            const int i = e >> 4, n4 = e & 15;
            float4 v = s_red[i * 16 + n4];
#pragma unroll
            for (int wv = 1; wv < NWV; ++wv) {
                const float4 u = s_red[(wv * CP + i) * 16 + n4];
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const int p4 = px0 + 4 * n4;
            const size_t off = ((size_t)b * C + i) * HW + p4;
            if constexpr (VEC4) {
                if (p4 < HW) {
                    if constexpr (ACT) {
                        const float4 gg = gp_cur[k];
                        v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
                    }
                    *reinterpret_cast<float4*>(ga + off) = v;
                }
            } else if (p4 + 3 < HW) {
                if constexpr (ACT) {
                    const float4 gg = gp_cur[k];
                    v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
                }
                cfd_st4u(ga + off, v);
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
                const float gv[4] = {gp_cur[k].x, gp_cur[k].y, gp_cur[k].z, gp_cur[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p4 + j < HW) {
                        float x = vv[j];
                        if constexpr (ACT) x *= gv[j];
                        ga[off + j] = x;
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < NST; ++k) gp_cur[k] = gp_next[k];
        t0 = t1;
        t1 = t2;
        if (!(CFD_HEAD_EXP & 2)) __syncthreads();
    }
    // ---- this block's partial parameter gradients: [gw1 128*C | gb1 128 | gw2 Co*128 | gb2 Co] ----
    const int o_gb1 = HEAD_HD * C, o_gw2 = o_gb1 + HEAD_HD, o_gb2 = o_gw2 + Co * HEAD_HD;
    // block-major partials part[block][element] (round 5): a wave's stores are 64-byte runs of consecutive elements.  (Rounds 1-4 wrote
    // them element-major, stride gridDim.x: 2948 x 512 scattered 4-byte stores = 1.5 M write transactions per launch, the kernel's
    // counter-based write traffic was 1.58x its algorithmic bytes -- profiles/r04z_pmc_traffic.json; k_head_reduce walks the blocks.)
    float* dst = part + (size_t)blockIdx.x * (head_part_floats_dev(C, Co));
    const size_t nb = 1;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jh = HPW * wave + 16 * t + 4 * q + r;
#pragma unroll
            for (int v = 0; v < MU; ++v) {
                const int i = 16 * v + n;
                if (i < C) dst[(size_t)(jh * C + i) * nb] = aw1[t][v][r];
            }
            // sums over the 16 pixel lanes of this q group
            float sa = acc2a[t][r >> 1][r & 1], sb = acc2b[t][r >> 1][r & 1], s1 = accb1[t][r >> 1][r & 1];
#pragma unroll
            for (int m = 1; m <= 8; m <<= 1) { sa += cfd_shfl_xor(sa, m); sb += cfd_shfl_xor(sb, m); s1 += cfd_shfl_xor(s1, m); }
            if (n == 0) {
                dst[(size_t)(o_gb1 + jh) * nb] = s1;
                dst[(size_t)(o_gw2 + jh) * nb] = sa;
                if (Co > 1) dst[(size_t)(o_gw2 + HEAD_HD + jh) * nb] = sb;
            }
        }
    if (wave == (FUSE ? 0 : 1)) {  // the wave that formed / staged the upstream gradients summed them (tiles past the end: zeros)
        gb2a0 = cfd_wave_sum(gb2a0);
        gb2a1 = cfd_wave_sum(gb2a1);
        if (lane == 0) {
            dst[(size_t)o_gb2 * nb] = gb2a0;
            if (Co > 1) dst[(size_t)(o_gb2 + 1) * nb] = gb2a1;
        }
        if constexpr (FUSE) {  // three more rows of the partial-sum block: sum d^2, sum |d|, sum (label*mask)^2  (MseLoss, loss.py:27-35)
            lsq = cfd_wave_sum(lsq);
            labs = cfd_wave_sum(labs);
            lab2 = cfd_wave_sum(lab2);
            if (lane == 0) {
                dst[(size_t)(o_gb2 + Co) * nb] = lsq;
                dst[(size_t)(o_gb2 + Co + 1) * nb] = labs;
                dst[(size_t)(o_gb2 + Co + 2) * nb] = lab2;
            }
        }
    }
}

// Sum of the per-block partial records part[block][PSTR] (block-major): workgroup = 16 consecutive elements (cfd_record_sum16: every
// load instruction reads 64-byte runs, fixed summation order, deterministic).
__global__ __launch_bounds__(256) void k_head_reduce(const HeadTail t) {
    __shared__ float s_scr[64];
    head_reduce_group(blockIdx.x, t, s_scr);
}

HeadTail cfd_int_head_tail(const void* ws, float* gw1, float* gb1, float* gw2, float* gb2, float* sums, int B, int C, int Co, int HW, float count) {
    const int pstr = (int)head_part_floats(C, Co);
    // rows reduced: parameter gradients; + sum d^2, sum |d| for the one-pass head (sums != NULL); + sum (label*mask)^2 when count > 0
    const int ps = pstr - HEAD_LOSS_ROWS + (sums ? (count > 0.f ? 3 : 2) : 0);
    return HeadTail{(const float*)ws, gw1, gb1, gw2, gb2, sums, head_bwd_blocks(B, HW), ps, pstr, C, Co, count};
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
    CFD_PROF_W("k_head_bwd", st, 4.0 * B * HW * (2.0 * C + 1 + 2.0 * Co), 2.0 * B * HW * (double)HEAD_HD * (3.0 * C + 2.0 * Co));
    const bool ap3 = cfd_act_pieces() == 3;
#define CFD_HB_P(K_, V_, A_, P_)                                                                                   \
    hipLaunchKernelGGL((k_head_bwd<K_, V_, A_, false, float, P_>), dim3(blocks), dim3(256), 0, st, a, mask, label, preds, gpreds_ext, \
                       coef, w1, b1, w2, ga, part, B, C, Co, HW, (float*)nullptr, (const float*)nullptr, 0.f, 0.f)
#define CFD_HB(K_, V_, A_) do { if (ap3) CFD_HB_P(K_, V_, A_, 3); else CFD_HB_P(K_, V_, A_, 2); } while (0)
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
#undef CFD_HB_P
    }
    CFD_LAUNCH_CHECK("cfd_fno_head_bwd");
    const HeadTail ht = cfd_int_head_tail(part, gw1, gb1, gw2, gb2, nullptr, B, C, Co, HW, 0.f);  // (the loss rows belong to the one-pass kernel)
    CFD_PROF_W("k_head_reduce", st, 0.0, 0.0);
    hipLaunchKernelGGL(k_head_reduce, dim3((ht.PS + 15) / 16), dim3(256), 0, st, ht);
    CFD_LAUNCH_CHECK("cfd_fno_head_bwd(reduce)");
    return CFD_OK;
}

// Training head in ONE pass (k_head_bwd<FUSE>): predictions, loss sums (sums[0] = sum d^2, sums[1] = sum |d|; sums[2], sums[3] and
// `coef` must already hold sum label^2, the element count and the loss-gradient coefficients -- cfd_label_energy +
// cfd_loss_coef), d loss / d a and the four parameter gradients.  Replaces cfd_fno_head_fwd + cfd_loss_coef + cfd_fno_head_bwd
// of a training step whose loss is one of mse / nmse / mae (fno2d.py:228-237, loss.py:22-37, train_auto.py:255).
extern "C" int cfd_fno_head_train(const float* a, const float* mask, const float* label, const float* coef, const float* w1,
                                  const float* b1, const float* w2, const float* b2, float* preds, float* sums, float* ga,
                                  float* gw1, float* gb1, float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co,
                                  int HW, int act_in, void* stream) {
    return cfd_int_fno_head_train(a, mask, label, coef, w1, b1, w2, b2, preds, sums, ga, gw1, gb1, gw2, gb2, ws, B, C, Hd, Co, HW, act_in,
                                  CFD_DT_F32, stream);
}

// `a` stored as dt (CFD_DT_BF16: bf16-storage training); everything else fp32
int cfd_int_fno_head_train(const void* a, const float* mask, const float* label, const float* coef, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* preds, float* sums, float* ga, float* gw1, float* gb1,
                           float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co, int HW, int act_in, int dt, void* stream) {
    CFD_REQUIRE(coef, CFD_ERR_INVALID_ARG, "cfd_fno_head_train: NULL pointer");
    return cfd_int_fno_head_train_f(a, mask, label, coef, 0.f, 0.f, 0.f, w1, b1, w2, b2, preds, sums, ga, gw1, gb1, gw2, gb2, ws, B, C, Hd, Co, HW,
                                    act_in, dt, stream, nullptr);
}

int cfd_int_fno_head_train_f(const void* a, const float* mask, const float* label, const float* coef, float c0v, float c1v, float count,
                             const float* w1, const float* b1, const float* w2, const float* b2, float* preds, float* sums, float* ga,
                             float* gw1, float* gb1, float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co, int HW, int act_in,
                             int dt, void* stream, HeadTail* defer) {
    if (defer) defer->part = nullptr;
    CFD_REQUIRE(a && label && w1 && b1 && w2 && b2 && preds && sums && ga && gw1 && gb1 && gw2 && gb2 && ws,
                CFD_ERR_INVALID_ARG, "cfd_fno_head_train: NULL pointer");
    CFD_TRY(head_check("cfd_fno_head_train", B, C, Hd, Co, HW));
    CFD_REQUIRE(B >= 1, CFD_ERR_INVALID_ARG, "cfd_fno_head_train: empty batch");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = head_bwd_blocks(B, HW);
    float* part = (float*)ws;
    const bool v4 = HW % 4 == 0 && ((uintptr_t)a % 16) == 0 && ((uintptr_t)ga % 16) == 0 && ((uintptr_t)preds % 16) == 0;
    {
    CFD_PROF_W("k_head_train", st, B * HW * ((4.0 + cfd_dt_size(dt)) * C + 4.0 + 8.0 * Co), 2.0 * B * HW * (double)HEAD_HD * (3.0 * C + 3.0 * Co));
    const bool ap3 = dt == CFD_DT_F32 && cfd_act_pieces() == 3;
    // eight waves of 16 hidden units (k_head_bwd: NWV) where the four-wave workgroup leaves a CU with one wave per SIMD (21 .. 32 channels);
    // "head_waves" = 4 / 8 forces either form
    const int hw = cfd_tune_get(CFD_TUNE_HEAD_WAVES);
    const bool w8 = dt == CFD_DT_F32 && (hw == 8 || (hw != 4 && C > 20));
#define CFD_HT_P(K_, V_, A_, T_, P_)                                                                                      \
    do {                                                                                                                  \
        if constexpr (sizeof(T_) == 4) {                                                                                  \
            if (w8) {                                                                                                     \
                hipLaunchKernelGGL((k_head_bwd<K_, V_, A_, true, T_, P_, 8>), dim3(blocks), dim3(512), 0, st, (const T_*)a, mask, label, (const float*)nullptr, \
                                   (const float*)nullptr, coef, w1, b1, w2, ga, part, B, C, Co, HW, preds, b2, c0v, c1v); \
                break;                                                                                                    \
            }                                                                                                             \
        }                                                                                                                 \
        hipLaunchKernelGGL((k_head_bwd<K_, V_, A_, true, T_, P_>), dim3(blocks), dim3(256), 0, st, (const T_*)a, mask, label, (const float*)nullptr, \
                           (const float*)nullptr, coef, w1, b1, w2, ga, part, B, C, Co, HW, preds, b2, c0v, c1v);         \
    } while (0)
#define CFD_HT(K_, V_, A_, T_)                                                                        \
    do {                                                                                              \
        if constexpr (sizeof(T_) == 4) { if (ap3) CFD_HT_P(K_, V_, A_, T_, 3); else CFD_HT_P(K_, V_, A_, T_, 2); } \
        else CFD_HT_P(K_, V_, A_, T_, 2);                                                             \
    } while (0)
#define CFD_HT_VA(K_)                                                     \
    do {                                                                  \
        if (dt == CFD_DT_BF16) {                                          \
            if (v4 && act_in) CFD_HT(K_, true, true, __bf16);             \
            else if (v4) CFD_HT(K_, true, false, __bf16);                 \
            else if (act_in) CFD_HT(K_, false, true, __bf16);             \
            else CFD_HT(K_, false, false, __bf16);                        \
        } else if (v4 && act_in) CFD_HT(K_, true, true, float);           \
        else if (v4) CFD_HT(K_, true, false, float);                      \
        else if (act_in) CFD_HT(K_, false, true, float);                  \
        else CFD_HT(K_, false, false, float);                             \
    } while (0)
    if (C <= 8) CFD_HT_VA(2);
    else if (C <= 20) CFD_HT_VA(5);
    else CFD_HT_VA(8);
#undef CFD_HT_VA
#undef CFD_HT
#undef CFD_HT_P
    }
    CFD_LAUNCH_CHECK("cfd_fno_head_train");
    const HeadTail ht = cfd_int_head_tail(part, gw1, gb1, gw2, gb2, sums, B, C, Co, HW, count);
    if (defer) {  // the caller's next carrying launch reduces the records (cfd_tail.h)
        *defer = ht;
        return CFD_OK;
    }
    CFD_PROF_W("k_head_reduce", st, 0.0, 0.0);
    hipLaunchKernelGGL(k_head_reduce, dim3((ht.PS + 15) / 16), dim3(256), 0, st, ht);
    CFD_LAUNCH_CHECK("cfd_fno_head_train(reduce)");
    return CFD_OK;
}
