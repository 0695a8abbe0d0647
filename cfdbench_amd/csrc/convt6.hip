// ConvTranspose2d(kernel 2, stride 2) of the U-Net's up path (src/models/unet.py:80) as GEMMs on the bf16 matrix pipe with
// THREE-PIECE operands (cfd_mfma_bf16x6: six v_mfma_f32_16x16x32_bf16 per product, fp32-exact class like conv6.hip).  With the
// weights read as the matrix Wm[i][c], c = 4 o + 2 dy + dx (their memory order), a pixel p = (b, y, x) of the INPUT grid and its
// 2 x 2 patch of the output grid:
//   forward         out[b][o][2y+dy][2x+dx] = bias[o] + sum_i Wm[i][c] in[b][i][p]          M = c (4 Co), N = pixels, K = i
//   input gradient  gin[b][i][p]            = sum_c Wm[i][c] gout[b][o][2y+dy][2x+dx]       M = i (Ci),   N = pixels, K = c
//   weight gradient gw[i][c]                = sum_p in[b][i][p] gout[b][o][2y+dy][2x+dx]    M = i, N = c, K = pixels
// The layers are small (1.2 GFLOP per pass over all four levels of the configs[2] U-Net: ~1 us of matrix-pipe time) -- what the
// kernels are built for is streaming: every operand goes global -> registers -> split -> MFMA, no LDS staging, no barriers in the
// loops, lanes n = consecutive pixels so that loads and the 2 x 2 patch stores (one float2 per output row) are contiguous runs.
// (The VALU kernels of conv.hip they replace took 32 / 22 us per level whatever its size: one thread per pixel looping over the
// channels with a broadcast LDS weight read per four FMAs -- latency-bound at the deep levels, LDS-issue-bound at the wide ones.)
#include "cfd_conv.h"

typedef cfd_u32x4 u4;

// ------------------------------------------------------------------------------------------------------
// forward (BWD = false) and input gradient (BWD = true).  A wave owns NT tiles of 16 consecutive pixels and MT tiles of 16 rows
// (blockIdx.y picks the row group); the K loop runs over 32-wide steps.  MFMA view: A row n of tile mt = matrix row 16 (mb + mt) + n
// with K slots 8 q .. 8 q + 7; B column n = pixel n of the tile; D[r] = row 4 q + r, column n.
// forward rows are c = 4 o + 2 dy + dx: a lane's four D registers are the 2 x 2 output patch of channel o = 4 (mb + mt) + q.
// ------------------------------------------------------------------------------------------------------
// KSPL (NT = 1): the levels with few pixels.  All four waves of a workgroup share ONE pixel tile and row group and take the K steps
// in turn; their partial tiles are summed through LDS in wave order.  (Without it the 2048-pixel level of the U-Net ran 128
// workgroups of 12 serial steps, one exposed memory round trip each: 35 us.)
template <bool BWD, int MT, int NT, bool KSPL>
__global__ __launch_bounds__(256) void k_convt6(const float* __restrict__ src, const float* __restrict__ w,
                                                const float* __restrict__ bias, float* __restrict__ dst, int Ci, int Co, int H, int W,
                                                unsigned total, CfdDiv dHW, CfdDiv dW, unsigned fbs) {
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int HW = H * W, C4 = 4 * Co;
    const int M = BWD ? Ci : C4, K = BWD ? C4 : Ci;
    const int mb = blockIdx.y * MT;
    static_assert(!KSPL || NT == 1, "the K split is built for single pixel tiles");
    __shared__ float s_red[KSPL ? 4 * MT * 256 : 1];
    const unsigned P0 = KSPL ? blockIdx.x * 16u : (blockIdx.x * 4u + wave) * (16u * NT);
    if (!KSPL && P0 >= total) return;  // (whole wave; the split form has a barrier below and no empty workgroups)
    // this lane's pixel of each tile
    unsigned sbase[NT];  // forward: element offset of in[b][0][p]; input gradient: of gout[b][0][2y][2x]
    unsigned dbase[NT];  // forward: of out[b][0][2y][2x]; input gradient: of gin[b][0][p]
    bool live[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const unsigned p = P0 + 16u * u + n;
        live[u] = p < total;
        const unsigned pc = live[u] ? p : total - 1;
        const unsigned b = cfd_div(pc, dHW), rem = pc - b * (unsigned)HW, y = cfd_div(rem, dW), x = rem - y * (unsigned)W;
        const unsigned fine = b * fbs + (2 * y) * (unsigned)(2 * W) + 2 * x,  // fbs: batch stride of the 2H x 2W tensor (>= Co 4 HW)
                        coarse = b * (unsigned)(Ci * HW) + rem;
        sbase[u] = BWD ? fine : coarse;
        dbase[u] = BWD ? coarse : fine;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float bv = 0.f;
        if constexpr (!BWD) {
            const int o = 4 * (mb + mt) + q;
            if (bias && (!KSPL || wave == 0)) bv = o < Co ? cfd_opaque_f(bias[o < Co ? o : Co - 1]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[mt][u] = f32x4{bv, bv, bv, bv};
    }
    const int ksteps = (K + 31) / 32;
#pragma unroll 1
    for (int ks = KSPL ? wave : 0; ks < ksteps; ks += KSPL ? 4 : 1) {
        const int k0 = 32 * ks + 8 * q;
        // All global loads of the step are issued back to back from clamped (always valid) addresses, made opaque, and only THEN
        // zeroed where out of range: `in_range ? load : 0` is compiled to a branch around each load with a full wait inside
        // (the first version: 24 serialized L2 round trips per step, 39 us for the 2048-pixel level).
        float bvv[NT][8];
        if constexpr (!BWD) {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = k0 + j;
                    bvv[u][j] = src[(size_t)sbase[u] + (size_t)(c < Ci ? c : Ci - 1) * HW];
                }
        } else {
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // K slots 4 h .. 4 h + 3 = the 2 x 2 patch of output channel k0 / 4 + h
                    const int o = (k0 >> 2) + h;
                    const float* gp = src + (size_t)sbase[u] + (size_t)(o < Co ? o : Co - 1) * (4 * HW);
                    const float2 g0 = *reinterpret_cast<const float2*>(gp), g1 = *reinterpret_cast<const float2*>(gp + 2 * W);
                    bvv[u][4 * h] = g0.x, bvv[u][4 * h + 1] = g0.y, bvv[u][4 * h + 2] = g1.x, bvv[u][4 * h + 3] = g1.y;
                }
        }
        float avv[MT][8];
        f32x4 av4[BWD ? MT : 1][2];  // input gradient: a row of Wm is contiguous in K (= 4 Co values, whole float4 groups)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * (mb + mt) + n, mc = m < M ? m : M - 1;
            if constexpr (BWD) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + 4 * h;
                    av4[mt][h] = *reinterpret_cast<const f32x4*>(w + (size_t)mc * C4 + (k < K ? k : K - 4));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = k0 + j, kc = k < K ? k : K - 1;
                    avv[mt][j] = w[(size_t)kc * C4 + mc];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) bvv[u][j] = cfd_opaque_f(bvv[u][j]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (BWD) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    av4[mt][h] = cfd_opaque_f4(av4[mt][h]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) avv[mt][4 * h + e] = av4[mt][h][e];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) avv[mt][j] = cfd_opaque_f(avv[mt][j]);
            }
        }
        bf16x8 bs[NT][3];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bvv[u][j] = k0 + j < K ? bvv[u][j] : 0.f;
            const CfdSplit8x3 s = cfd_split8x3(bvv[u]);
            bs[u][0] = s.p[0], bs[u][1] = s.p[1], bs[u][2] = s.p[2];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (cfd_uniform(16 * (mb + mt)) >= M) continue;
            const bool mok = 16 * (mb + mt) + n < M;
#pragma unroll
            for (int j = 0; j < 8; ++j) avv[mt][j] = (mok && k0 + j < K) ? avv[mt][j] : 0.f;
            const CfdSplit8x3 s = cfd_split8x3(avv[mt]);
            const bf16x8 as[3] = {s.p[0], s.p[1], s.p[2]};
            cfd_mfma_bf16x6_n<NT>(as, bs, acc[mt]);
        }
    }
    if constexpr (KSPL) {  // wave v < MT leaves with the sum of tile v, the others are done
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * MT + mt) * 4 + r) * 64 + lane] = acc[mt][0][r];
        __syncthreads();
        if (wave >= MT) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* sp = s_red + (wave * 4 + r) * 64 + lane;
            const float tot = (sp[0] + sp[MT * 256]) + (sp[2 * MT * 256] + sp[3 * MT * 256]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][0][r] = tot;  // (only tile `wave` is stored below)
        }
    }
    // ---- stores ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (KSPL && mt != wave) continue;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (!live[u]) continue;
            if constexpr (!BWD) {
                const int o = 4 * (mb + mt) + q;
                if (o < Co) {
                    float* op = dst + (size_t)dbase[u] + (size_t)o * (4 * HW);
                    *reinterpret_cast<float2*>(op) = make_float2(acc[mt][u][0], acc[mt][u][1]);
                    *reinterpret_cast<float2*>(op + 2 * W) = make_float2(acc[mt][u][2], acc[mt][u][3]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * (mb + mt) + 4 * q + r;
                    if (i < Ci) dst[(size_t)dbase[u] + (size_t)i * HW] = acc[mt][u][r];
                }
            }
        }
    }
}

// fine_bstride: elements between consecutive images of the 2H x 2W tensor (0 = Co 4 H W: dense); a larger stride addresses the
// tensor as a channel slice of a wider one (the U-Net's skip concatenation, unet.py:80-88)
static bool convt6_small(int B, int Ci, int Co, int H, int W, long fine_bstride = 0) {  // 32-bit element offsets
    const long fbs = fine_bstride > 0 ? fine_bstride : (long)Co * 4 * H * W;
    return (long)B * Ci * H * W < (1L << 31) && (long)B * fbs < (1L << 31) && fbs >= (long)Co * 4 * H * W;
}

bool cfd_convt6_covers(int B, int Ci, int Co, int H, int W, long fine_bstride) { return B >= 1 && convt6_small(B, Ci, Co, H, W, fine_bstride); }

template <bool BWD>
static int convt6_launch(const float* src, const float* w, const float* bias, float* dst, int B, int Ci, int Co, int H, int W,
                         long fine_bstride, hipStream_t st, const char* what) {
    constexpr int MT = 3;
    const unsigned total = (unsigned)((long)B * H * W);
    const int M = BWD ? Ci : 4 * Co, mgroups = ((M + 15) / 16 + MT - 1) / MT;
    // four pixel tiles per wave (one weight fragment feeds four MFMA chains) where that still leaves >= 512 workgroups
    // (the CPU emulator build sets the threshold to 2 so that small test shapes reach both forms)
#ifndef CFD_CONVT6_NT4_MIN_WGS
#define CFD_CONVT6_NT4_MIN_WGS 512
#endif
    const long wg4 = ((long)total + 255) / 256 * mgroups;
    const CfdDiv dHW = cfd_div_make((unsigned)(H * W)), dW = cfd_div_make((unsigned)W);
    if (wg4 >= CFD_CONVT6_NT4_MIN_WGS) {
        const dim3 grid((unsigned)(((long)total + 255) / 256), mgroups);
        hipLaunchKernelGGL((k_convt6<BWD, MT, 4, false>), grid, dim3(256), 0, st, src, w, bias, dst, Ci, Co, H, W, total, dHW, dW,
                           (unsigned)fine_bstride);
    } else {  // one 16-pixel tile per workgroup, the K steps dealt to its four waves
        const dim3 grid((unsigned)(((long)total + 15) / 16), mgroups);
        hipLaunchKernelGGL((k_convt6<BWD, MT, 1, true>), grid, dim3(256), 0, st, src, w, bias, dst, Ci, Co, H, W, total, dHW, dW,
                           (unsigned)fine_bstride);
    }
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

int cfd_convt6_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int H, int W, hipStream_t st,
                   const char* what, long out_bstride) {
    if (!cfd_convt6_covers(B, Ci, Co, H, W, out_bstride)) return CFD_ERR_UNSUPPORTED;
    return convt6_launch<false>(in, w, bias, out, B, Ci, Co, H, W, out_bstride > 0 ? out_bstride : (long)Co * 4 * H * W, st, what);
}

int cfd_convt6_bwd_in(const float* gout, const float* w, float* gin, int B, int Ci, int Co, int H, int W, hipStream_t st,
                      const char* what, long gout_bstride) {
    if (!cfd_convt6_covers(B, Ci, Co, H, W, gout_bstride)) return CFD_ERR_UNSUPPORTED;
    return convt6_launch<true>(gout, w, nullptr, gin, B, Ci, Co, H, W, gout_bstride > 0 ? gout_bstride : (long)Co * 4 * H * W, st, what);
}

// ------------------------------------------------------------------------------------------------------
// weight gradient (+ bias gradient): gw[i][c] = sum_p in[b][i][p] gout[b][o][2y+dy][2x+dx],  gb[o] = sum gout[b][o][..]
// K = pixels, 32 per step: lane (q, n) holds pixels P + 8 q .. + 7 of one image -- two runs of four neighbours in a row each
// (W % 4 == 0, H W % 8 == 0), so the A operand (row n = input channel) is two aligned float4 loads and the B operand (column n =
// (o, dy, dx)) the even or odd elements of two pairs of them.
// Workgroup (x, y, z) = k-step range x, row group y (MTG tiles of 16 channels), column group z (NTG tiles of 16 columns = 4 output
// channels each); its four waves take the steps of the range in turn, are summed through LDS in a fixed order and leave ONE
// partial block in slice x of the workspace: [Ci][4 Co] weights | [Co] bias sums (row group 0 only), reduced by k_part_reduce.
// ------------------------------------------------------------------------------------------------------
template <int MTG, int NTG>
__global__ __launch_bounds__(256) void k_convt6_wgrad(const float* __restrict__ in, const float* __restrict__ gout,
                                                      float* __restrict__ part, int Ci, int Co, int H, int W, unsigned total,
                                                      int ksteps, int per, int want_gb, CfdDiv dHW, CfdDiv dW, unsigned gbs) {
    __shared__ float s_red[4 * (MTG * NTG * 256 + NTG * 16)];
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int HW = H * W, C4 = 4 * Co;
    const int mb = blockIdx.y * MTG, nb = blockIdx.z * NTG;
    const int kbeg = blockIdx.x * per, kend = kbeg + per < ksteps ? kbeg + per : ksteps;
    const bool sum_cols = want_gb && blockIdx.y == 0;
    f32x4 acc[MTG][NTG];
    float colsum[NTG];
#pragma unroll
    for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTG; ++nt) colsum[nt] = 0.f;
#pragma unroll 1
    for (int ks = kbeg + wave; ks < kend; ks += 4) {
        const unsigned p8 = 32u * ks + 8u * q;
        const bool live = p8 < total;
        const unsigned pc = live ? p8 : 0u;
        const unsigned b = cfd_div(pc, dHW), rem = pc - b * (unsigned)HW;
        // the eight pixels are two runs of four in a row each (W % 4 == 0): (y, x) of both
        unsigned yq[2], xq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            yq[h] = cfd_div(rem + 4u * h, dW);
            xq[h] = rem + 4u * h - yq[h] * (unsigned)W;
        }
        // (loads first, opaque as whole vectors, then the zeroing: see k_convt6)
        f32x4 gf[NTG][4], af[MTG][2];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int c = 16 * (nb + nt) + n, o = c >> 2, dy = (c >> 1) & 1;
            const float* gplane = gout + (size_t)b * gbs + (size_t)(o < Co ? o : Co - 1) * (4 * HW);  // gbs: batch stride of gout
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4* gp = reinterpret_cast<const f32x4*>(gplane + (size_t)(2 * yq[h] + dy) * (2 * W) + 2 * xq[h]);
                gf[nt][2 * h] = gp[0], gf[nt][2 * h + 1] = gp[1];
            }
        }
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            const int i = 16 * (mb + mt) + n;
            const f32x4* ip = reinterpret_cast<const f32x4*>(in + (size_t)(b * (unsigned)Ci + (i < Ci ? i : Ci - 1)) * HW + rem);
            af[mt][0] = ip[0], af[mt][1] = ip[1];
        }
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) gf[nt][k] = cfd_opaque_f4(gf[nt][k]);
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) af[mt][k] = cfd_opaque_f4(af[mt][k]);
        bf16x8 bs[NTG][3];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int c = 16 * (nb + nt) + n, o = c >> 2, dx = c & 1;
            const bool ok = live && o < Co;
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] = dx ? gf[nt][k][1] : gf[nt][k][0];
                v[2 * k + 1] = dx ? gf[nt][k][3] : gf[nt][k][2];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? v[j] : 0.f;
            if (sum_cols) colsum[nt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            const CfdSplit8x3 s = cfd_split8x3(v);
            bs[nt][0] = s.p[0], bs[nt][1] = s.p[1], bs[nt][2] = s.p[2];
        }
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            if (cfd_uniform(16 * (mb + mt)) >= Ci) continue;
            const bool ok = live && 16 * (mb + mt) + n < Ci;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? af[mt][j >> 2][j & 3] : 0.f;
            const CfdSplit8x3 s = cfd_split8x3(v);
            const bf16x8 as[3] = {s.p[0], s.p[1], s.p[2]};
            cfd_mfma_bf16x6_n<NTG>(as, bs, acc[mt]);
        }
    }
    // ---- the four waves' sums, in wave order ----
    float* mine = s_red + wave * (MTG * NTG * 256 + NTG * 16);
#pragma unroll
    for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[((mt * NTG + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    if (sum_cols) {
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            float s = colsum[nt];
            s += cfd_shfl_xor(s, 16);   // the four pixel octets
            s += cfd_shfl_xor(s, 32);
            s += cfd_shfl_xor(s, 1);    // dx
            s += cfd_shfl_xor(s, 2);    // dy
            if (lane < 16) mine[MTG * NTG * 256 + nt * 16 + lane] = s;  // (lanes 4 k .. 4 k + 3 hold channel k's sum)
        }
    }
    __syncthreads();
    float* slice = part + (size_t)blockIdx.x * ((size_t)Ci * C4 + (want_gb ? Co : 0));
    constexpr int WS = MTG * NTG * 256 + NTG * 16;
    for (int e = threadIdx.x; e < MTG * NTG * 256; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8, mt = tile / NTG, nt = tile - mt * NTG;
        const int i = 16 * (mb + mt) + 4 * (ln >> 4) + r, c = 16 * (nb + nt) + (ln & 15);
        if (i < Ci && c < C4) slice[(size_t)i * C4 + c] = (s_red[e] + s_red[WS + e]) + (s_red[2 * WS + e] + s_red[3 * WS + e]);
    }
    if (sum_cols && threadIdx.x < NTG * 4) {
        const int nt = threadIdx.x >> 2, k = threadIdx.x & 3, o = 4 * (nb + nt) + k, e = MTG * NTG * 256 + nt * 16 + 4 * k;
        if (o < Co) slice[(size_t)Ci * C4 + o] = (s_red[e] + s_red[WS + e]) + (s_red[2 * WS + e] + s_red[3 * WS + e]);
    }
}

struct Convt6WgPlan {
    int ksteps, per, groups, mgroups, ngroups;
    size_t bytes;
    bool ok;
};

static Convt6WgPlan convt6_wg_plan(int B, int Ci, int Co, int H, int W, long gout_bstride = 0) {
    Convt6WgPlan P{};
    if (B < 1 || W % 4 != 0 || (H * W) % 8 != 0 || !convt6_small(B, Ci, Co, H, W, gout_bstride) || gout_bstride % 4 != 0) return P;
    const long px = (long)B * H * W;
    P.ksteps = (int)((px + 31) / 32);
    P.mgroups = ((Ci + 15) / 16 + 2) / 3;
    P.ngroups = ((4 * Co + 15) / 16 + 2) / 3;
    // ~1024 workgroups over the launch, at least two steps per wave (each step is an exposed memory round trip: four per wave
    // made every level take 18 us whatever its size), partials capped at ~16 MB
    long want = 1024 / ((long)P.mgroups * P.ngroups);
    if (want < 1) want = 1;
    const long most = (P.ksteps + 7) / 8;
    if (want > most) want = most;
    const long cap = (16L << 20) / (((long)Ci * 4 * Co + Co) * 4 + 1);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    P.per = (int)((P.ksteps + want - 1) / want);
    P.groups = (P.ksteps + P.per - 1) / P.per;
    P.bytes = cfd_align_up((size_t)P.groups * ((size_t)Ci * 4 * Co + Co) * sizeof(float), 256);
    P.ok = true;
    return P;
}

bool cfd_convt6_wgrad_covers(int B, int Ci, int Co, int H, int W) { return convt6_wg_plan(B, Ci, Co, H, W).ok; }
size_t cfd_convt6_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W) { return convt6_wg_plan(B, Ci, Co, H, W).bytes; }

int cfd_convt6_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int H, int W,
                     hipStream_t st, const char* what, long gout_bstride) {
    const Convt6WgPlan P = convt6_wg_plan(B, Ci, Co, H, W, gout_bstride);
    if (!P.ok || ((size_t)in & 15) || ((size_t)gout & 15)) return CFD_ERR_UNSUPPORTED;
    const dim3 grid(P.groups, P.mgroups, P.ngroups);
    hipLaunchKernelGGL((k_convt6_wgrad<3, 3>), grid, dim3(256), 0, st, in, gout, (float*)ws, Ci, Co, H, W, (unsigned)((long)B * H * W),
                       P.ksteps, P.per, gb ? 1 : 0, cfd_div_make((unsigned)(H * W)), cfd_div_make((unsigned)W),
                       (unsigned)(gout_bstride > 0 ? gout_bstride : (long)Co * 4 * H * W));
    CFD_LAUNCH_CHECK(what);
    const long n1 = (long)Ci * 4 * Co;
    cfd_conv_part_reduce((const float*)ws, gw, n1 + (gb ? Co : 0), P.groups, st, gb, n1);
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}
