// 1x1 convolutions of the conv baselines (U-Net OutConv, src/models/unet.py:102-108; ResidualBlock's res_conv, resnet.py:50-52) as
// GEMMs over the pixels on the bf16 matrix pipe with THREE-PIECE operands (cfd_mfma_bf16x6, fp32-exact class), streamed like the
// transposed convolutions of convt6.hip: global -> registers -> split -> MFMA, no LDS staging, lanes n = consecutive pixels.
//   forward         out[b][o][p] = bias[o] + sum_i w[o][i] in[b][i][p]       M = o, N = pixels, K = i
//   input gradient  gin[b][i][p] = sum_o w[o][i] g[b][o][p]                  M = i, N = pixels, K = o
//   weight gradient gw[o][i]     = sum_{b,p} g[b][o][p] in[b][i][p], gb[o] = sum g       M = o, N = i, K = pixels
// (The general kernels of conv.hip they replace -- a gather kernel per direction plus a copy through the "extended" grid and a
// separate bias sum -- took 20 / 16 + 7 / 36 us for the 12 -> 2 channel output layer of the configs[2] U-Net: 25 MB of input.)
// Written at the very end of round 3: parity green on MI355X (kernel tests at the U-Net / ResNet sizes, the models' golden tests with
// the knob on), ONE timing: U-Net step 2.797 -> 2.775 ms.  Off unless the `conv1_mfma` knob is 1 until the whole GPU suite has run
// with it and the per-kernel times are profiled.
#include "cfd_conv.h"

// ------------------------------------------------------------------------------------------------------
// forward (DG = false) / input gradient (DG = true).  Wave = NT tiles of 16 consecutive pixels x MT row tiles (blockIdx.y: row
// group); K steps of 32 channels.  A row n of tile mt = output row 16 (mb + mt) + n: forward w[m][k] (K contiguous), input
// gradient w[k][m].  KSPL: the four waves share one pixel tile and deal the K steps (few pixels, many channels).
// ------------------------------------------------------------------------------------------------------
template <bool DG, int MT, int NT, bool KSPL>
__global__ __launch_bounds__(256) void k_conv1(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias,
                                               float* __restrict__ dst, int Ci, int Co, int HW, unsigned total, CfdDiv dHW) {
    static_assert(!KSPL || NT == 1, "the K split is built for single pixel tiles");
    __shared__ float s_red[KSPL ? 4 * MT * 256 : 1];
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int M = DG ? Ci : Co, K = DG ? Co : Ci;
    const int mb = blockIdx.y * MT;
    const unsigned P0 = KSPL ? blockIdx.x * 16u : (blockIdx.x * 4u + wave) * (16u * NT);
    if (!KSPL && P0 >= total) return;
    unsigned sbase[NT], dbase[NT];  // element offsets of src[b][0][p] / dst[b][0][p]
    bool live[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const unsigned p = P0 + 16u * u + n;
        live[u] = p < total;
        const unsigned pc = live[u] ? p : total - 1;
        const unsigned b = cfd_div(pc, dHW), rem = pc - b * (unsigned)HW;
        sbase[u] = b * (unsigned)(K * HW) + rem;
        dbase[u] = b * (unsigned)(M * HW) + rem;
    }
    f32x4 acc[MT][NT];
    float braw[MT][4];
    const bool with_bias = !DG && bias && (!KSPL || wave == 0);  // (wave-uniform)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 16 * (mb + mt) + 4 * q + r;
            braw[mt][r] = with_bias ? bias[o < Co ? o : Co - 1] : 0.f;  // all twelve requested before the first is used
        }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = cfd_opaque_f(braw[mt][r]);
            bv[r] = 16 * (mb + mt) + 4 * q + r < Co ? t : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[mt][u] = bv;
    }
    const int ksteps = (K + 31) / 32;
#pragma unroll 1
    for (int ks = KSPL ? wave : 0; ks < ksteps; ks += KSPL ? 4 : 1) {
        const int k0 = 32 * ks + 8 * q;
        // every load of the step from a clamped address, then opaque, then the zeroing (`ok ? load : 0` is a
        // branch with a full wait per load)
        float bvv[NT][8], avv[MT][8];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = k0 + j;
                bvv[u][j] = src[(size_t)sbase[u] + (size_t)(c < K ? c : K - 1) * HW];
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = 16 * (mb + mt) + n, mc = m < M ? m : M - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j, kc = k < K ? k : K - 1;
                avv[mt][j] = DG ? w[(size_t)kc * Ci + mc] : w[(size_t)mc * Ci + kc];  // w is (Co, Ci)
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) bvv[u][j] = cfd_opaque_f(bvv[u][j]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 8; ++j) avv[mt][j] = cfd_opaque_f(avv[mt][j]);
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
    if constexpr (KSPL) {  // wave v < MT leaves with the sum of tile v (wave order), the others are done
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
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (KSPL && mt != wave) continue;
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            if (!live[u]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * (mb + mt) + 4 * q + r;
                if (m < M) dst[(size_t)dbase[u] + (size_t)m * HW] = acc[mt][u][r];
            }
        }
    }
}

static bool conv1_small(int B, int Ci, int Co, int HW) { return (long)B * Ci * HW < (1L << 31) && (long)B * Co * HW < (1L << 31); }

bool cfd_conv1_covers(int B, int Ci, int Co, int HW) { return B >= 1 && HW >= 1 && conv1_small(B, Ci, Co, HW); }

template <bool DG>
static int conv1_launch(const float* src, const float* w, const float* bias, float* dst, int B, int Ci, int Co, int HW, hipStream_t st,
                        const char* what) {
    constexpr int MT = 3;
    const unsigned total = (unsigned)((long)B * HW);
    const int M = DG ? Ci : Co, mgroups = ((M + 15) / 16 + MT - 1) / MT;
#ifndef CFD_CONV1_NT4_MIN_WGS
#define CFD_CONV1_NT4_MIN_WGS 512  // (the CPU emulator build sets 2 so that small test shapes reach both forms)
#endif
    const long wg4 = ((long)total + 255) / 256 * mgroups;
    const CfdDiv dHW = cfd_div_make((unsigned)HW);
    if (wg4 >= CFD_CONV1_NT4_MIN_WGS) {
        const dim3 grid((unsigned)(((long)total + 255) / 256), mgroups);
        hipLaunchKernelGGL((k_conv1<DG, MT, 4, false>), grid, dim3(256), 0, st, src, w, bias, dst, Ci, Co, HW, total, dHW);
    } else {
        const dim3 grid((unsigned)(((long)total + 15) / 16), mgroups);
        hipLaunchKernelGGL((k_conv1<DG, MT, 1, true>), grid, dim3(256), 0, st, src, w, bias, dst, Ci, Co, HW, total, dHW);
    }
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

int cfd_conv1_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW, hipStream_t st,
                  const char* what) {
    if (!cfd_conv1_covers(B, Ci, Co, HW)) return CFD_ERR_UNSUPPORTED;
    return conv1_launch<false>(in, w, bias, out, B, Ci, Co, HW, st, what);
}

int cfd_conv1_dgrad(const float* gout, const float* w, float* gin, int B, int Ci, int Co, int HW, hipStream_t st, const char* what) {
    if (!cfd_conv1_covers(B, Ci, Co, HW)) return CFD_ERR_UNSUPPORTED;
    return conv1_launch<true>(gout, w, nullptr, gin, B, Ci, Co, HW, st, what);
}

// ------------------------------------------------------------------------------------------------------
// weight (+ bias) gradient.  K = pixels, 32 per step: lane (q, n) holds the eight consecutive pixels P + 8 q .. + 7 of one image
// (H W % 8 == 0, 16-byte aligned tensors) of row n of its operand: two float4 loads each.  Workgroup (x, y, z) = k-step range x,
// output-channel group y (MTG tiles), input-channel group z (NTG tiles); its four waves take the steps in turn and are summed
// through LDS in wave order; one partial block per k-range in slice x: [Co][Ci] weights | [Co] bias sums (column group 0 only).
// ------------------------------------------------------------------------------------------------------
template <int MTG, int NTG>
__global__ __launch_bounds__(256) void k_conv1_wgrad(const float* __restrict__ gout, const float* __restrict__ in, float* __restrict__ part,
                                                     int Ci, int Co, int HW, unsigned total, int ksteps, int per, int want_gb, CfdDiv dHW) {
    __shared__ float s_red[4 * (MTG * NTG * 256 + MTG * 16)];
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    const int mb = blockIdx.y * MTG, nb = blockIdx.z * NTG;
    const int kbeg = blockIdx.x * per, kend = kbeg + per < ksteps ? kbeg + per : ksteps;
    const bool sum_rows = want_gb && blockIdx.z == 0;
    f32x4 acc[MTG][NTG];
    float rowsum[MTG];
#pragma unroll
    for (int mt = 0; mt < MTG; ++mt) {
        rowsum[mt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int ks = kbeg + wave; ks < kend; ks += 4) {
        const unsigned p8 = 32u * ks + 8u * q;
        const bool live = p8 < total;
        const unsigned pc = live ? p8 : 0u;
        const unsigned b = cfd_div(pc, dHW), rem = pc - b * (unsigned)HW;
        f32x4 gf[MTG][2], xf[NTG][2];
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            const int o = 16 * (mb + mt) + n;
            const f32x4* gp = reinterpret_cast<const f32x4*>(gout + (size_t)(b * (unsigned)Co + (o < Co ? o : Co - 1)) * HW + rem);
            gf[mt][0] = gp[0], gf[mt][1] = gp[1];
        }
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int i = 16 * (nb + nt) + n;
            const f32x4* ip = reinterpret_cast<const f32x4*>(in + (size_t)(b * (unsigned)Ci + (i < Ci ? i : Ci - 1)) * HW + rem);
            xf[nt][0] = ip[0], xf[nt][1] = ip[1];
        }
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) gf[mt][0] = cfd_opaque_f4(gf[mt][0]), gf[mt][1] = cfd_opaque_f4(gf[mt][1]);
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) xf[nt][0] = cfd_opaque_f4(xf[nt][0]), xf[nt][1] = cfd_opaque_f4(xf[nt][1]);
        bf16x8 bs[NTG][3];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const bool ok = live && 16 * (nb + nt) + n < Ci;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? xf[nt][j >> 2][j & 3] : 0.f;
            const CfdSplit8x3 s = cfd_split8x3(v);
            bs[nt][0] = s.p[0], bs[nt][1] = s.p[1], bs[nt][2] = s.p[2];
        }
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            if (cfd_uniform(16 * (mb + mt)) >= Co) continue;
            const bool ok = live && 16 * (mb + mt) + n < Co;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? gf[mt][j >> 2][j & 3] : 0.f;
            if (sum_rows) rowsum[mt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            const CfdSplit8x3 s = cfd_split8x3(v);
            const bf16x8 as[3] = {s.p[0], s.p[1], s.p[2]};
            cfd_mfma_bf16x6_n<NTG>(as, bs, acc[mt]);
        }
    }
    constexpr int WS = MTG * NTG * 256 + MTG * 16;
    float* mine = s_red + wave * WS;
#pragma unroll
    for (int mt = 0; mt < MTG; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[((mt * NTG + nt) * 4 + r) * 64 + lane] = acc[mt][nt][r];
    if (sum_rows) {
#pragma unroll
        for (int mt = 0; mt < MTG; ++mt) {
            float s = rowsum[mt];
            s += cfd_shfl_xor(s, 16);  // the four pixel octets of the step
            s += cfd_shfl_xor(s, 32);
            if (lane < 16) mine[MTG * NTG * 256 + mt * 16 + lane] = s;
        }
    }
    __syncthreads();
    float* slice = part + (size_t)blockIdx.x * ((size_t)Co * Ci + (want_gb ? Co : 0));
    for (int e = threadIdx.x; e < MTG * NTG * 256; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8, mt = tile / NTG, nt = tile - mt * NTG;
        const int o = 16 * (mb + mt) + 4 * (ln >> 4) + r, i = 16 * (nb + nt) + (ln & 15);
        if (o < Co && i < Ci) slice[(size_t)o * Ci + i] = (s_red[e] + s_red[WS + e]) + (s_red[2 * WS + e] + s_red[3 * WS + e]);
    }
    if (sum_rows && threadIdx.x < MTG * 16) {
        const int mt = threadIdx.x >> 4, o = 16 * (mb + mt) + (threadIdx.x & 15), e = MTG * NTG * 256 + threadIdx.x;
        if (o < Co) slice[(size_t)Co * Ci + o] = (s_red[e] + s_red[WS + e]) + (s_red[2 * WS + e] + s_red[3 * WS + e]);
    }
}

struct Conv1WgPlan {
    int ksteps, per, groups, mgroups, ngroups;
    size_t bytes;
    bool ok;
};

static Conv1WgPlan conv1_wg_plan(int B, int Ci, int Co, int HW) {
    Conv1WgPlan P{};
    if (B < 1 || HW % 8 != 0 || !conv1_small(B, Ci, Co, HW)) return P;
    P.ksteps = (int)(((long)B * HW + 31) / 32);
    P.mgroups = ((Co + 15) / 16 + 2) / 3;
    P.ngroups = ((Ci + 15) / 16 + 2) / 3;
    long want = 1024 / ((long)P.mgroups * P.ngroups);  // ~1024 workgroups, at least two steps per wave, partials capped at ~16 MB
    if (want < 1) want = 1;
    const long most = (P.ksteps + 7) / 8;
    if (want > most) want = most;
    const long cap = (16L << 20) / (((long)Co * Ci + Co) * 4 + 1);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    P.per = (int)((P.ksteps + want - 1) / want);
    P.groups = (P.ksteps + P.per - 1) / P.per;
    P.bytes = cfd_align_up((size_t)P.groups * ((size_t)Co * Ci + Co) * sizeof(float), 256);
    P.ok = true;
    return P;
}

bool cfd_conv1_wgrad_covers(int B, int Ci, int Co, int HW) { return conv1_wg_plan(B, Ci, Co, HW).ok; }
size_t cfd_conv1_wgrad_ws_bytes(int B, int Ci, int Co, int HW) { return conv1_wg_plan(B, Ci, Co, HW).bytes; }

int cfd_conv1_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW, hipStream_t st,
                    const char* what) {
    const Conv1WgPlan P = conv1_wg_plan(B, Ci, Co, HW);
    if (!P.ok || ((size_t)in & 15) || ((size_t)gout & 15)) return CFD_ERR_UNSUPPORTED;
    const dim3 grid(P.groups, P.mgroups, P.ngroups);
    hipLaunchKernelGGL((k_conv1_wgrad<3, 3>), grid, dim3(256), 0, st, gout, in, (float*)ws, Ci, Co, HW, (unsigned)((long)B * HW), P.ksteps,
                       P.per, gb ? 1 : 0, cfd_div_make((unsigned)HW));
    CFD_LAUNCH_CHECK(what);
    const long n1 = (long)Co * Ci;
    cfd_conv_part_reduce((const float*)ws, gw, n1 + (gb ? Co : 0), P.groups, st, gb, n1);
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}
