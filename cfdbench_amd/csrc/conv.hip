// Convolution stack of the U-Net / ResNet baselines on the CDNA4 matrix pipe.  k = 3, 5 and 7 run on the three-piece split-bf16
// kernels of conv6.hip; this file holds the C ABI, the general gather kernels (any odd k <= 7, exact-fp32
// v_mfma_f32_16x16x4_f32: k = 1 / 5, and k = 3 / 7 when no workspace is given), BatchNorm, pooling and the transposed conv:
//   nn.Conv2d(k, padding=k/2, padding_mode="replicate")   src/models/unet.py:20-43 (k=3), src/models/resnet.py:35-55 (k=7)
//   nn.BatchNorm2d + ReLU                                  src/models/unet.py:28-30,41-43
//   nn.MaxPool2d(2), nn.ConvTranspose2d(k=2, s=2)          src/models/unet.py:59,80
// All three convolution passes are implicit GEMMs whose data operand is gathered straight from the NCHW tensors
// (replicate padding = index clamping; no im2col buffer):
//   forward : out[o][p]   = sum_{(i,ky,kx)} w[o][(i,ky,kx)] * in[i][clamp(p + (ky,kx) - pad)]          M = Cout, N = pixels
//   input gradient: the forward is V(P(in)) with P = replicate pad, V = valid convolution, so gin = P^T(V^T(g)):
//             V^T(g) is the same gather kernel on the (H+2pad) x (W+2pad) extended grid with zero outside g, and
//             P^T folds the pad ring back onto the border pixels (k_fold_pad) -- deterministic, no atomics
//   weight gradient: gw[o][(i,ky,kx)] = sum_p g[o][p] * in[i][clamp(...)]                    M = Cout, N = Cin*k*k, K = pixels
#include "cfd_conv.h"

#define CV_WAVES 4

// (i, ky, kx) of flat index k over Ci*ks*ks
__device__ __forceinline__ void conv_split_k(int k, int ks, int& i, int& ky, int& kx) {
    const int kk = ks * ks;
    i = k / kk;
    const int r = k - i * kk;
    ky = r / ks;
    kx = r - ky * ks;
}

// ------------------------------------------------------------------------------------------------------
// forward / transposed-valid gather GEMM:  dst[b][m][p] = bias[m] + sum_k A(m,k) * src[b][c(k)][pos(p,k)]
//   EXT = false: forward.    p over H x W, pos = clamp(p + (ky,kx) - pad);                A(m,k) = w[m][k]       (m = o)
//   EXT = true : V^T(g).     p over (H+2pad) x (W+2pad), pos = p - (ky,kx), zero outside; A(m,k) = w[o][m][ky][kx] (m = i)
// A wave owns 16 consecutive (linear) pixels of one batch entry and every output channel (MT tiles of 16).
// ------------------------------------------------------------------------------------------------------
template <int MT, bool EXT>
__global__ __launch_bounds__(64 * CV_WAVES) void k_conv_gather(const float* __restrict__ src, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ dst,
                                                               ConvGeom g, int tiles_per_b, long total_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int pad = g.ks / 2, kk = g.ks * g.ks;
    const int Cs = EXT ? g.Co : g.Ci;       // channels of src (contracted)
    const int Cm = EXT ? g.Ci : g.Co;       // channels of dst (M)
    const int K = Cs * kk;
    const int Hd = EXT ? g.H + 2 * pad : g.H, Wd = EXT ? g.W + 2 * pad : g.W;  // dst grid
    const int HWd = Hd * Wd, HWs = g.H * g.W;
    const int mbase = blockIdx.y * 16 * MT;  // this workgroup's slice of the output channels
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long tile = (long)blockIdx.x * CV_WAVES + wave; tile < total_tiles; tile += (long)gridDim.x * CV_WAVES) {
        const int b = (int)(tile / tiles_per_b);
        const int p = (int)(tile - (long)b * tiles_per_b) * 16 + n;  // this lane's pixel (data operand column)
        const bool pvalid = p < HWd;
        const int py = pvalid ? p / Wd : 0, px = pvalid ? p - py * Wd : 0;
        const float* sb = src + (size_t)b * Cs * HWs;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = zero;
        // (channel, ky, kx) of this lane's k = k0 + q advances by carry: two runtime integer divisions per k-step cost
        // more VALU time than the MFMAs they feed
        int c, ky, kx;
        conv_split_k(q, g.ks, c, ky, kx);
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + q;
            float bv = 0.f;
            if (k < K && pvalid) {
                if constexpr (EXT) {
                    const int y = py - ky, x = px - kx;
                    if (y >= 0 && y < g.H && x >= 0 && x < g.W) bv = sb[(size_t)c * HWs + y * g.W + x];
                } else {
                    int y = py + ky - pad, x = px + kx - pad;
                    y = y < 0 ? 0 : (y >= g.H ? g.H - 1 : y);
                    x = x < 0 ? 0 : (x >= g.W ? g.W - 1 : x);
                    bv = sb[(size_t)c * HWs + y * g.W + x];
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mbase + 16 * mt + n;  // A operand row
                float av = 0.f;
                if (m < Cm && k < K) av = EXT ? w[((size_t)c * g.Ci + m) * kk + ky * g.ks + kx] : w[(size_t)m * K + k];
                acc[mt] = cfd_mfma16x16x4(av, bv, acc[mt]);
            }
            kx += 4;
            while (kx >= g.ks) { kx -= g.ks; ++ky; }
            while (ky >= g.ks) { ky -= g.ks; ++c; }
        }
        if (pvalid) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mbase + 16 * mt + 4 * q + r;
                    if (m < Cm) dst[((size_t)b * Cm + m) * HWd + p] = acc[mt][r] + (bias ? bias[m] : 0.f);
                }
        }
    }
}

// out[e] = bias[channel(e)] + sum_z part[z][e]   (fixed order; e over (B, Cm, HWd))
__global__ __launch_bounds__(256) void k_splitk_sum(const float* __restrict__ part, const float* __restrict__ bias,
                                                    float* __restrict__ out, unsigned n, int nz, int Cm, CfdDiv dHW, CfdDiv dC) {
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        float s = 0.f;
        if (bias) {
            const unsigned row = cfd_div(e, dHW);
            s = bias[row - cfd_div(row, dC) * (unsigned)Cm];
        }
        for (int z = 0; z < nz; ++z) s += part[(size_t)z * n + e];
        out[e] = s;
    }
}

int cfd_conv_splitk_sum(const float* part, const float* bias, float* out, long n, int nz, int Cm, long HWd, hipStream_t st,
                        const char* what) {
    CFD_REQUIRE_I31(n, what);
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_splitk_sum, dim3((unsigned)blocks), dim3(256), 0, st, part, bias, out, (unsigned)n, nz, Cm,
                       cfd_div_make((unsigned)HWd), cfd_div_make((unsigned)Cm));
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

void cfd_conv_tile_shape(int Hd, int Wd, int B, int& TW, int& TH, int& NB) {
    TW = Wd >= 32 ? 32 : (Wd > 8 ? 16 : (Wd > 4 ? 8 : 4));
    const int rows = 256 / TW;  // rows available per workgroup
    int th = 1;
    while (th < (Hd < rows ? Hd : rows)) th <<= 1;  // power of two so that NB*TH*TW == 256 exactly
    TH = th > rows ? rows : th;
    NB = 256 / (TH * TW);
    // Grids that are not powers of two -- above all the (H+2pad) x (W+2pad) extended grid of the input-gradient pass -- waste
    // whole workgroups on power-of-two tiles (66x66 on 32x8 tiles: 27 tiles per image for 17 tiles' worth of pixels).  Search
    // the tile shapes with TW*TH <= 256 for the fewest workgroups per image (a partly filled workgroup costs as much as a full
    // one); ties go to the wider tile (longer store runs).
    const auto cost = [&](int tw, int thh) {
        const int nb = 256 / (tw * thh);
        return (double)((Wd + tw - 1) / tw) * ((Hd + thh - 1) / thh) / (double)(nb < B ? nb : B);
    };
    double best = cost(TW, TH) * 0.97;  // keep the power-of-two shape unless another one is clearly better
    int btw = TW, bth = TH;
    for (int tw = 4; tw <= 64 && tw <= Wd; ++tw)
        for (int thh = 1; thh * tw <= 256 && thh <= Hd; ++thh) {
            // at least half of the lanes busy -- or the tile is the WHOLE image and several images share the workgroup (round 6: the
            // 10 x 10 and 6 x 6 extended grids of the 8 x 8 / 4 x 4 U-Net levels fell back to one 16 x 16 / four 8 x 8 tiles per workgroup,
            // 39 % / 56 % of the lanes, where two 10 x 10 / seven 6 x 6 images fill 78 % / 98 %; thin tiles of many images are not
            // admitted: their halo is three times their area)
            if (tw * thh < 128 && !(tw == Wd && thh == Hd)) continue;
            const double c = cost(tw, thh);
            if (c < best - 1e-9 || (c < best + 1e-9 && tw > btw)) { best = c; btw = tw; bth = thh; }
        }
    TW = btw;
    TH = bth;
    NB = 256 / (TW * TH);
}

template <bool EXT>
static int launch_conv_gather(const float* src, const float* w, const float* bias, float* dst, const ConvGeom& g,
                              hipStream_t st, const char* what) {
    const int pad = g.ks / 2;
    const int HWd = EXT ? (g.H + 2 * pad) * (g.W + 2 * pad) : g.H * g.W;
    const int tiles_per_b = (HWd + 15) / 16;
    const long total = (long)g.B * tiles_per_b;
    long blocks = (total + CV_WAVES - 1) / CV_WAVES;
    if (blocks > 4096) blocks = 4096;
    const int Cm = EXT ? g.Ci : g.Co, MT = (Cm + 15) / 16;
    // Few pixels (deep, wide layers): split the output channels over blockIdx.y so that the grid still fills the chip;
    // many pixels: one wave keeps every output channel (the gathered data operand is then loaded once).
    int mtw = MT;  // M tiles per wave
    while (mtw > 1 && total * ((MT + mtw - 1) / mtw) < 2048) mtw = (mtw + 1) / 2;
    if (mtw == 5) mtw = 6;
    if (mtw > 6 && mtw < 12) mtw = 8;
    const int mgroups = (MT + mtw - 1) / mtw;
#define CV_L(M_) hipLaunchKernelGGL((k_conv_gather<M_, EXT>), dim3((unsigned)blocks, mgroups), dim3(64 * CV_WAVES), 0, st, src, w, bias, dst, g, tiles_per_b, total)
    switch (mtw) {
        case 1: CV_L(1); break;
        case 2: CV_L(2); break;
        case 3: CV_L(3); break;
        case 4: CV_L(4); break;
        case 6: CV_L(6); break;
        case 8: CV_L(8); break;
        case 12: CV_L(12); break;
        default:
            cfd_set_error("%s: %d output channels unsupported (max 192)", what, Cm);
            return CFD_ERR_UNSUPPORTED;
    }
#undef CV_L
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

static int conv_check(const char* fn, int B, int Ci, int Co, int H, int W, int ks) {
    CFD_REQUIRE(B >= 0 && Ci >= 1 && Co >= 1 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "%s: bad sizes", fn);
    CFD_REQUIRE(ks >= 1 && ks <= 7 && (ks & 1), CFD_ERR_UNSUPPORTED, "%s: kernel size %d unsupported (odd, <= 7)", fn, ks);
    CFD_REQUIRE(Ci <= 192 && Co <= 192, CFD_ERR_UNSUPPORTED, "%s: channels (%d -> %d) unsupported (max 192)", fn, Ci, Co);
    return CFD_OK;
}

// out (B,Co,H,W) = conv2d(in (B,Ci,H,W), w (Co,Ci,ks,ks), bias, padding=ks/2, padding_mode="replicate")
extern "C" size_t cfd_conv2d_fwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks) {
    if (B <= 0 || Ci < 1 || Co < 1 || H < 1 || W < 1) return 0;
    const ConvGeom g{B, Ci, Co, H, W, ks};
    if (cfd_conv6_covers(g, false)) return cfd_conv6_ws_bytes(g, false);  // weight fragments (+ split-K partials)
    return 0;
}

// The same forward, also emitting per-channel records for the training-mode BatchNorm that follows the conv:
// stats (Co, slots, 4) = (m, m2, n, -) = mean of out - bias, sum of squared deviations from it and pixel count of each slot (disjoint
// pixel sets), slots = cfd_conv2d_fwd_stats_slots() (0: this layer cannot emit them -- use cfd_conv2d_fwd and cfd_batchnorm_fwd).
// cfd_batchnorm_fwd_stats consumes them.
extern "C" int cfd_conv2d_fwd_stats_slots(int B, int Ci, int Co, int H, int W, int ks) {
    if (B <= 0 || Ci < 1 || Co < 1 || H < 1 || W < 1) return 0;
    const ConvGeom g{B, Ci, Co, H, W, ks};
    return cfd_conv6_covers(g, false) ? cfd_conv6_stats_slots(g) : 0;
}

// Fragments of the weights made ahead of the calls that use them (conv6.hip: the form depends on (Ci, Co, ks, transposed) only):
// cfd_conv2d_wprep_batch fills the buffers of any number of layers in ONE launch; cfd_conv2d_fwd_ex / cfd_conv2d_bwd_ex take them
// and skip their own preparation launch.  transposed = 0: the forward pass; 1: the input-gradient pass.
extern "C" size_t cfd_conv2d_wfrag_bytes(int Ci, int Co, int ks, int transposed) {
    if (Ci < 1 || Co < 1 || Ci > 192 || Co > 192) return 0;
    return cfd_conv6_wfrag_bytes(Ci, Co, ks, transposed != 0);
}

extern "C" int cfd_conv2d_wprep_batch(int n, const float* const* w, void* const* wfrag, const int* Ci, const int* Co, const int* ks,
                                      const int* transposed, void* stream) {
    CFD_REQUIRE(n >= 0, CFD_ERR_INVALID_ARG, "cfd_conv2d_wprep_batch: negative count");
    if (n == 0) return CFD_OK;
    CFD_REQUIRE(w && wfrag && Ci && Co && ks && transposed, CFD_ERR_INVALID_ARG, "cfd_conv2d_wprep_batch: NULL table");
    for (int i = 0; i < n; ++i)
        CFD_REQUIRE(cfd_conv2d_wfrag_bytes(Ci[i], Co[i], ks[i], transposed[i]) > 0, CFD_ERR_UNSUPPORTED,
                    "cfd_conv2d_wprep_batch: item %d (%d -> %d channels, kernel size %d) has no fragment form", i, Ci[i], Co[i], ks[i]);
    CFD_PROF_W("k_conv_wprep", (hipStream_t)stream, 0.0, 0.0);  // (an implementation detail of the k = 3 / 7 kernels: no algorithmic bytes)
    return cfd_conv6_wprep_batch(n, w, wfrag, Ci, Co, ks, transposed, (hipStream_t)stream, "cfd_conv2d_wprep_batch");
}

// ws: cfd_conv2d_fwd_workspace_bytes() bytes, or NULL (then k = 3 / 7 run on the exact-fp32 gather kernel of this file instead of
// the three-piece bf16 kernels of conv6.hip).  stats: NULL or the (Co, slots, 4) records above.  wfrag: NULL or the weights'
// forward fragments (cfd_conv2d_wprep_batch); ignored where the layer does not run on the fragment kernels.
extern "C" int cfd_conv2d_fwd_ex(const float* in, const float* w, const float* bias, float* out, void* ws, float* stats,
                                 const void* wfrag, int B, int Ci, int Co, int H, int W, int ks, void* stream) {
    CFD_REQUIRE(in && w && out, CFD_ERR_INVALID_ARG, "cfd_conv2d_fwd: NULL pointer");
    CFD_TRY(conv_check("cfd_conv2d_fwd", B, Ci, Co, H, W, ks));
    if (stats) {
        CFD_REQUIRE(ws, CFD_ERR_INVALID_ARG, "cfd_conv2d_fwd: statistics need the workspace");
        CFD_REQUIRE(B >= 1, CFD_ERR_INVALID_ARG, "cfd_conv2d_fwd: statistics of an empty batch");
        CFD_REQUIRE(cfd_conv2d_fwd_stats_slots(B, Ci, Co, H, W, ks) > 0, CFD_ERR_UNSUPPORTED,
                    "cfd_conv2d_fwd: this layer emits no statistics (cfd_conv2d_fwd_stats_slots() == 0)");
    }
    if (B == 0) return CFD_OK;
    const ConvGeom g{B, Ci, Co, H, W, ks};
    CFD_PROF_W("k_conv_fwd", (hipStream_t)stream, 4.0 * ((double)B * (Ci + Co) * H * W + (double)Co * Ci * ks * ks),
               2.0 * B * H * W * (double)Co * Ci * ks * ks);
    if (ws && cfd_conv6_covers(g, false))
        return cfd_conv6_run(in, w, bias, out, ws, g, false, nullptr, nullptr, (hipStream_t)stream, "cfd_conv2d_fwd", stats, wfrag);
    if (ks == 1 && cfd_tune_get(CFD_TUNE_CONV1_MFMA) != 0 && cfd_conv1_covers(B, Ci, Co, H * W))
        return cfd_conv1_fwd(in, w, bias, out, B, Ci, Co, H * W, (hipStream_t)stream, "cfd_conv2d_fwd");
    return launch_conv_gather<false>(in, w, bias, out, g, (hipStream_t)stream, "cfd_conv2d_fwd");
}

extern "C" int cfd_conv2d_fwd(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int Ci, int Co,
                              int H, int W, int ks, void* stream) {
    return cfd_conv2d_fwd_ex(in, w, bias, out, ws, nullptr, nullptr, B, Ci, Co, H, W, ks, stream);
}

extern "C" int cfd_conv2d_fwd_stats(const float* in, const float* w, const float* bias, float* out, void* ws, float* stats, int B,
                                    int Ci, int Co, int H, int W, int ks, void* stream) {
    CFD_REQUIRE(stats, CFD_ERR_INVALID_ARG, "cfd_conv2d_fwd_stats: NULL pointer");
    return cfd_conv2d_fwd_ex(in, w, bias, out, ws, stats, nullptr, B, Ci, Co, H, W, ks, stream);
}

// ------------------------------------------------------------------------------------------------------
// nn.Conv2d(k, padding = k / 2) with ZERO padding (the CNN branch of src/models/auto_deeponet_cnn.py:17-33) on the conv6 kernels: the
// halo outside the image loads as zeros instead of the clamped pixel; the input gradient has nothing to fold -- every image pixel is
// one position of the extended grid and is written straight to gin.  (Until round 4 the model zero-padded its input by k / 2 with
// F.pad, ran the replicate kernel on the larger grid and cropped: ~6 extra launches per convolution and direction.)
// ------------------------------------------------------------------------------------------------------
static int chan_sum(const float* g, float* out, void* ws, int B, int C, int HW, hipStream_t st, const char* what);  // (defined below)
extern "C" int cfd_conv2d_zeropad_supported(int B, int Ci, int Co, int H, int W, int ks) {
    if (B < 1 || Ci < 1 || Co < 1 || H < 1 || W < 1) return 0;
    const ConvGeom g{B, Ci, Co, H, W, ks, 1};
    return cfd_conv6_zeropad_covers(g) ? 1 : 0;
}

extern "C" int cfd_conv2d_zeropad_fwd(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int Ci, int Co,
                                      int H, int W, int ks, void* stream) {
    CFD_REQUIRE(in && w && out && ws, CFD_ERR_INVALID_ARG, "cfd_conv2d_zeropad_fwd: NULL pointer");
    CFD_TRY(conv_check("cfd_conv2d_zeropad_fwd", B, Ci, Co, H, W, ks));
    CFD_REQUIRE(cfd_conv2d_zeropad_supported(B, Ci, Co, H, W, ks), CFD_ERR_UNSUPPORTED,
                "cfd_conv2d_zeropad_fwd: shape outside the conv6 kernels (cfd_conv2d_zeropad_supported() == 0)");
    const ConvGeom g{B, Ci, Co, H, W, ks, 1};
    CFD_PROF_W("k_conv_fwd", (hipStream_t)stream, 4.0 * ((double)B * (Ci + Co) * H * W + (double)Co * Ci * ks * ks),
               2.0 * B * H * W * (double)Co * Ci * ks * ks);
    return cfd_conv6_run(in, w, bias, out, ws, g, false, nullptr, nullptr, (hipStream_t)stream, "cfd_conv2d_zeropad_fwd", nullptr, nullptr);
}

extern "C" int cfd_conv2d_zeropad_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                                      int B, int Ci, int Co, int H, int W, int ks, void* stream) {
    CFD_REQUIRE(gout && in && w && ws, CFD_ERR_INVALID_ARG, "cfd_conv2d_zeropad_bwd: NULL pointer");
    CFD_TRY(conv_check("cfd_conv2d_zeropad_bwd", B, Ci, Co, H, W, ks));
    CFD_REQUIRE(cfd_conv2d_zeropad_supported(B, Ci, Co, H, W, ks), CFD_ERR_UNSUPPORTED,
                "cfd_conv2d_zeropad_bwd: shape outside the conv6 kernels (cfd_conv2d_zeropad_supported() == 0)");
    hipStream_t st = (hipStream_t)stream;
    const ConvGeom g{B, Ci, Co, H, W, ks, 1};
    const int HW = H * W, pad = ks / 2;
    const size_t ext_bytes = cfd_align_up((size_t)B * Ci * (H + 2 * pad) * (W + 2 * pad) * sizeof(float), 256);
    if (gin) {  // (ws layout of cfd_conv2d_bwd: [extended grid -- unused here | fragments of the input-gradient pass | weight-gradient partials])
        bool direct = false;
        CFD_PROF_W("k_conv_dgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks), 2.0 * B * HW * (double)Co * Ci * ks * ks);
        CFD_TRY(cfd_conv6_run(gout, w, nullptr, (float*)ws, (char*)ws + ext_bytes, g, true, gin, &direct, st, "cfd_conv2d_zeropad_bwd(dgrad)",
                              nullptr, nullptr));
    }
    if (gw) {
        void* wws = (char*)ws + ext_bytes + cfd_conv6_ws_bytes(g, true);
        CFD_PROF_W("k_conv_wgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks), 2.0 * B * HW * (double)Co * Ci * ks * ks);
        CFD_TRY(cfd_conv6_wgrad(gout, in, gw, gb, wws, g, st, "cfd_conv2d_zeropad_bwd(wgrad)", nullptr));
    } else if (gb) {
        CFD_TRY(chan_sum(gout, gb, ws, B, Co, HW, st, "cfd_conv2d_zeropad_bwd(bias)"));
    }
    return CFD_OK;
}

// gin[b][i][y][x] = sum over the extended positions that replicate padding maps to (y, x)
__global__ __launch_bounds__(256) void k_fold_pad(const float* __restrict__ ext, float* __restrict__ gin, unsigned total,
                                                  int H, int W, int pad, CfdDiv dHW, CfdDiv dW) {
    const int He = H + 2 * pad, We = W + 2 * pad;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHW);
        const int p = (int)(e - img * (unsigned)(H * W)), y = (int)cfd_div((unsigned)p, dW), x = p - y * W;
        const int y0 = y == 0 ? 0 : y + pad, y1 = y == H - 1 ? He - 1 : y + pad;
        const int x0 = x == 0 ? 0 : x + pad, x1 = x == W - 1 ? We - 1 : x + pad;
        const float* s = ext + (size_t)img * He * We;
        float acc = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) acc += s[yy * We + xx];
        gin[e] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------
// weight gradient: gw[o][j] = sum_{b,p} g[b][o][p] * in[b][i(j)][clamp(p + (ky,kx)(j) - pad)],  j over Ci*ks*ks
// Workgroup = one 32 x 64 block of the (o, j) plane x one chunk of pixels; partials are reduced in a fixed order.
// ------------------------------------------------------------------------------------------------------
#define CW_MT 2
#define CW_NT 4
// TRANSPOSED = false: convolution.     A rows = output channels of gout, B columns j = (i, ky, kx) gathered from `in` with
//                                       replicate clamping.
// TRANSPOSED = true : ConvTranspose2d(k=2, s=2) (w (Ci,Co,2,2)): A rows = input channels i of `in` (on the H x W grid),
//                                       B columns j = (o, ky, kx) gathered from gout at (2y+ky, 2x+kx) on the 2H x 2W grid.
// Pixels are one linear space over (batch, H*W); workgroup = one 32 x 64 block of the (row, j) plane x one chunk of it.
template <bool TRANSPOSED>
__global__ __launch_bounds__(64 * CV_WAVES) void k_conv_wgrad(const float* __restrict__ arows, const float* __restrict__ bsrc,
                                                              float* __restrict__ part, ConvGeom g, unsigned chunk_px,
                                                              CfdDiv dHW, CfdDiv dW) {
    __shared__ float s_red[CV_WAVES * CW_MT * CW_NT * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int ks = TRANSPOSED ? 2 : g.ks, pad = ks / 2, kk = ks * ks;
    const int R = TRANSPOSED ? g.Ci : g.Co;          // rows of the result (A operand channels)
    const int Cb = TRANSPOSED ? g.Co : g.Ci;         // channels of the gathered tensor
    const int J = Cb * kk, HW = g.H * g.W;
    const int HWb = TRANSPOSED ? 4 * HW : HW, Wb = TRANSPOSED ? 2 * g.W : g.W;
    const int r0 = blockIdx.y * 16 * CW_MT, j0 = blockIdx.z * 16 * CW_NT;
    int cj[CW_NT], dy[CW_NT], dx[CW_NT];
    bool jv[CW_NT];
#pragma unroll
    for (int nt = 0; nt < CW_NT; ++nt) {
        const int j = j0 + 16 * nt + n;
        jv[nt] = j < J;
        int ky, kx;
        conv_split_k(jv[nt] ? j : 0, ks, cj[nt], ky, kx);
        dy[nt] = TRANSPOSED ? ky : ky - pad;
        dx[nt] = TRANSPOSED ? kx : kx - pad;
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[CW_MT][CW_NT];
#pragma unroll
    for (int a = 0; a < CW_MT; ++a)
#pragma unroll
        for (int c = 0; c < CW_NT; ++c) acc[a][c] = zero;
    // 31-bit pixel indices (checked by the callers) and magic-number division: the per-k-step index split used to be
    // a 64-bit and a 32-bit hardware-less division, more VALU work than the eight MFMAs it feeds
    const unsigned total = (unsigned)g.B * HW;
    const unsigned pbeg = blockIdx.x * chunk_px;
    const unsigned pend = pbeg + chunk_px < total ? pbeg + chunk_px : total;
    for (unsigned p0 = pbeg + 4 * wave; p0 < pend; p0 += 4 * CV_WAVES) {  // waves interleave 4-pixel k-steps
        const unsigned P = p0 + q;
        const bool pv = P < pend;
        const int b = pv ? (int)cfd_div(P, dHW) : 0;
        const int p = pv ? (int)(P - (unsigned)b * HW) : 0;
        const int y = (int)cfd_div((unsigned)p, dW), x = p - y * g.W;
        const float* ab = arows + (size_t)b * R * HW;
        const float* bb = bsrc + (size_t)b * Cb * HWb;
        float av[CW_MT], bv[CW_NT];
#pragma unroll
        for (int a = 0; a < CW_MT; ++a) {
            const int r = r0 + 16 * a + n;
            av[a] = (pv && r < R) ? ab[(size_t)r * HW + p] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CW_NT; ++c) {
            float v = 0.f;
            if (pv && jv[c]) {
                int yy, xx;
                if constexpr (TRANSPOSED) { yy = 2 * y + dy[c]; xx = 2 * x + dx[c]; }
                else {
                    yy = y + dy[c]; xx = x + dx[c];
                    yy = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
                    xx = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);
                }
                v = bb[(size_t)cj[c] * HWb + yy * Wb + xx];
            }
            bv[c] = v;
        }
#pragma unroll
        for (int a = 0; a < CW_MT; ++a)
#pragma unroll
            for (int c = 0; c < CW_NT; ++c) acc[a][c] = cfd_mfma16x16x4(av[a], bv[c], acc[a][c]);
    }
    // block reduction of the 4 waves (fixed order), one partial tile per (chunk, row block, j block)
#pragma unroll
    for (int a = 0; a < CW_MT; ++a)
#pragma unroll
        for (int c = 0; c < CW_NT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[(((wave * CW_MT + a) * CW_NT + c) * 4 + r) * 64 + lane] = acc[a][c][r];
    __syncthreads();
    float* dst = part + (size_t)blockIdx.x * R * J;
    for (int e = threadIdx.x; e < CW_MT * CW_NT * 4 * 64; e += blockDim.x) {
        const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
        const int a = tile / CW_NT, c = tile % CW_NT;
        const int row = r0 + 16 * a + 4 * (ln >> 4) + r, j = j0 + 16 * c + (ln & 15);
        if (row < R && j < J) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < CV_WAVES; ++wv) sum += s_red[(((wv * CW_MT + a) * CW_NT + c) * 4 + r) * 64 + ln];
            dst[(size_t)row * J + j] = sum;
        }
    }
}

// Chunks of the linear (batch, pixel) space: enough workgroups to fill the chip, but no more partial tiles than ~32 MB.
static void wgrad_plan(long total_px, int R, int J, long& chunk_px, int& nchunk) {
    const long tiles = (long)((R + 16 * CW_MT - 1) / (16 * CW_MT)) * ((J + 16 * CW_NT - 1) / (16 * CW_NT));
    long want = (2048 + tiles - 1) / tiles;                      // workgroups ~ 2048
    const long cap = (32L << 20) / ((long)R * J * 4 + 1);        // partial-sum bytes
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    chunk_px = (total_px + want - 1) / want;
    if (chunk_px < 64) chunk_px = 64;
    chunk_px = (chunk_px + 15) / 16 * 16;
    nchunk = (int)((total_px + chunk_px - 1) / chunk_px);
}

// out[e] = sum_chunk part[chunk][e], fixed order.  G thread groups of a workgroup share an element range and split
// the chunks (group k takes chunks k, k+G, ...; 4 loads in flight each), then combine through LDS: a layer with a small
// weight tensor and ~1000 partial slices (the wide, shallow U-Net levels) would otherwise leave each of a handful of
// threads a serial chain of 1000 dependent-latency loads (measured 60 us for 1296 outputs).
template <int G>
__device__ __forceinline__ void part_reduce_body(const float* __restrict__ part, float* __restrict__ out, long n, int nchunk,
                                                 float* __restrict__ out2, long n1, int blk, int nblk, float* s_p) {
    constexpr int E = 256 / G;
    const int el = threadIdx.x % E, grp = threadIdx.x / E;
    for (long e0 = (long)blk * E; e0 < n; e0 += (long)nblk * E) {
        const long e = e0 + el;
        float s = 0.f;
        if (e < n) {
            int c = grp;
            for (; c + 3 * G < nchunk; c += 4 * G) {
                const float v0 = part[(size_t)c * n + e], v1 = part[(size_t)(c + G) * n + e];
                const float v2 = part[(size_t)(c + 2 * G) * n + e], v3 = part[(size_t)(c + 3 * G) * n + e];
                s += (v0 + v1) + (v2 + v3);
            }
            for (; c < nchunk; c += G) s += part[(size_t)c * n + e];
        }
        if (G == 1) {
            if (e < n) { if (e < n1) out[e] = s; else out2[e - n1] = s; }
        } else {
            s_p[threadIdx.x] = s;
            __syncthreads();
            if (grp == 0 && e < n) {
                float tot = 0.f;
#pragma unroll
                for (int k = 0; k < G; ++k) tot += s_p[k * E + el];
                if (e < n1) out[e] = tot; else out2[e - n1] = tot;
            }
            __syncthreads();
        }
    }
}

template <int G>
__global__ __launch_bounds__(256) void k_part_reduce(const float* __restrict__ part, float* __restrict__ out, long n, int nchunk,
                                                     float* __restrict__ out2, long n1) {
    __shared__ float s_p[256];
    part_reduce_body<G>(part, out, n, nchunk, out2, n1, blockIdx.x, gridDim.x, s_p);
}

// workgroup `blk` of job.nblk (256 threads) of a carried reduction
__device__ __forceinline__ void cfd_part_reduce_job_run(const CfdPartReduceJob& j, int blk, float* s_p) {
    if (j.G == 16) part_reduce_body<16>(j.part, j.out, j.n, j.nchunk, j.out2, j.n1, blk, j.nblk, s_p);
    else if (j.G == 4) part_reduce_body<4>(j.part, j.out, j.n, j.nchunk, j.out2, j.n1, blk, j.nblk, s_p);
    else part_reduce_body<1>(j.part, j.out, j.n, j.nchunk, j.out2, j.n1, blk, j.nblk, s_p);
}

// The same for the border pixels only (top and bottom rows, then the left and right columns without their corners): the interior
// of gin was written by the input-gradient kernel itself (conv6.hip, `gin_direct`), so the fold touches 2 (H + W) - 4 pixels per
// image instead of H * W and the extended buffer is read along its rim only.  The first job.nblk workgroups of the launch run a
// carried partial-sum reduction instead (the weight gradient of the same cfd_conv2d_bwd call).
__global__ __launch_bounds__(256) void k_fold_border(const float* __restrict__ ext, float* __restrict__ gin, unsigned total, int H,
                                                     int W, int pad, CfdDiv dNB, const CfdPartReduceJob job) {
    __shared__ float s_p[256];
    if ((int)blockIdx.x < job.nblk) {
        cfd_part_reduce_job_run(job, blockIdx.x, s_p);
        return;
    }
    const unsigned blk = blockIdx.x - job.nblk, nblk = gridDim.x - job.nblk;
    const int He = H + 2 * pad, We = W + 2 * pad, nb = 2 * W + 2 * (H - 2);
    for (unsigned e = blk * blockDim.x + threadIdx.x; e < total; e += nblk * blockDim.x) {
        const unsigned img = cfd_div(e, dNB);
        const int j = (int)(e - img * (unsigned)nb);
        int y, x;
        if (j < W) { y = 0; x = j; }
        else if (j < 2 * W) { y = H - 1; x = j - W; }
        else if (j < 2 * W + H - 2) { y = 1 + (j - 2 * W); x = 0; }
        else { y = 1 + (j - 2 * W - (H - 2)); x = W - 1; }
        const int y0 = y == 0 ? 0 : y + pad, y1 = y == H - 1 ? He - 1 : y + pad;
        const int x0 = x == 0 ? 0 : x + pad, x1 = x == W - 1 ? We - 1 : x + pad;
        const float* s = ext + (size_t)img * He * We;
        float acc = 0.f;
        for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx) acc += s[yy * We + xx];
        gin[(size_t)img * H * W + y * W + x] = acc;
    }
}

// elements [0, n1) of a partial slice go to out, elements [n1, n) to out2 (the conv6 weight gradient carries the bias gradient behind
// the weights); out2 == NULL: n1 = n
CfdPartReduceJob cfd_conv_part_reduce_job(const float* part, float* out, long n, int nchunk, float* out2, long n1) {
    CfdPartReduceJob j{part, out, out2, n, out2 ? n1 : n, nchunk, 0, 1};
    long blocks;
    if (nchunk >= 128) { j.G = 16; blocks = (n + 15) / 16; if (blocks > 2048) blocks = 2048; }
    else if (nchunk >= 24) { j.G = 4; blocks = (n + 63) / 64; if (blocks > 2048) blocks = 2048; }
    else { j.G = 1; blocks = (n + 255) / 256; if (blocks > 1024) blocks = 1024; }
    j.nblk = (int)blocks;
    return j;
}

void cfd_conv_part_reduce(const float* part, float* out, long n, int nchunk, hipStream_t st, float* out2, long n1) {
    const CfdPartReduceJob j = cfd_conv_part_reduce_job(part, out, n, nchunk, out2, n1);
    if (j.G == 16) hipLaunchKernelGGL((k_part_reduce<16>), dim3((unsigned)j.nblk), dim3(256), 0, st, part, out, n, nchunk, out2, j.n1);
    else if (j.G == 4) hipLaunchKernelGGL((k_part_reduce<4>), dim3((unsigned)j.nblk), dim3(256), 0, st, part, out, n, nchunk, out2, j.n1);
    else hipLaunchKernelGGL((k_part_reduce<1>), dim3((unsigned)j.nblk), dim3(256), 0, st, part, out, n, nchunk, out2, j.n1);
}

// per-channel sum over (B, HW) in two parallel stages (defined next to the BatchNorm reductions it shares)
static int chan_sum(const float* g, float* out, void* ws, int B, int C, int HW, hipStream_t st, const char* what);
static size_t chan_sum_ws_bytes(int C);

extern "C" size_t cfd_conv2d_bwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks) {
    if (B <= 0) return 0;
    const int pad = ks / 2;
    const size_t ext = cfd_align_up((size_t)B * Ci * (H + 2 * pad) * (W + 2 * pad) * sizeof(float), 256);
    long chunk_px;
    int nchunk;
    wgrad_plan((long)B * H * W, Co, Ci * ks * ks, chunk_px, nchunk);
    size_t part = cfd_align_up((size_t)nchunk * Co * Ci * ks * ks * sizeof(float), 256);
    {
        const ConvGeom g{B, Ci, Co, H, W, ks};
        if (cfd_conv6_wgrad_covers(g)) part = cfd_conv6_wgrad_ws_bytes(g);
    }
    if (ks == 1) {
        const size_t c1 = cfd_conv1_wgrad_ws_bytes(B, Ci, Co, H * W);
        if (c1 > part) part = c1;
    }
    const size_t cs = chan_sum_ws_bytes(Co);
    // [extended input gradient | weight fragments and split-K partials of the input-gradient pass (conv6.hip)]; the
    // weight-gradient partials and the bias sums reuse the front of the buffer afterwards
    const ConvGeom gg{B, Ci, Co, H, W, ks};
    const size_t dg = ext + (cfd_conv6_covers(gg, true) ? cfd_conv6_ws_bytes(gg, true) : 0);
    // conv6 route: the weight-gradient partials live BEHIND the input-gradient region (their reduction rides in the border-fold
    // launch, which still reads the extended buffer)
    const size_t m = (cfd_conv6_covers(gg, true) && cfd_conv6_wgrad_covers(gg)) ? dg + part : (dg > part ? dg : part);
    return m > cs ? m : cs;
}

// gin (B,Ci,H,W), gw (Co,Ci,ks,ks), gb (Co) from gout (B,Co,H,W); any of gin / gw / gb may be NULL.  wfrag_t: NULL or the weights'
// input-gradient fragments (cfd_conv2d_wprep_batch, transposed = 1).
extern "C" int cfd_conv2d_bwd_ex(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                                 const void* wfrag_t, int B, int Ci, int Co, int H, int W, int ks, void* stream);

extern "C" int cfd_conv2d_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                              int B, int Ci, int Co, int H, int W, int ks, void* stream) {
    return cfd_conv2d_bwd_ex(gout, in, w, gin, gw, gb, ws, nullptr, B, Ci, Co, H, W, ks, stream);
}

extern "C" int cfd_conv2d_bwd_ex(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                                 const void* wfrag_t, int B, int Ci, int Co, int H, int W, int ks, void* stream) {
    CFD_REQUIRE(gout && in && w && ws, CFD_ERR_INVALID_ARG, "cfd_conv2d_bwd: NULL pointer");
    CFD_TRY(conv_check("cfd_conv2d_bwd", B, Ci, Co, H, W, ks));
    CFD_REQUIRE(B >= 1, CFD_ERR_INVALID_ARG, "cfd_conv2d_bwd: empty batch");
    hipStream_t st = (hipStream_t)stream;
    const ConvGeom g{B, Ci, Co, H, W, ks};
    const int HW = H * W, pad = ks / 2;
    if (ks == 1 && cfd_tune_get(CFD_TUNE_CONV1_MFMA) != 0 && cfd_conv1_covers(B, Ci, Co, HW)) {  // conv1.hip: no extended grid, no fold
        if (gin) {
            CFD_PROF_W("k_conv_dgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci), 2.0 * B * HW * (double)Co * Ci);
            CFD_TRY(cfd_conv1_dgrad(gout, w, gin, B, Ci, Co, HW, st, "cfd_conv2d_bwd(dgrad)"));
            gin = nullptr;
        }
        if (gw && cfd_conv1_wgrad_covers(B, Ci, Co, HW) && !((size_t)in & 15) && !((size_t)gout & 15)) {
            CFD_PROF_W("k_conv_wgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci), 2.0 * B * HW * (double)Co * Ci);
            CFD_TRY(cfd_conv1_wgrad(gout, in, gw, gb, ws, B, Ci, Co, HW, st, "cfd_conv2d_bwd(wgrad)"));
            gw = nullptr, gb = nullptr;
        }
    }
    if (gin) {
        float* ext = (float*)ws;
        bool direct = false;
        // side_stream bit 4 (round 6 experiment): the weight gradient on the library's side stream BESIDE the input gradient -- neither
        // depends on the other, and on the deep U-Net levels (16 x 16 and below) neither fills the chip; forked BEFORE the input gradient
        // is enqueued, joined in front of the fold launch that carries the weight gradient's partial-sum reduction
        hipStream_t side = (gw && cfd_conv6_wgrad_covers(g)) ? cfd_side_fork(st, 4) : st;
        {
            CFD_PROF_W("k_conv_dgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks),
                       2.0 * B * HW * (double)Co * Ci * ks * ks);
            const size_t ext_bytes = cfd_align_up((size_t)B * Ci * (H + 2 * pad) * (W + 2 * pad) * sizeof(float), 256);
            if (cfd_conv6_covers(g, true)) {
                CFD_TRY(cfd_conv6_run(gout, w, nullptr, ext, (char*)ws + ext_bytes, g, true, gin, &direct, st, "cfd_conv2d_bwd(dgrad)",
                                      nullptr, wfrag_t));
            } else {
                CFD_TRY(launch_conv_gather<true>(gout, w, nullptr, ext, g, st, "cfd_conv2d_bwd(dgrad)"));
            }
        }
        // conv6 route with a direct interior: the weight gradient runs BEFORE the border fold and leaves its partial-sum reduction to
        // the fold launch's first workgroups (one launch less; the partials sit behind the input-gradient region of ws)
        CfdPartReduceJob job{};
        if (direct && gw && cfd_conv6_wgrad_covers(g)) {
            const size_t ext_bytes = cfd_align_up((size_t)B * Ci * (H + 2 * pad) * (W + 2 * pad) * sizeof(float), 256);
            void* wws = (char*)ws + ext_bytes + cfd_conv6_ws_bytes(g, true);
            {
                CFD_PROF_W("k_conv_wgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks),
                           2.0 * B * HW * (double)Co * Ci * ks * ks);
                CFD_TRY(cfd_conv6_wgrad(gout, in, gw, gb, wws, g, side, "cfd_conv2d_bwd(wgrad)", &job));
            }
            gw = nullptr, gb = nullptr;  // final once the fold launch below has run
        }
        CFD_TRY(cfd_side_join(st, side));
        CFD_PROF_W("k_fold_pad", st, 0.0, 0.0);  // pure data movement of the extended-grid formulation: no algorithmic bytes
        if (direct) {
            const long total = (long)B * Ci * (2 * W + 2 * (H - 2));
            CFD_REQUIRE_I31(total, "cfd_conv2d_bwd");
            long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(k_fold_border, dim3((unsigned)(blocks + job.nblk)), dim3(256), 0, st, (const float*)ext, gin, (unsigned)total,
                               H, W, pad, cfd_div_make((unsigned)(2 * W + 2 * (H - 2))), job);
        } else {
            const long total = (long)B * Ci * HW;
            long blocks = (total + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            CFD_REQUIRE_I31(total, "cfd_conv2d_bwd");
            hipLaunchKernelGGL(k_fold_pad, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ext, gin, (unsigned)total, H, W,
                               pad, cfd_div_make((unsigned)HW), cfd_div_make((unsigned)W));
        }
        CFD_LAUNCH_CHECK("cfd_conv2d_bwd(fold)");
    }
    if (gw && cfd_conv6_wgrad_covers(g)) {
        CFD_PROF_W("k_conv_wgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks), 2.0 * B * HW * (double)Co * Ci * ks * ks);
        CFD_TRY(cfd_conv6_wgrad(gout, in, gw, gb, ws, g, st, "cfd_conv2d_bwd(wgrad)"));
        gb = nullptr;  // done in the same launches
    } else if (gw) {
        long chunk_px;
        int nchunk;
        const int J = Ci * ks * ks;
        wgrad_plan((long)B * HW, Co, J, chunk_px, nchunk);
        const dim3 grid(nchunk, (Co + 16 * CW_MT - 1) / (16 * CW_MT), (J + 16 * CW_NT - 1) / (16 * CW_NT));
        {
            CFD_PROF_W("k_conv_wgrad", st, 4.0 * ((double)B * (Ci + Co) * HW + (double)Co * Ci * ks * ks),
                       2.0 * B * HW * (double)Co * Ci * ks * ks);
            CFD_REQUIRE_I31((long)B * HW, "cfd_conv2d_bwd");
            hipLaunchKernelGGL((k_conv_wgrad<false>), grid, dim3(64 * CV_WAVES), 0, st, gout, in, (float*)ws, g,
                               (unsigned)chunk_px, cfd_div_make((unsigned)HW), cfd_div_make((unsigned)W));
        }
        CFD_LAUNCH_CHECK("cfd_conv2d_bwd(wgrad)");
        const long n = (long)Co * J;
        CFD_PROF_W("k_part_reduce", st, 0.0, 0.0);  // partial sums are an implementation detail
        cfd_conv_part_reduce((const float*)ws, gw, n, nchunk, st);
        CFD_LAUNCH_CHECK("cfd_conv2d_bwd(reduce)");
    }
    if (gb) CFD_TRY(chan_sum(gout, gb, ws, B, Co, HW, st, "cfd_conv2d_bwd(bias)"));
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// BatchNorm2d (+ ReLU)   (src/models/unet.py:28-30; torch defaults eps = 1e-5, momentum = 0.1)
// ------------------------------------------------------------------------------------------------------
// Per-channel statistics over (B, H, W) in two deterministic stages: grid (C, nsplit) partials, then one wave per
// channel.  Mean and variance come from ONE pass over shifted values (x - x[first element of the channel]), which keeps
// fp32 accurate when |mean| >> std without the second read of the two-pass form.
#define BN_SPLIT 64  // partial sums per channel (one lane of k_bn_final each)

// MODE 0: sum x;  MODE 1: sum (x - mean[c])^2;  MODE 2: sums of gz and gz*xhat with gz = gy * (y > 0 if relu);
// MODE 3: sum (x - K) and sum (x - K)^2 with the shift K = the channel's first element (one-pass mean + variance: the
// shift is a sample of the channel, so |mean - K| ~ std and the cancellation in E[d^2] - E[d]^2 costs ~1 ulp)
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ x, const float* __restrict__ aux,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ part, int B, int C, int HW, int relu,
                                                    CfdDiv dHW) {
    __shared__ float s_r[8];
    const int c = blockIdx.x, sp = blockIdx.y;
    const unsigned n = (unsigned)B * HW;  // < 2^31 (checked by the callers)
    const unsigned per = (n + BN_SPLIT - 1) / BN_SPLIT;
    const unsigned e0 = sp * per, e1 = e0 + per < n ? e0 + per : n;
    const float mu = MODE == 3 ? x[(size_t)c * HW] : (MODE >= 1 ? mean[c] : 0.f), rs = MODE == 2 ? rstd[c] : 0.f;
    const float ga = (MODE == 2 && relu) ? gamma[c] : 0.f, be = (MODE == 2 && relu) ? beta[c] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    auto item = [&](unsigned e, float v, float a) {
        (void)e;
        if (MODE == 0) s0 += v;
        else if (MODE == 1) { const float d = v - mu; s0 = fmaf(d, d, s0); }
        else if (MODE == 3) { const float d = v - mu; s0 += d; s1 = fmaf(d, d, s1); }
        else {
            // x = layer input, aux = upstream gradient; relu mask from the normalised output sign
            const float xh = (v - mu) * rs;
            float gz = a;
            if (relu) {
                const float yv = fmaf(xh, ga, be);
                gz = yv > 0.f ? gz : 0.f;
            }
            s0 += gz;
            s1 = fmaf(gz, xh, s1);
        }
    };
    auto offset = [&](unsigned e) {
        const unsigned b = cfd_div(e, dHW);
        return ((size_t)b * C + c) * HW + (e - b * (unsigned)HW);
    };
    unsigned e = e0 + threadIdx.x;
    for (; e + 3 * blockDim.x < e1; e += 4 * blockDim.x) {  // four independent loads in flight per thread
        size_t o[4];
        float v[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = offset(e + k * blockDim.x);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = x[o[k]]; a[k] = MODE == 2 ? aux[o[k]] : 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) item(e + k * blockDim.x, v[k], a[k]);
    }
    for (; e < e1; e += blockDim.x) {
        const size_t o = offset(e);
        item(e, x[o], MODE == 2 ? aux[o] : 0.f);
    }
    s0 = cfd_wave_sum(s0);
    s1 = cfd_wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_r[wave] = s0; s_r[4 + wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)c * BN_SPLIT + sp) * 2] = (s_r[0] + s_r[1]) + (s_r[2] + s_r[3]);
        part[((size_t)c * BN_SPLIT + sp) * 2 + 1] = (s_r[4] + s_r[5]) + (s_r[6] + s_r[7]);
    }
}

// stage 0 -> mean;  stage 1 -> rstd (+ running statistics);  stage 2 -> (gbeta, ggamma);
// stage 3 -> mean and rstd (+ running statistics) from the shifted sums of MODE 3 (shift[c] = x[c*HW])
__global__ __launch_bounds__(64) void k_bn_final(const float* __restrict__ part, int stage, float count, float eps,
                                                 float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                 float* __restrict__ run_mean, float* __restrict__ run_var,
                                                 float* __restrict__ gbeta, float* __restrict__ ggamma, int C,
                                                 const float* __restrict__ x, int HW) {
    const int c = blockIdx.x, lane = threadIdx.x;
    float a = lane < BN_SPLIT ? part[((size_t)c * BN_SPLIT + lane) * 2] : 0.f;
    float b = lane < BN_SPLIT ? part[((size_t)c * BN_SPLIT + lane) * 2 + 1] : 0.f;
    a = cfd_wave_sum(a);
    b = cfd_wave_sum(b);
    if (lane != 0) return;
    if (stage == 3) {
        const float d = a / count;
        const float mu = x[(size_t)c * HW] + d;
        float m2 = b - a * d;  // sum (x - mean)^2 = sum (x-K)^2 - n (mean-K)^2
        m2 = m2 > 0.f ? m2 : 0.f;
        const float var = m2 / count;
        mean[c] = mu;
        rstd[c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            const float unb = count > 1.f ? m2 / (count - 1.f) : var;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
        }
    } else if (stage == 0) mean[c] = a / count;
    else if (stage == 1) {
        const float var = a / count;  // biased: what normalises (torch.nn.functional.batch_norm, training=True)
        rstd[c] = 1.0f / sqrtf(var + eps);
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
            const float unb = count > 1.f ? a / (count - 1.f) : var;  // running_var tracks the unbiased estimate
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
        }
    } else {
        gbeta[c] = a;
        ggamma[c] = b;
    }
    (void)C;
}

static size_t chan_sum_ws_bytes(int C) { return cfd_align_up(((size_t)C * BN_SPLIT * 2 + C) * sizeof(float), 256); }

static int chan_sum(const float* g, float* out, void* ws, int B, int C, int HW, hipStream_t st, const char* what) {
    float* part = (float*)ws;
    float* dummy = part + (size_t)C * BN_SPLIT * 2;
    hipLaunchKernelGGL((k_bn_partial<0>), dim3(C, BN_SPLIT), dim3(256), 0, st, g, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, B, C, HW,
                       0, cfd_div_make((unsigned)HW));
    hipLaunchKernelGGL(k_bn_final, dim3(C), dim3(64), 0, st, (const float*)part, 2, 1.f, 0.f, 0.f, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, out, dummy, C, (const float*)nullptr, 0);
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

// Sum of the BN_SPLIT partial pairs of channel c by wave 0 (the arithmetic of k_bn_final), broadcast through LDS.
__device__ __forceinline__ void bn_partials(const float* __restrict__ part, int c, float* s_ab, float& a, float& b,
                                            int nsplit = BN_SPLIT) {
    // every thread takes the pairs t, t + 256, ... (all loads in flight at once: a one-wave loop over 512 conv-emitted pairs cost
    // eight dependent memory latencies, +5 us per launch), then wave sums and the four waves in a fixed order.  With
    // nsplit == BN_SPLIT wave 0 holds one pair per lane and the others add zeros: the arithmetic of k_bn_final.
    float pa = 0.f, pb = 0.f;
    for (int i = threadIdx.x; i < nsplit; i += blockDim.x) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)c * nsplit + i) * 2);
        pa += v.x;
        pb += v.y;
    }
    pa = cfd_wave_sum(pa);
    pb = cfd_wave_sum(pb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_ab[2 + 2 * wave] = pa; s_ab[3 + 2 * wave] = pb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = s_ab[2], tb = s_ab[3];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { ta += s_ab[2 + 2 * w]; tb += s_ab[3 + 2 * w]; }
        s_ab[0] = ta;
        s_ab[1] = tb;
    }
    __syncthreads();
    a = s_ab[0];
    b = s_ab[1];
}

#define BN_EW 4  // elements per thread and trip of the normalising kernels (their loads in flight together)
// Block sums of two per-thread values: wave sums, then the four waves in a fixed order through s_ab (10 floats).
__device__ __forceinline__ void bn_block_sum2(float pa, float pb, float* s_ab, float& a, float& b) {
    pa = cfd_wave_sum(pa);
    pb = cfd_wave_sum(pb);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_ab[2 + 2 * wave] = pa; s_ab[3 + 2 * wave] = pb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = s_ab[2], tb = s_ab[3];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { ta += s_ab[2 + 2 * w]; tb += s_ab[3 + 2 * w]; }
        s_ab[0] = ta;
        s_ab[1] = tb;
    }
    __syncthreads();
    a = s_ab[0];
    b = s_ab[1];
}

// Mean and sum of squared deviations of channel c from per-slot records (m, m2, n, -) = mean of x - shift, sum of squared deviations
// from it and element count of the slot (cfd_conv2d_fwd_stats).  With r = the first slot's mean as the common reference:
//   A = sum_s n_s (m_s - r),  Q = sum_s [m2_s + n_s (m_s - r)^2];   mean - shift = r + A / N,  sum (x - mean)^2 = Q - A^2 / N
// -- the slot means scatter around r by no more than the data do, so Q - A^2 / N is a difference of comparable numbers only when the
// variance really is that small against the scatter (one pass and one block reduction, like the plain sums).  Returns d = mean - shift
// and m2.
__device__ __forceinline__ void bn_slot_stats(const float* __restrict__ rec, int c, float* s_ab, int nslots, float count, float& d,
                                              float& m2) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(rec) + (size_t)c * nslots;
    // (a per-lane load on purpose: as a scalar load of a uniform address the compiler waits for it before it issues the record loads
    //  below -- one more trip to memory)
    const float r = rec[((size_t)c * nslots + cfd_opaque(0)) * 4];
    float pa = 0.f, pq = 0.f;
    for (int i = threadIdx.x; i < nslots; i += blockDim.x) {
        const f32x4 v = r4[i];
        const float e = v[0] - r, ne = v[2] * e;
        pa += ne;
        pq += fmaf(ne, e, v[1]);
    }
    float a, qq;
    bn_block_sum2(pa, pq, s_ab, a, qq);
    const float da = a / count;
    d = r + da;
    m2 = qq - a * da;
}

// y = [relu]((x - mean) * rstd * gamma + beta) with the statistics finished in the same launch: workgroup (c, sp) sums the
// channel's partial pairs of k_bn_partial<3> itself (training) or reads the running statistics (eval) and normalises slice sp
// of the channel; slice 0 also writes save_mean / save_rstd and updates the running statistics.  (Round 2 had a one-workgroup
// k_bn_final launch per reduction: 59 launches of ~4.5 us per U-Net step, 6 % of it.)
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ part,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float* __restrict__ run_mean, float* __restrict__ run_var,
                                                  float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                  float* __restrict__ y, int B, int C, int HW, float count, float eps,
                                                  float momentum, int training, int relu, CfdDiv dHW, int nsplit,
                                                  const float* __restrict__ kbase, int kstride, int records) {
    // records: `part` holds the (m, m2, n, -) slot records of cfd_conv2d_fwd_stats (kbase = the producer's shift, stride kstride);
    // otherwise the (sum, sum of squares) partials of k_bn_partial<3> about kbase[c * kstride] (an explicit mode since round 5:
    // it used to be encoded as kstride == 0)
    __shared__ float s_ab[10];
    const int c = blockIdx.x, sp = blockIdx.y;
    // Everything that does not depend on the statistics is REQUESTED first -- the per-channel parameters and the slice's first batch
    // of elements: each of them behind the reduction was one more trip to memory during which the whole grid (every workgroup runs
    // the same prologue at the same time) had nothing in flight (gamma / beta as scalar loads behind the last barrier: +2.8 us on a
    // 15 us launch).
    const float ga = gamma[c], be = beta[c];
    const float kb = (training && kbase) ? kbase[(size_t)c * kstride] : 0.f;
    const unsigned n = (unsigned)B * HW;
    const unsigned per = (n + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = sp * per, e1 = e0 + per < n ? e0 + per : n;
    const bool any = e0 < e1;
    size_t o[BN_EW];
    float xv[BN_EW];
    const auto fetch = [&](unsigned eb) {  // elements eb + k blockDim.x (clamped into the slice: no conditional loads)
#pragma unroll
        for (int k = 0; k < BN_EW; ++k) {
            unsigned e = eb + k * blockDim.x;
            e = e < e1 ? e : e1 - 1;
            const unsigned b = cfd_div(e, dHW);
            o[k] = ((size_t)b * C + c) * HW + (e - b * (unsigned)HW);
            xv[k] = x[o[k]];
        }
    };
    if (any) fetch(e0 + threadIdx.x);
    float mu, rs;
    if (training) {
        float d, m2;
        if (records) {  // per-slot records of the conv that produced x (cfd_conv2d_fwd_stats); kbase = its bias
            bn_slot_stats(part, c, s_ab, nsplit, count, d, m2);
        } else {  // sums of (x - K) and (x - K)^2, K = the channel's first element (k_bn_partial<3>)
            float a, b;
            bn_partials(part, c, s_ab, a, b, nsplit);
            d = a / count;
            m2 = b - a * d;  // sum (x - mean)^2 = sum (x-K)^2 - n (mean-K)^2
        }
        mu = kb + d;
        m2 = m2 > 0.f ? m2 : 0.f;
        const float var = m2 / count;  // biased: what normalises (torch.nn.functional.batch_norm, training=True)
        rs = 1.0f / sqrtf(var + eps);
        if (sp == 0 && threadIdx.x == 0 && run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
            const float unb = count > 1.f ? m2 / (count - 1.f) : var;  // running_var tracks the unbiased estimate
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
        }
    } else {
        mu = run_mean[c];
        rs = 1.0f / sqrtf(run_var[c] + eps);
    }
    if (sp == 0 && threadIdx.x == 0) { save_mean[c] = mu; save_rstd[c] = rs; }
    if (!any) return;
    for (unsigned eb = e0 + threadIdx.x;;) {
#pragma unroll
        for (int k = 0; k < BN_EW; ++k) {
            float v = fmaf((xv[k] - mu) * rs, ga, be);
            if (relu) v = v > 0.f ? v : 0.f;
            if (eb + k * blockDim.x < e1) y[o[k]] = v;
        }
        eb += BN_EW * blockDim.x;
        if (eb - threadIdx.x >= e1) break;  // (uniform)
        fetch(eb);
    }
}

// gx = gamma * rstd * (gz - [gbeta/N + xhat * ggamma/N] if training); (gbeta, ggamma) = the sums of k_bn_partial<2>, finished here
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ gy,
                                                      const float* __restrict__ part, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ gbeta,
                                                      float* __restrict__ ggamma, float* __restrict__ gx, int B, int C, int HW,
                                                      float inv_count, int relu, int training, CfdDiv dHW) {
    __shared__ float s_ab[10];
    const int c = blockIdx.x, sp = blockIdx.y;
    // (parameters and the slice's first batch requested ahead of the reduction, as in k_bn_apply)
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    const unsigned n = (unsigned)B * HW;
    const unsigned per = (n + gridDim.y - 1) / gridDim.y;
    const unsigned e0 = sp * per, e1 = e0 + per < n ? e0 + per : n;
    const bool any = e0 < e1;
    size_t o[BN_EW];
    float xv[BN_EW], gv[BN_EW];
    const auto fetch = [&](unsigned eb) {
#pragma unroll
        for (int k = 0; k < BN_EW; ++k) {
            unsigned e = eb + k * blockDim.x;
            e = e < e1 ? e : e1 - 1;
            const unsigned b = cfd_div(e, dHW);
            o[k] = ((size_t)b * C + c) * HW + (e - b * (unsigned)HW);
            xv[k] = x[o[k]];
            gv[k] = gy[o[k]];
        }
    };
    if (any) fetch(e0 + threadIdx.x);
    float gb, gg;
    bn_partials(part, c, s_ab, gb, gg);
    if (sp == 0 && threadIdx.x == 0) { gbeta[c] = gb; ggamma[c] = gg; }
    if (!any) return;
    for (unsigned eb = e0 + threadIdx.x;;) {
#pragma unroll
        for (int k = 0; k < BN_EW; ++k) {
            const float xh = (xv[k] - mu) * rs;
            float gz = gv[k];
            if (relu && !(fmaf(xh, ga, be) > 0.f)) gz = 0.f;
            float t = gz;
            if (training) t -= (gb + xh * gg) * inv_count;
            if (eb + k * blockDim.x < e1) gx[o[k]] = ga * rs * t;
        }
        eb += BN_EW * blockDim.x;
        if (eb - threadIdx.x >= e1) break;  // (uniform)
        fetch(eb);
    }
}

// slices per channel of the normalising kernels: ~16 k elements each, enough workgroups to fill the chip
static unsigned bn_slices(int B, int C, int HW) {
    long per_c = (long)B * HW, s = (per_c + 16383) / 16384;
    while (s * C < 1024 && s * 2048 < per_c) s *= 2;
    return (unsigned)(s < 1 ? 1 : (s > 1024 ? 1024 : s));
}

static unsigned ew_blocks(long total) {
    long b = (total + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

extern "C" size_t cfd_batchnorm_workspace_bytes(int C) {
    // [partials C*SPLIT*2 | packed (mean, gamma, beta) 3C] floats
    return cfd_align_up(((size_t)C * BN_SPLIT * 2 + 3 * (size_t)C) * sizeof(float), 256);
}

// y = [relu](batch_norm(x)).  training != 0: batch statistics (saved into save_mean / save_rstd, running statistics
// updated when non-NULL); training == 0: running statistics.  x, y: (B,C,H*W); everything else: (C).
extern "C" int cfd_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                 float* y, float* save_mean, float* save_rstd, void* ws, int B, int C, int HW, float eps,
                                 float momentum, int training, int relu, void* stream) {
    CFD_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && ws, CFD_ERR_INVALID_ARG, "cfd_batchnorm_fwd: NULL pointer");
    CFD_REQUIRE(B >= 1 && C >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_batchnorm_fwd: bad sizes");
    CFD_REQUIRE_I31((long)B * C * HW, "cfd_batchnorm_fwd");
    CFD_REQUIRE(training || (run_mean && run_var), CFD_ERR_INVALID_ARG, "cfd_batchnorm_fwd: eval mode needs running statistics");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    const float count = (float)((double)B * HW);
    if (training) {
        CFD_PROF_W("k_bn_stats", st, 4.0 * B * C * HW, 3.0 * B * C * HW);
        hipLaunchKernelGGL((k_bn_partial<3>), dim3(C, BN_SPLIT), dim3(256), 0, st, x, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, part, B, C,
                           HW, 0, cfd_div_make((unsigned)HW));
        CFD_LAUNCH_CHECK("cfd_batchnorm_fwd(stats)");
    }
    CFD_PROF_W("k_bn_apply", st, 8.0 * B * C * HW, 2.0 * B * C * HW);
    hipLaunchKernelGGL(k_bn_apply, dim3(C, bn_slices(B, C, HW)), dim3(256), 0, st, x, (const float*)part, gamma, beta, run_mean,
                       run_var, save_mean, save_rstd, y, B, C, HW, count, eps, momentum, training, relu,
                       cfd_div_make((unsigned)HW), BN_SPLIT, x, HW, 0);
    CFD_LAUNCH_CHECK("cfd_batchnorm_fwd(apply)");
    return CFD_OK;
}

// Training-mode forward from the records that the producer of x emitted (cfd_conv2d_fwd_stats): stats (C, slots, 4) = (m, m2,
// n, -) = mean of x - shift[c], sum of squared deviations from it and element count of each slot (disjoint sets covering the channel;
// shift NULL: 0).
// One launch; otherwise cfd_batchnorm_fwd(training = 1).
extern "C" int cfd_batchnorm_fwd_stats(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var,
                                       float* y, float* save_mean, float* save_rstd, const float* stats, int slots,
                                       const float* shift, int B, int C, int HW, float eps, float momentum, int relu, void* stream) {
    CFD_REQUIRE(x && gamma && beta && y && save_mean && save_rstd && stats, CFD_ERR_INVALID_ARG, "cfd_batchnorm_fwd_stats: NULL pointer");
    CFD_REQUIRE(B >= 1 && C >= 1 && HW >= 1 && slots >= 1, CFD_ERR_INVALID_ARG, "cfd_batchnorm_fwd_stats: bad sizes");
    CFD_REQUIRE_I31((long)B * C * HW, "cfd_batchnorm_fwd_stats");
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W("k_bn_apply", st, 8.0 * B * C * HW, 2.0 * B * C * HW);
    hipLaunchKernelGGL(k_bn_apply, dim3(C, bn_slices(B, C, HW)), dim3(256), 0, st, x, stats, gamma, beta, run_mean, run_var,
                       save_mean, save_rstd, y, B, C, HW, (float)((double)B * HW), eps, momentum, 1, relu, cfd_div_make((unsigned)HW),
                       slots, shift, 1, 1);
    CFD_LAUNCH_CHECK("cfd_batchnorm_fwd_stats");
    return CFD_OK;
}

// gx, ggamma, gbeta from gy (gradient on the [relu]'d output), the layer input x and the saved statistics.
extern "C" int cfd_batchnorm_bwd(const float* gy, const float* x, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_rstd, float* gx, float* ggamma, float* gbeta,
                                 void* ws, int B, int C, int HW, int training, int relu, void* stream) {
    CFD_REQUIRE(gy && x && gamma && beta && save_mean && save_rstd && gx && ggamma && gbeta && ws, CFD_ERR_INVALID_ARG,
                "cfd_batchnorm_bwd: NULL pointer");
    CFD_REQUIRE(B >= 1 && C >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_batchnorm_bwd: bad sizes");
    CFD_REQUIRE_I31((long)B * C * HW, "cfd_batchnorm_bwd");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)ws;
    {
        CFD_PROF_W("k_bn_bwd_reduce", st, 8.0 * B * C * HW, 4.0 * B * C * HW);
        hipLaunchKernelGGL((k_bn_partial<2>), dim3(C, BN_SPLIT), dim3(256), 0, st, x, gy, save_mean, save_rstd, gamma, beta, part,
                           B, C, HW, relu, cfd_div_make((unsigned)HW));
    }
    CFD_LAUNCH_CHECK("cfd_batchnorm_bwd(reduce)");
    CFD_PROF_W("k_bn_bwd_apply", st, 12.0 * B * C * HW, 4.0 * B * C * HW);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(C, bn_slices(B, C, HW)), dim3(256), 0, st, x, gy, (const float*)part, save_mean,
                       save_rstd, gamma, beta, gbeta, ggamma, gx, B, C, HW, (float)(1.0 / ((double)B * HW)), relu, training,
                       cfd_div_make((unsigned)HW));
    CFD_LAUNCH_CHECK("cfd_batchnorm_bwd(apply)");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// MaxPool2d(2)  (src/models/unet.py:59): floor mode, gradient to the FIRST maximum in row-major window order
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool2(const float* __restrict__ x, float* __restrict__ y, unsigned total, int H,
                                                  int W, CfdDiv dHWo, CfdDiv dWo) {
    const int Ho = H / 2, Wo = W / 2;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHWo);
        const int p = (int)(e - img * (unsigned)(Ho * Wo)), yo = (int)cfd_div((unsigned)p, dWo), xo = p - yo * Wo;
        const float* s = x + (size_t)img * H * W + (2 * yo) * W + 2 * xo;
        float m = s[0];
        m = s[1] > m ? s[1] : m;
        m = s[W] > m ? s[W] : m;
        m = s[W + 1] > m ? s[W + 1] : m;
        y[e] = m;
    }
}

// `add` (optional): a second gradient of the pooled tensor's INPUT, (B, C, H, W) with batch stride add_bs floats -- the skip
// connection's share, read in place from the gradient of the decoder's concatenation (unet.py:80-88) -- summed in the same pass
// instead of a contiguous copy of the slice + autograd's add kernel.
__global__ __launch_bounds__(256) void k_maxpool2_bwd(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ gx, unsigned total, int H, int W, CfdDiv dHW,
                                                      CfdDiv dW, const float* __restrict__ add, unsigned add_bs, int C, CfdDiv dC) {
    const int Ho = H / 2, Wo = W / 2;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHW);
        const int p = (int)(e - img * (unsigned)(H * W)), yy = (int)cfd_div((unsigned)p, dW), xx = p - yy * W;
        const int yo = yy / 2, xo = xx / 2;
        float g = 0.f;
        if (yo < Ho && xo < Wo) {  // rows / columns cut off by floor mode get no gradient
            const float* s = x + (size_t)img * H * W + (2 * yo) * W + 2 * xo;
            int arg = 0;
            float m = s[0];
            if (s[1] > m) { m = s[1]; arg = 1; }
            if (s[W] > m) { m = s[W]; arg = 2; }
            if (s[W + 1] > m) { m = s[W + 1]; arg = 3; }
            if (arg == (yy - 2 * yo) * 2 + (xx - 2 * xo)) g = gy[(size_t)img * Ho * Wo + yo * Wo + xo];
        }
        if (add) {
            const unsigned b = cfd_div(img, dC), c = img - b * (unsigned)C;
            g += add[(size_t)b * add_bs + (size_t)c * (H * W) + p];
        }
        gx[e] = g;
    }
}

// The same for even H and W with one 2 x 2 window per thread: two 8-byte loads of x, the window's gradient once, 8-byte stores (the
// element-per-thread kernel above reads each window four times and issues four times the instructions: 15 us per U-Net level).
// Same selection rule (first maximum in row-major order), same single add: bitwise the element-per-thread results.
__global__ __launch_bounds__(256) void k_maxpool2_bwd_win(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ gx, unsigned nwin, int H, int W, CfdDiv dHWo, CfdDiv dWo,
                                                          const float* __restrict__ add, unsigned add_bs, int C, CfdDiv dC) {
    const int Ho = H / 2, Wo = W / 2;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < nwin; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHWo);
        const int p = (int)(e - img * (unsigned)(Ho * Wo)), yo = (int)cfd_div((unsigned)p, dWo), xo = p - yo * Wo;
        const size_t o = (size_t)img * H * W + (size_t)(2 * yo) * W + 2 * xo;
        const float2 r0 = *reinterpret_cast<const float2*>(x + o), r1 = *reinterpret_cast<const float2*>(x + o + W);
        int arg = 0;
        float m = r0.x;
        if (r0.y > m) { m = r0.y; arg = 1; }
        if (r1.x > m) { m = r1.x; arg = 2; }
        if (r1.y > m) { m = r1.y; arg = 3; }
        const float g = gy[e];
        float2 g0 = make_float2(arg == 0 ? g : 0.f, arg == 1 ? g : 0.f), g1 = make_float2(arg == 2 ? g : 0.f, arg == 3 ? g : 0.f);
        if (add) {
            const unsigned b = cfd_div(img, dC), c = img - b * (unsigned)C;
            const float* a = add + (size_t)b * add_bs + (size_t)c * (H * W) + (size_t)(2 * yo) * W + 2 * xo;
            const float2 a0 = *reinterpret_cast<const float2*>(a), a1 = *reinterpret_cast<const float2*>(a + W);
            g0.x += a0.x; g0.y += a0.y; g1.x += a1.x; g1.y += a1.y;
        }
        *reinterpret_cast<float2*>(gx + o) = g0;
        *reinterpret_cast<float2*>(gx + o + W) = g1;
    }
}

// launches the window kernel where it applies (even H, W; 8-byte aligned rows), else the element kernel
static void launch_maxpool2_bwd(const float* x, const float* gy, float* gx, long nimg, int H, int W, const float* add, size_t add_bs, int C,
                                hipStream_t st) {
    const bool win = H % 2 == 0 && W % 2 == 0 && !((size_t)x & 7) && !((size_t)gx & 7) && (!add || (!((size_t)add & 7) && add_bs % 2 == 0));
    if (win) {
        const long nwin = nimg * (H / 2) * (W / 2);
        hipLaunchKernelGGL(k_maxpool2_bwd_win, dim3(ew_blocks(nwin)), dim3(256), 0, st, x, gy, gx, (unsigned)nwin, H, W,
                           cfd_div_make((unsigned)((H / 2) * (W / 2))), cfd_div_make((unsigned)(W / 2)), add, (unsigned)add_bs, C,
                           cfd_div_make((unsigned)C));
    } else {
        hipLaunchKernelGGL(k_maxpool2_bwd, dim3(ew_blocks(nimg * H * W)), dim3(256), 0, st, x, gy, gx, (unsigned)(nimg * H * W), H, W,
                           cfd_div_make((unsigned)(H * W)), cfd_div_make((unsigned)W), add, (unsigned)add_bs, C, cfd_div_make((unsigned)C));
    }
}

extern "C" int cfd_maxpool2_fwd(const float* x, float* y, int nimg, int H, int W, void* stream) {
    CFD_REQUIRE(x && y && nimg >= 0 && H >= 2 && W >= 2, CFD_ERR_INVALID_ARG, "cfd_maxpool2_fwd: bad arguments");
    if (nimg == 0) return CFD_OK;
    CFD_REQUIRE_I31((long)nimg * H * W, "cfd_maxpool2_fwd");
    hipLaunchKernelGGL(k_maxpool2, dim3(ew_blocks((long)nimg * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, y,
                       (unsigned)((long)nimg * (H / 2) * (W / 2)), H, W, cfd_div_make((unsigned)((H / 2) * (W / 2))),
                       cfd_div_make((unsigned)(W / 2)));
    CFD_LAUNCH_CHECK("cfd_maxpool2_fwd");
    return CFD_OK;
}

extern "C" int cfd_maxpool2_bwd(const float* x, const float* gy, float* gx, int nimg, int H, int W, void* stream) {
    CFD_REQUIRE(x && gy && gx && nimg >= 0 && H >= 2 && W >= 2, CFD_ERR_INVALID_ARG, "cfd_maxpool2_bwd: bad arguments");
    if (nimg == 0) return CFD_OK;
    CFD_REQUIRE_I31((long)nimg * H * W, "cfd_maxpool2_bwd");
    launch_maxpool2_bwd(x, gy, gx, nimg, H, W, nullptr, 0, 1, (hipStream_t)stream);
    CFD_LAUNCH_CHECK("cfd_maxpool2_bwd");
    return CFD_OK;
}

extern "C" int cfd_maxpool2_bwd_add(const float* x, const float* gy, const float* add, size_t add_batch_stride, float* gx, int B, int C,
                                    int H, int W, void* stream) {
    CFD_REQUIRE(x && gy && add && gx && B >= 0 && C >= 1 && H >= 2 && W >= 2, CFD_ERR_INVALID_ARG, "cfd_maxpool2_bwd_add: bad arguments");
    CFD_REQUIRE(add_batch_stride >= (size_t)C * H * W && add_batch_stride < (1ul << 31), CFD_ERR_INVALID_ARG,
                "cfd_maxpool2_bwd_add: batch stride %zu of the second gradient", add_batch_stride);
    if (B == 0) return CFD_OK;
    const long total = (long)B * C * H * W;
    CFD_REQUIRE_I31(total, "cfd_maxpool2_bwd_add");
    launch_maxpool2_bwd(x, gy, gx, (long)B * C, H, W, add, add_batch_stride, C, (hipStream_t)stream);
    CFD_LAUNCH_CHECK("cfd_maxpool2_bwd_add");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)  (src/models/unet.py:74-76, the bilinear=True up path)
// ------------------------------------------------------------------------------------------------------
// Source coordinate of output index o: s = o * (in - 1) / (out - 1) (align_corners), i0 = floor(s), i1 = min(i0 + 1, in - 1),
// weights (1 - l, l) with l = s - i0 -- ATen's area_pixel_compute_source_index / compute_source_index_and_lambda in fp32.
struct CfdLerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ CfdLerp cfd_lerp_src(int o, int n_in, float scale) {
    const float s = scale * (float)o;
    CfdLerp r;
    r.i0 = (int)s;
    if (r.i0 > n_in - 1) r.i0 = n_in - 1;
    r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
    float l = s - (float)r.i0;
    l = l < 0.f ? 0.f : (l > 1.f ? 1.f : l);
    r.l1 = l;
    r.l0 = 1.f - l;
    return r;
}

__global__ __launch_bounds__(256) void k_upsample2_fwd(const float* __restrict__ x, float* __restrict__ y, unsigned total,
                                                       int H, int W, float sy, float sx, CfdDiv dHWo, CfdDiv dWo) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHWo);
        const int p = (int)(e - img * (unsigned)(Ho * Wo)), yo = (int)cfd_div((unsigned)p, dWo), xo = p - yo * Wo;
        const CfdLerp a = cfd_lerp_src(yo, H, sy), b = cfd_lerp_src(xo, W, sx);
        const float* s = x + (size_t)img * H * W;
        const float r0 = b.l0 * s[a.i0 * W + b.i0] + b.l1 * s[a.i0 * W + b.i1];
        const float r1 = b.l0 * s[a.i1 * W + b.i0] + b.l1 * s[a.i1 * W + b.i1];
        y[e] = a.l0 * r0 + a.l1 * r1;
    }
}

// gx[i][j] = sum over the output pixels that read input (i, j) of their weight * gy: a gather over the (at most 4 x 4)
// output rows / columns whose source interval touches the input pixel -- no atomics, fixed summation order.
__global__ __launch_bounds__(256) void k_upsample2_bwd(const float* __restrict__ gy, float* __restrict__ gx, unsigned total,
                                                       int H, int W, float sy, float sx, CfdDiv dHW, CfdDiv dW) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned img = cfd_div(e, dHW);
        const int p = (int)(e - img * (unsigned)(H * W)), i = (int)cfd_div((unsigned)p, dW), j = p - i * W;
        // candidate output rows 2i-2 .. 2i+3 (s = o (in-1)/(out-1) < o/2, so floor(s) = i needs o in (2i, 2i+3]; i1 = i one lower)
        float wy[6], wx[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int yo = 2 * i - 2 + k, xo = 2 * j - 2 + k;
            wy[k] = 0.f;
            wx[k] = 0.f;
            if (yo >= 0 && yo < Ho) {
                const CfdLerp a = cfd_lerp_src(yo, H, sy);
                wy[k] = (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
            }
            if (xo >= 0 && xo < Wo) {
                const CfdLerp b = cfd_lerp_src(xo, W, sx);
                wx[k] = (b.i0 == j ? b.l0 : 0.f) + (b.i1 == j ? b.l1 : 0.f);
            }
        }
        const float* g = gy + (size_t)img * Ho * Wo;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 6; ++ky) {
            const int yo = 2 * i - 2 + ky;
            if (wy[ky] == 0.f) continue;
            float row = 0.f;
#pragma unroll
            for (int kx = 0; kx < 6; ++kx) {
                const int xo = 2 * j - 2 + kx;
                if (wx[kx] != 0.f) row = fmaf(wx[kx], g[yo * Wo + xo], row);
            }
            acc = fmaf(wy[ky], row, acc);
        }
        gx[e] = acc;
    }
}

static float upsample2_scale(int n) { return n > 0 ? (float)(n - 1) / (float)(2 * n - 1) : 0.f; }

extern "C" int cfd_upsample2_bilinear_fwd(const float* x, float* y, int nimg, int H, int W, void* stream) {
    CFD_REQUIRE(x && y && nimg >= 0 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_upsample2_bilinear_fwd: bad arguments");
    if (nimg == 0) return CFD_OK;
    CFD_REQUIRE_I31((long)nimg * 4 * H * W, "cfd_upsample2_bilinear_fwd");
    const long total = (long)nimg * 4 * H * W;
    CFD_PROF_W("k_upsample2_fwd", (hipStream_t)stream, 5.0 * nimg * H * W * 4.0, 6.0 * total);
    hipLaunchKernelGGL(k_upsample2_fwd, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, (unsigned)total, H, W,
                       upsample2_scale(H), upsample2_scale(W), cfd_div_make((unsigned)(4 * H * W)), cfd_div_make((unsigned)(2 * W)));
    CFD_LAUNCH_CHECK("cfd_upsample2_bilinear_fwd");
    return CFD_OK;
}

extern "C" int cfd_upsample2_bilinear_bwd(const float* gy, float* gx, int nimg, int H, int W, void* stream) {
    CFD_REQUIRE(gy && gx && nimg >= 0 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_upsample2_bilinear_bwd: bad arguments");
    if (nimg == 0) return CFD_OK;
    CFD_REQUIRE_I31((long)nimg * 4 * H * W, "cfd_upsample2_bilinear_bwd");
    const long total = (long)nimg * H * W;
    CFD_PROF_W("k_upsample2_bwd", (hipStream_t)stream, 5.0 * nimg * H * W * 4.0, 2.0 * 16 * total);
    hipLaunchKernelGGL(k_upsample2_bwd, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, gy, gx, (unsigned)total, H, W,
                       upsample2_scale(H), upsample2_scale(W), cfd_div_make((unsigned)(H * W)), cfd_div_make((unsigned)W));
    CFD_LAUNCH_CHECK("cfd_upsample2_bilinear_bwd");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// ConvTranspose2d(kernel 2, stride 2)  (src/models/unet.py:80): out[b,o,2y+ky,2x+kx] = bias[o] + sum_i in[b,i,y,x] w[i,o,ky,kx]
// ------------------------------------------------------------------------------------------------------
// One thread per INPUT pixel and block of CT_OB output channels: the input value is loaded once per input channel and feeds
// 4 taps x CT_OB outputs from registers; the weights of the block ([Ci][CT_OB][4] floats) sit in LDS and are read as broadcast
// 16-byte vectors.  (The first version had one thread per OUTPUT element: two loads per FMA, 43 us for 38 MB of traffic.)
#define CT_OB 4
__global__ __launch_bounds__(256) void k_convt2_fwd(const float* __restrict__ in, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int B, int Ci,
                                                    int Co, int H, int W, CfdDiv dHW, CfdDiv dW) {
    CFD_DYN_SHARED(float4, s_w4);  // [Ci][CT_OB] x (4 taps)
    const int o0 = blockIdx.y * CT_OB, HW = H * W;
    for (int i = threadIdx.x; i < Ci * CT_OB; i += blockDim.x) {
        const int ci = i / CT_OB, ob = i - ci * CT_OB;
        const float* wp = w + ((size_t)ci * Co + o0 + ob) * 4;
        s_w4[i] = o0 + ob < Co ? make_float4(wp[0], wp[1], wp[2], wp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const unsigned total = (unsigned)B * HW;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned b = cfd_div(e, dHW);
        const int p = (int)(e - b * (unsigned)HW), y = (int)cfd_div((unsigned)p, dW), x = p - y * W;
        const float* ip = in + (size_t)b * Ci * HW + p;
        float acc[CT_OB][4];
#pragma unroll
        for (int ob = 0; ob < CT_OB; ++ob) {
            const float bv = (bias && o0 + ob < Co) ? bias[o0 + ob] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[ob][k] = bv;
        }
#pragma unroll 4
        for (int ci = 0; ci < Ci; ++ci) {
            const float v = ip[(size_t)ci * HW];
#pragma unroll
            for (int ob = 0; ob < CT_OB; ++ob) {
                const float4 wv = s_w4[ci * CT_OB + ob];
                acc[ob][0] = fmaf(v, wv.x, acc[ob][0]);
                acc[ob][1] = fmaf(v, wv.y, acc[ob][1]);
                acc[ob][2] = fmaf(v, wv.z, acc[ob][2]);
                acc[ob][3] = fmaf(v, wv.w, acc[ob][3]);
            }
        }
#pragma unroll
        for (int ob = 0; ob < CT_OB; ++ob) {
            if (o0 + ob < Co) {
                float* op = out + ((size_t)b * Co + o0 + ob) * 4 * HW + (size_t)(2 * y) * (2 * W) + 2 * x;
                *reinterpret_cast<float2*>(op) = make_float2(acc[ob][0], acc[ob][1]);
                *reinterpret_cast<float2*>(op + 2 * W) = make_float2(acc[ob][2], acc[ob][3]);
            }
        }
    }
}

// gin[b][i][y][x] = sum_o sum_taps g[b][o][2y+ky][2x+kx] w[i][o][ky][kx]: one thread per input pixel and block of CT_OB INPUT
// channels; the 2x2 patch of the upstream gradient is loaded once per output channel (two 8-byte loads) and feeds CT_OB sums.
__global__ __launch_bounds__(256) void k_convt2_bwd_in(const float* __restrict__ g, const float* __restrict__ w,
                                                       float* __restrict__ gin, int B, int Ci, int Co, int H, int W,
                                                       CfdDiv dHW, CfdDiv dW) {
    CFD_DYN_SHARED(float4, s_w4);  // [Co][CT_OB] x (4 taps)
    const int i0 = blockIdx.y * CT_OB, HW = H * W;
    for (int i = threadIdx.x; i < Co * CT_OB; i += blockDim.x) {
        const int o = i / CT_OB, ib = i - o * CT_OB;
        const float* wp = w + ((size_t)(i0 + ib) * Co + o) * 4;
        s_w4[i] = i0 + ib < Ci ? make_float4(wp[0], wp[1], wp[2], wp[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const unsigned total = (unsigned)B * HW;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned b = cfd_div(e, dHW);
        const int p = (int)(e - b * (unsigned)HW), y = (int)cfd_div((unsigned)p, dW), x = p - y * W;
        const float* gp = g + (size_t)b * Co * 4 * HW + (size_t)(2 * y) * (2 * W) + 2 * x;
        float acc[CT_OB];
#pragma unroll
        for (int ib = 0; ib < CT_OB; ++ib) acc[ib] = 0.f;
#pragma unroll 4
        for (int o = 0; o < Co; ++o) {
            const float2 g0 = *reinterpret_cast<const float2*>(gp + (size_t)o * 4 * HW);
            const float2 g1 = *reinterpret_cast<const float2*>(gp + (size_t)o * 4 * HW + 2 * W);
#pragma unroll
            for (int ib = 0; ib < CT_OB; ++ib) {
                const float4 wv = s_w4[o * CT_OB + ib];
                acc[ib] = fmaf(g0.x, wv.x, acc[ib]);
                acc[ib] = fmaf(g0.y, wv.y, acc[ib]);
                acc[ib] = fmaf(g1.x, wv.z, acc[ib]);
                acc[ib] = fmaf(g1.y, wv.w, acc[ib]);
            }
        }
#pragma unroll
        for (int ib = 0; ib < CT_OB; ++ib)
            if (i0 + ib < Ci) gin[((size_t)b * Ci + i0 + ib) * HW + p] = acc[ib];
    }
}

extern "C" int cfd_convt2_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int H,
                              int W, void* stream) {
    CFD_REQUIRE(in && w && out, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd: NULL pointer");
    CFD_REQUIRE(B >= 0 && Ci >= 1 && Co >= 1 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd: bad sizes");
    CFD_REQUIRE(((size_t)out & 7) == 0, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd: out must be 8-byte aligned (pixel pairs are stored as one vector)");
    if (B == 0) return CFD_OK;
    CFD_PROF_W("k_convt2_fwd", (hipStream_t)stream, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
    CFD_REQUIRE_I31((long)B * Co * 4 * H * W, "cfd_convt2_fwd");
    if (cfd_tune_get(CFD_TUNE_CONVT_MFMA) != 0 && cfd_convt6_covers(B, Ci, Co, H, W))
        return cfd_convt6_fwd(in, w, bias, out, B, Ci, Co, H, W, (hipStream_t)stream, "cfd_convt2_fwd");
    {
        const long px = (long)B * H * W;
        long bx = (px + 255) / 256;
        if (bx > 2048) bx = 2048;
        hipLaunchKernelGGL(k_convt2_fwd, dim3((unsigned)bx, (Co + CT_OB - 1) / CT_OB), dim3(256), (size_t)Ci * CT_OB * sizeof(float4),
                           (hipStream_t)stream, in, w, bias, out, B, Ci, Co, H, W, cfd_div_make((unsigned)(H * W)),
                           cfd_div_make((unsigned)W));
    }
    CFD_LAUNCH_CHECK("cfd_convt2_fwd");
    return CFD_OK;
}

extern "C" size_t cfd_convt2_bwd_workspace_bytes(int B, int Ci, int Co, int H, int W) {
    if (B <= 0) return 0;
    long chunk_px;
    int nchunk;
    wgrad_plan((long)B * H * W, Ci, Co * 4, chunk_px, nchunk);
    size_t a = cfd_align_up((size_t)nchunk * Ci * Co * 4 * sizeof(float), 256);
    const size_t b = chan_sum_ws_bytes(Co), c = cfd_convt6_wgrad_ws_bytes(B, Ci, Co, H, W);
    if (c > a) a = c;
    return a > b ? a : b;
}

extern "C" int cfd_convt2_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                              int B, int Ci, int Co, int H, int W, void* stream) {
    CFD_REQUIRE(gout && in && w && ws, CFD_ERR_INVALID_ARG, "cfd_convt2_bwd: NULL pointer");
    CFD_REQUIRE(B >= 1 && Ci >= 1 && Co >= 1 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_convt2_bwd: bad sizes");
    CFD_REQUIRE(((size_t)gout & 7) == 0, CFD_ERR_INVALID_ARG, "cfd_convt2_bwd: gout must be 8-byte aligned (pixel pairs are loaded as one vector)");
    CFD_REQUIRE_I31((long)B * (Ci > 4 * Co ? Ci : 4 * Co) * H * W, "cfd_convt2_bwd");
    hipStream_t st = (hipStream_t)stream;
    if (gin) {
        CFD_PROF_W("k_convt2_bwd_in", st, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
        const long px = (long)B * H * W;
        long bx = (px + 255) / 256;
        if (bx > 2048) bx = 2048;
        if (cfd_tune_get(CFD_TUNE_CONVT_MFMA) != 0 && cfd_convt6_covers(B, Ci, Co, H, W) && !((size_t)w & 15))  // (weight rows as float4)
            CFD_TRY(cfd_convt6_bwd_in(gout, w, gin, B, Ci, Co, H, W, st, "cfd_convt2_bwd(input)"));
        else
        hipLaunchKernelGGL(k_convt2_bwd_in, dim3((unsigned)bx, (Ci + CT_OB - 1) / CT_OB), dim3(256), (size_t)Co * CT_OB * sizeof(float4),
                           st, gout, w, gin, B, Ci, Co, H, W, cfd_div_make((unsigned)(H * W)), cfd_div_make((unsigned)W));
        CFD_LAUNCH_CHECK("cfd_convt2_bwd(input)");
    }
    if (gw && cfd_tune_get(CFD_TUNE_CONVT_MFMA) != 0 && cfd_convt6_wgrad_covers(B, Ci, Co, H, W) && !((size_t)in & 15) &&
        !((size_t)gout & 15)) {
        CFD_PROF_W("k_convt2_wgrad", st, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
        CFD_TRY(cfd_convt6_wgrad(gout, in, gw, gb, ws, B, Ci, Co, H, W, st, "cfd_convt2_bwd(wgrad)"));
        gw = nullptr, gb = nullptr;  // (the bias gradient rode in the same launches)
    }
    if (gw) {
        long chunk_px;
        int nchunk;
        const int J = Co * 4;
        wgrad_plan((long)B * H * W, Ci, J, chunk_px, nchunk);
        const ConvGeom g{B, Ci, Co, H, W, 2};
        const dim3 grid(nchunk, (Ci + 16 * CW_MT - 1) / (16 * CW_MT), (J + 16 * CW_NT - 1) / (16 * CW_NT));
        {
            CFD_PROF_W("k_convt2_wgrad", st, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
            hipLaunchKernelGGL((k_conv_wgrad<true>), grid, dim3(64 * CV_WAVES), 0, st, in, gout, (float*)ws, g,
                               (unsigned)chunk_px, cfd_div_make((unsigned)(H * W)), cfd_div_make((unsigned)W));
        }
        CFD_LAUNCH_CHECK("cfd_convt2_bwd(wgrad)");
        const long n = (long)Ci * Co * 4;
        cfd_conv_part_reduce((const float*)ws, gw, n, nchunk, st);
        CFD_LAUNCH_CHECK("cfd_convt2_bwd(reduce)");
    }
    if (gb) CFD_TRY(chan_sum(gout, gb, ws, B, Co, 4 * H * W, st, "cfd_convt2_bwd(bias)"));
    return CFD_OK;
}

// The same two calls with the 2H x 2W tensor addressed as a channel slice of a wider one: `out` / `gout` point at the slice's first
// channel of image 0, *_bstride is the number of elements between consecutive images (>= Co 4 H W).  The U-Net's up path writes the
// transposed convolution straight into the second half of its skip concatenation and reads the gradient from there (unet.py:80-88:
// torch.cat([x2, x1])), which saves a pass over x1 in each direction.  Matrix-pipe kernels only (convt6.hip): CFD_ERR_UNSUPPORTED
// where they do not apply -- the caller then uses the dense calls and a concatenation of its own.
extern "C" int cfd_convt2_fwd_ex(const float* in, const float* w, const float* bias, float* out, long out_bstride, int B, int Ci, int Co,
                                 int H, int W, void* stream) {
    if (out_bstride == 0 || out_bstride == (long)Co * 4 * H * W) return cfd_convt2_fwd(in, w, bias, out, B, Ci, Co, H, W, stream);
    CFD_REQUIRE(in && w && out, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd_ex: NULL pointer");
    CFD_REQUIRE(B >= 1 && Ci >= 1 && Co >= 1 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd_ex: bad sizes");
    CFD_REQUIRE(((size_t)out & 7) == 0 && out_bstride % 2 == 0, CFD_ERR_INVALID_ARG, "cfd_convt2_fwd_ex: out must be 8-byte aligned");
    CFD_REQUIRE(cfd_tune_get(CFD_TUNE_CONVT_MFMA) != 0 && cfd_convt6_covers(B, Ci, Co, H, W, out_bstride), CFD_ERR_UNSUPPORTED,
                "cfd_convt2_fwd_ex: a strided output needs the matrix-pipe kernel, which does not take this shape");
    CFD_PROF_W("k_convt2_fwd", (hipStream_t)stream, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
    return cfd_convt6_fwd(in, w, bias, out, B, Ci, Co, H, W, (hipStream_t)stream, "cfd_convt2_fwd_ex", out_bstride);
}

extern "C" int cfd_convt2_bwd_ex(const float* gout, long gout_bstride, const float* in, const float* w, float* gin, float* gw, float* gb,
                                 void* ws, int B, int Ci, int Co, int H, int W, void* stream) {
    if (gout_bstride == 0 || gout_bstride == (long)Co * 4 * H * W) return cfd_convt2_bwd(gout, in, w, gin, gw, gb, ws, B, Ci, Co, H, W, stream);
    CFD_REQUIRE(gout && in && w && ws, CFD_ERR_INVALID_ARG, "cfd_convt2_bwd_ex: NULL pointer");
    CFD_REQUIRE(B >= 1 && Ci >= 1 && Co >= 1 && H >= 1 && W >= 1, CFD_ERR_INVALID_ARG, "cfd_convt2_bwd_ex: bad sizes");
    CFD_REQUIRE(!gb || gw, CFD_ERR_UNSUPPORTED, "cfd_convt2_bwd_ex: the bias gradient rides in the weight-gradient launches (ask for both)");
    const bool ok = cfd_tune_get(CFD_TUNE_CONVT_MFMA) != 0 && cfd_convt6_covers(B, Ci, Co, H, W, gout_bstride) && !((size_t)gout & 15) &&
                    gout_bstride % 4 == 0 && (!gin || !((size_t)w & 15)) && (!gw || (cfd_convt6_wgrad_covers(B, Ci, Co, H, W) && !((size_t)in & 15)));
    CFD_REQUIRE(ok, CFD_ERR_UNSUPPORTED, "cfd_convt2_bwd_ex: a strided gradient needs the matrix-pipe kernels, which do not take this shape");
    hipStream_t st = (hipStream_t)stream;
    if (gin) {
        CFD_PROF_W("k_convt2_bwd_in", st, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
        CFD_TRY(cfd_convt6_bwd_in(gout, w, gin, B, Ci, Co, H, W, st, "cfd_convt2_bwd_ex(input)", gout_bstride));
    }
    if (gw) {
        CFD_PROF_W("k_convt2_wgrad", st, 4.0 * B * H * W * ((double)Ci + 4.0 * Co), 8.0 * B * H * W * (double)Ci * Co);
        CFD_TRY(cfd_convt6_wgrad(gout, in, gw, gb, ws, B, Ci, Co, H, W, st, "cfd_convt2_bwd_ex(wgrad)", gout_bstride));
    }
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// output epilogue of the conv baselines: preds = (x + residual) * mask  (src/models/unet.py:206-208)
// ------------------------------------------------------------------------------------------------------
// x, out: (B,C,HW); resid: rows of (B,Cr,HW) with Cr >= C (the first C channels are used); mask: (B,HW) or NULL
__global__ __launch_bounds__(256) void k_resid_mask(const float* __restrict__ x, const float* __restrict__ resid,
                                                    const float* __restrict__ mask, float* __restrict__ out, unsigned total,
                                                    int C, int Cr, int HW, CfdDiv dHW, CfdDiv dC) {
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned row = cfd_div(e, dHW), b = cfd_div(row, dC);
        const int p = (int)(e - row * (unsigned)HW), c = (int)(row - b * (unsigned)C);
        float v = x[e];
        if (resid) v += resid[((size_t)b * Cr + c) * HW + p];
        if (mask) v *= mask[(size_t)b * HW + p];
        out[e] = v;
    }
}

extern "C" int cfd_residual_mask(const float* x, const float* resid, const float* mask, float* out, int B, int C, int Cr,
                                 int HW, void* stream) {
    CFD_REQUIRE(x && out && B >= 0 && C >= 1 && HW >= 1 && (!resid || Cr >= C), CFD_ERR_INVALID_ARG, "cfd_residual_mask: bad arguments");
    if (B == 0) return CFD_OK;
    const long total = (long)B * C * HW;
    CFD_REQUIRE_I31(total, "cfd_residual_mask");
    hipLaunchKernelGGL(k_resid_mask, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, resid, mask, out,
                       (unsigned)total, C, Cr, HW, cfd_div_make((unsigned)HW), cfd_div_make((unsigned)C));
    CFD_LAUNCH_CHECK("cfd_residual_mask");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// nn.Dropout (src/models/resnet.py:45,76): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from a counter-based hash of
// (seed, element index).  The reference draws from torch's Philox stream, which cannot be reproduced; parity of the
// ResNet is therefore tested in eval mode (dropout = identity), and the training path is tested for its statistics
// and for forward/backward consistency (the backward pass regenerates the same mask from the same seed).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dropout(const float* __restrict__ x, float* __restrict__ y, size_t n, float p,
                                                 unsigned long long seed) {
    const float scale = 1.0f / (1.0f - p);
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = cfd_hash32(seed * 0x9E3779B97F4A7C15ULL + i) >= thresh ? x[i] * scale : 0.f;
}

// forward and backward are the same map (applied to x resp. the upstream gradient) for the same (p, seed)
extern "C" int cfd_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, void* stream) {
    CFD_REQUIRE(x && y, CFD_ERR_INVALID_ARG, "cfd_dropout: NULL pointer");
    CFD_REQUIRE(p >= 0.f && p < 1.f, CFD_ERR_INVALID_ARG, "cfd_dropout: p must be in [0, 1)");
    if (n == 0) return CFD_OK;
    hipLaunchKernelGGL(k_dropout, dim3(ew_blocks((long)n)), dim3(256), 0, (hipStream_t)stream, x, y, n, p, seed);
    CFD_LAUNCH_CHECK("cfd_dropout");
    return CFD_OK;
}
