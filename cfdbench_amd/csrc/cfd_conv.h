// Shared between conv.hip (exact-fp32 MFMA kernels, BatchNorm, pooling, the C ABI of the convolution stack) and conv6.hip (the
// three-piece split-bf16 implicit-GEMM kernels for k = 3, 5 and 7).
#pragma once
#include "cfd_common.h"

struct ConvGeom {
    int B, Ci, Co, H, W, ks;  // ks = kernel size (odd), pad = ks/2
    int zpad;                 // 1: ZERO padding (nn.Conv2d's default, src/models/auto_deeponet_cnn.py:17-33) on the conv6 kernels; 0: replicate
};

struct ConvTile {
    int TW, TH, NB;       // tile shape, NB*TH*TW <= 256
    int tiles_x, tiles_y; // tiles per image
    int LW, LH;           // LDS halo tile: (TH + ks - 1) x (TW + ks - 1) used, row stride LW
    CfdDiv dUsed, dLH;    // magic-number dividers by the used halo width (TW + ks - 1) and by LH (staging index split)
};

// Destination tile shape of the LDS-tiled forward / input-gradient kernels for a Hd x Wd grid: the power-of-two shape, or the
// shape with TW*TH <= 256 that needs the fewest workgroups per image (conv.hip).
void cfd_conv_tile_shape(int Hd, int Wd, int B, int& TW, int& TH, int& NB);

// out[e] = bias[channel(e)] + sum_z part[z][e] over split-K partial outputs (conv.hip)
int cfd_conv_splitk_sum(const float* part, const float* bias, float* out, long n, int nz, int Cm, long HWd, hipStream_t st,
                        const char* what);
// out[e] = sum_chunk part[chunk][e] in a fixed order (conv.hip)
void cfd_conv_part_reduce(const float* part, float* out, long n, int nchunk, hipStream_t st, float* out2 = nullptr, long n1 = 0);
// The same reduction as a job that rides in the first `nblk` workgroups of a LATER launch of the same call (k_fold_border carries the
// weight-gradient reduction of cfd_conv2d_bwd: one ~5-us launch less per convolution and backward pass; same code, same order).
struct CfdPartReduceJob {
    const float* part;  // NULL: no job
    float* out;
    float* out2;
    long n, n1;
    int nchunk, nblk, G;
};
CfdPartReduceJob cfd_conv_part_reduce_job(const float* part, float* out, long n, int nchunk, float* out2, long n1);

// ---- conv6.hip: k = 3 / k = 7 on three-piece split-bf16 operands (fp32-exact pieces, six bf16 MFMAs per product) ----
// forward (ext = false: dst (B,Co,H,W) = conv(src (B,Ci,H,W)) + bias) and the transposed-valid pass of the input gradient
// (ext = true: dst (B,Ci,H+2p,W+2p) from src = gout (B,Co,H,W)).  ws: cfd_conv6_ws_bytes(); returns CFD_ERR_UNSUPPORTED when the
// layer is not covered (the caller then uses the fp32 kernels).
bool cfd_conv6_covers(const ConvGeom& g, bool ext);
bool cfd_conv6_zeropad_covers(const ConvGeom& g);
size_t cfd_conv6_ws_bytes(const ConvGeom& g, bool ext);
// ext with `gin` != NULL: extended positions that map one-to-one onto an interior pixel may be written straight to gin (B,Ci,H,W)
// instead of dst; *direct then says so and the caller folds only the border pixels (k_fold_border) instead of every pixel.
// forward with `stats` != NULL ((Co, cfd_conv6_stats_slots(), 2) floats): per-channel partial sums of (out - bias) and its square.
// `wfrag` != NULL: the weights' fragments made earlier by cfd_conv6_wprep_batch (cfd_conv6_wfrag_bytes() bytes for this
// (Ci, Co, ks, ext)); the call then skips its own preparation launch.
int cfd_conv6_run(const float* src, const float* w, const float* bias, float* dst, void* ws, const ConvGeom& g, bool ext, float* gin,
                  bool* direct, hipStream_t st, const char* what, float* stats = nullptr, const void* wfrag = nullptr);
size_t cfd_conv6_wfrag_bytes(int Ci, int Co, int ks, bool ext);  // 0: no fragment form (kernel size other than 3 / 7)
int cfd_conv6_wprep_batch(int n, const float* const* w, void* const* wfrag, const int* Ci, const int* Co, const int* ks, const int* ext,
                          hipStream_t st, const char* what);
int cfd_conv6_stats_slots(const ConvGeom& g);
// weight gradient gw (Co,Ci,ks,ks) = sum over (b, p) of gout[b][o][p] * in[b][i][clamp(p + tap)]
bool cfd_conv6_wgrad_covers(const ConvGeom& g);
size_t cfd_conv6_wgrad_ws_bytes(const ConvGeom& g);
// gb != NULL: the bias gradient gb (Co) = sum over (b, p) of gout rides in the same launches (the gradient tile is staged anyway)
// defer != NULL: the reduction of the partial slices is not launched but described in *defer for a later launch to carry
int cfd_conv6_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, const ConvGeom& g, hipStream_t st,
                    const char* what, CfdPartReduceJob* defer = nullptr);

// ---- convt6.hip: ConvTranspose2d(2, stride 2) on three-piece split-bf16 operands (fp32-exact pieces, six bf16 MFMAs per product) ----
// *_bstride: elements between consecutive images of the 2H x 2W tensor (0 = dense): the tensor as a channel slice of a wider one
bool cfd_convt6_covers(int B, int Ci, int Co, int H, int W, long fine_bstride = 0);
int cfd_convt6_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int H, int W, hipStream_t st,
                   const char* what, long out_bstride = 0);
int cfd_convt6_bwd_in(const float* gout, const float* w, float* gin, int B, int Ci, int Co, int H, int W, hipStream_t st,
                      const char* what, long gout_bstride = 0);
// weight gradient gw (Ci,Co,2,2) and, when gb != NULL, the bias gradient gb (Co) in the same launches; needs W % 4 == 0, H W % 8 == 0 and 16-byte
// aligned tensors (CFD_ERR_UNSUPPORTED otherwise: the caller keeps the fp32 kernel).  ws: cfd_convt6_wgrad_ws_bytes().
bool cfd_convt6_wgrad_covers(int B, int Ci, int Co, int H, int W);
size_t cfd_convt6_wgrad_ws_bytes(int B, int Ci, int Co, int H, int W);
int cfd_convt6_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int H, int W,
                     hipStream_t st, const char* what, long gout_bstride = 0);

// ---- conv1.hip: 1x1 convolutions as streamed three-piece bf16 GEMMs over the pixels (off unless the conv1_mfma knob is 1) ----
bool cfd_conv1_covers(int B, int Ci, int Co, int HW);
int cfd_conv1_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW, hipStream_t st,
                  const char* what);
int cfd_conv1_dgrad(const float* gout, const float* w, float* gin, int B, int Ci, int Co, int HW, hipStream_t st, const char* what);
// needs H W % 8 == 0 and 16-byte aligned tensors (CFD_ERR_UNSUPPORTED otherwise); gb rides in the same launches
bool cfd_conv1_wgrad_covers(int B, int Ci, int Co, int HW);
size_t cfd_conv1_wgrad_ws_bytes(int B, int Ci, int Co, int HW);
int cfd_conv1_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW, hipStream_t st,
                    const char* what);
