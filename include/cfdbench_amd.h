/* cfdbench_amd -- C ABI of the MI355X-native (gfx950) hot path behind CFDBench's AutoCfdModel plugin surface.
 *
 * The reference (luo-yining/CFDBench) is pure Python/PyTorch and has NO native boundary of its own
 * (SURVEY.md section 8b); its extension surface is the Python classes in src/models/base_model.py:41-81.  The entry
 * points below are what a `torch.autograd.Function` under those classes binds (INTEGRATION.md shows the ctypes
 * stub).  Each function names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C, raw DEVICE pointers, explicit sizes, the HIP stream to enqueue on (void* = hipStream_t);
 *   - every function only ENQUEUES work on `stream` and returns an int status: 0 = ok, <0 = error
 *     (cfd_last_error() gives the text); no C++ exception crosses the boundary;
 *   - all buffers are owned by the caller (PyTorch's caching allocator); any scratch is a caller-provided
 *     workspace whose size the matching *_workspace_bytes() function reports;
 *   - fp32 tensors are NCHW-contiguous; complex64 tensors are interleaved (re,im) float pairs, exactly
 *     torch.view_as_real of the reference's parameters (state_dict ABI);
 *   - functions are re-entrant; a cfd_plan is immutable after creation and may be shared between threads;
 *   - numerics: fp32 storage and accumulation everywhere.  On 64-wide grids the DFT / inverse-DFT contractions, the
 *     1x1-conv weight gradients and the GEMMs of the projection head run as 3-term split-bf16 products on the bf16
 *     matrix cores (relative error <= ~2^-16 per product, measured nMSE <= 5e-11 vs an fp64 oracle; the benchmark's
 *     parity budget is 1e-5); everything else is exact fp32 FMA / MFMA arithmetic.
 */
#ifndef CFDBENCH_AMD_H
#define CFDBENCH_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFD_OK 0
#define CFD_ERR_INVALID_ARG (-1)
#define CFD_ERR_UNSUPPORTED (-2)
#define CFD_ERR_HIP (-3)
#define CFD_ERR_WORKSPACE (-4)

#define CFD_MAX_LAYERS 16

/* ABI version of this header; cfd_version() returns the library's.  Bumped whenever a buffer layout behind an unchanged signature
 * changes (500, round 5: the statistics records of cfd_conv2d_fwd_stats / cfd_batchnorm_fwd_stats are (C, slots, 4) floats since
 * round 4 -- a caller that still allocates (C, slots, 2) must fail at load time, not write out of bounds; the default of the
 * "act_pieces" knob is 3).  The Python binding refuses a library whose version differs (cfdbench_amd/_capi.py). */
#define CFD_ABI_VERSION 601
int cfd_version(void);
const char* cfd_last_error(void);

/* Dispatch overrides for tests and timing tools: which kernel variant a launch helper picks.  Knobs: "mix_nwv" (waves per
 * LDS-weight mixing workgroup, 0 = the lane = mode kernel), "wgrad_wg" (workgroups the tiled spectral weight gradient
 * aims at), "fused_variant" (0 = adjoint mix and spectral weight gradient as two launches), "block_fuse" (0 = the 1x1
 * weight gradient of a FnoBlock as its own kernel), "general_b3" (0 = grids other than 64 x 64 on the exact-fp32 generic
 * transform kernels instead of the split-bf16 ones), "head_blocks" (workgroup cap of the projection-head kernels).  value -1 restores the built-in choice.  The environment variables
 * CFD_MIX_NWV / CFD_WGRAD_WG / CFD_FUSED_VARIANT / CFD_BLOCK_FUSE are read once per process, at the first launch.  Every
 * route computes the same function (the reference has no such switch: it has one ATen call per op).                */
int cfd_tune_set(const char* name, int value);

/* Per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).  cfd_prof_end synchronises and
 * writes "kernel_name launches total_ms algorithmic_bytes flops\n" lines into buf (bytes / flops summed over the launches; 0 where a
 * launch site declares none).  Do not enable during stream capture.               */
int cfd_prof_begin(void);
int cfd_prof_end(char* buf, size_t cap);

/* ---- plan: pruned-DFT operator tables for one grid (H,W) and mode count (m1,m2) -------------------------
 * Replaces the implicit FFT plans behind torch.fft.rfft2 / irfft2 (src/models/fno/fno2d.py:62,81) and the
 * per-call host-side np.linspace coordinate grids of Fno2d.get_coords (fno2d.py:244-255).
 * Allocates a few tens of KB of device memory; create once per (H,W,m1,m2), never inside stream capture. */
typedef struct cfd_plan cfd_plan;
int cfd_plan_create(int H, int W, int m1, int m2, cfd_plan** out);
void cfd_plan_destroy(cfd_plan* plan);

/* ---- SpectralConv2d_fast pieces (src/models/fno/fno2d.py:59-82) ----------------------------------------
 * Kept modes: rows K = [0,m1) U [H-m1,H) (2*m1 rows, in that order) x columns [0,m2).  M = 2*m1*m2.        */

/* xh[img, row, l] = sum_{x,y} f(x[img,x,y]) e^{-2 pi i (K[row] x/H + l y/W)}   (== rfft2(x)[..., K, :m2], fno2d.py:62)
 * x: (nimg,H,W) f32; xh: (nimg,2*m1,m2) c64.  act_in: 0 = identity, 1 = exact-erf GELU applied on load.   */
int cfd_spectral_dft(const cfd_plan* plan, const float* x, float* xh, int nimg, int act_in, void* stream);

/* conj_t = 0:  z[b,o,row,l] = sum_i xh[b,i,row,l] * Wsel[i,o,row,l]           (compl_mul2d, fno2d.py:54-57,73-78)
 * conj_t = 1:  z[b,i,row,l] = sum_o conj(Wsel[i,o,row,l]) * xh[b,o,row,l]     (its adjoint w.r.t. the input)
 * Wsel = w1 for rows < m1, w2 for rows >= m1; w1,w2: (Cin,Cout,m1,m2) c64.                                  */
int cfd_spectral_mix(const cfd_plan* plan, const float* xh, const float* w1, const float* w2, float* z,
                     int B, int Cin, int Cout, int conj_t, void* stream);

/* out[img,x,y] = epi( (1/HW) Re sum_{row,l} c_l z[img,row,l] e^{+2 pi i (K[row] x/H + l y/W)} )
 * (== irfft2 of the zero-filled spectrum, fno2d.py:65-81; c_0 = 1, c_l = 2).
 * epi 0: value;  epi 1: value + addend;  epi 2: (value + addend) * gelu'(aprev).  addend may alias out.   */
int cfd_spectral_idft(const cfd_plan* plan, const float* z, const float* addend, const float* aprev, float* out,
                      int nimg, int epi, void* stream);

/* gw{1,2}[i,o,rm,l] = sum_b conj(xh[b,i,row,l]) * (c_l/HW) * gh[b,o,row,l]   (autograd of fno2d.py:73-78)
 * ws: cfd_spectral_wgrad_workspace_bytes() bytes.  Overwrites gw1, gw2.                                    */
size_t cfd_spectral_wgrad_workspace_bytes(const cfd_plan* plan, int B, int Cin, int Cout);
int cfd_spectral_wgrad(const cfd_plan* plan, const float* xh, const float* gh, float* gw1, float* gw2, void* ws,
                       int B, int Cin, int Cout, void* stream);

/* Both mode-domain consumers of the gradient modes gh in one call (autograd of compl_mul2d, fno2d.py:54-57,73-78):
 * gz = cfd_spectral_mix(gh, conj_t = 1) and gw{1,2} = cfd_spectral_wgrad(xh, gh).  One kernel launch for the pair
 * where the fused kernel applies (Cin == Cout == 20 or 32), the two calls above otherwise; same results either way up to
 * fp32 summation order.  ws: cfd_spectral_wgrad_workspace_bytes() bytes.                                     */
int cfd_spectral_mix_adj_wgrad(const cfd_plan* plan, const float* xh, const float* gh, const float* w1, const float* w2,
                               float* gz, float* gw1, float* gw2, void* ws, int B, int Cin, int Cout, void* stream);

/* y = SpectralConv2d_fast(x) (fno2d.py:59-82).  xh_out (B,Cin,2*m1,m2) c64 receives the kept input modes (saved
 * for the backward pass); z_ws (B,Cout,2*m1,m2) c64 is scratch.                                              */
int cfd_spectral_conv2d_fwd(const cfd_plan* plan, const float* x, const float* w1, const float* w2, float* y,
                            float* xh_out, float* z_ws, int B, int Cin, int Cout, void* stream);

/* Backward of the above given gy: gx (B,Cin,H,W), gw1, gw2 (Cin,Cout,m1,m2) c64.  xh = modes saved by forward.
 * ws: cfd_spectral_conv2d_bwd_workspace_bytes() bytes.  Any of gx / (gw1,gw2) may be NULL to skip it.       */
size_t cfd_spectral_conv2d_bwd_workspace_bytes(const cfd_plan* plan, int B, int Cin, int Cout);
int cfd_spectral_conv2d_bwd(const cfd_plan* plan, const float* gy, const float* xh, const float* w1, const float* w2,
                            float* gx, float* gw1, float* gw2, void* ws, int B, int Cin, int Cout, void* stream);

/* Fused FnoBlock body (fno2d.py:106-112): out[b,o] = b0[o] + sum_i w0[o,i] f(a[b,i]) + irfft2-of-kept-modes(z[b,o]),
 * i.e. cfd_chanmix followed by cfd_spectral_idft(epi=1) in ONE pass over the activation (the block's GELU is applied
 * by the consumers on load).  a: (B,Cin,H,W); z: (B,Cout,2*m1,m2) c64; w0: (Cout,Cin); out: (B,Cout,H,W).
 * Grids other than W == 64, H % 16 == 0 run the two-pass form internally.                                    */
int cfd_fno_block_fwd(const cfd_plan* plan, const float* a, const float* z, const float* w0, const float* b0,
                      float* out, int B, int Cin, int Cout, int act_in, void* stream);
/* Its input gradient: gin[b,i] = (sum_o w0[o,i] g[b,o] + irfft2-of-kept-modes(gz[b,i])) * gelu'(aprev[b,i]);
 * aprev == NULL skips the gelu' factor (first block, whose input is not an activation output).               */
int cfd_fno_block_bwd_input(const cfd_plan* plan, const float* g, const float* gz, const float* w0,
                            const float* aprev, float* gin, int B, int Cin, int Cout, void* stream);

/* ---- pointwise / channel-mixing pieces -----------------------------------------------------------------*/

/* out[b,o,p] = bias[o] + sum_i w[o,i] f(in[b,i,p])   (nn.Conv2d(k=1): fno2d.py:104,150; transpose=1 uses w[i,o]
 * and is the input-gradient of the same conv).  in: (B,Ci,HW); out: (B,Co,HW); w: (Co,Ci) (or (Ci,Co) rows if
 * transpose); bias may be NULL.  act_in as in cfd_spectral_dft.  Ci,Co <= 32.                              */
int cfd_chanmix(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW,
                int act_in, int transpose, void* stream);

/* gw[o,i] = sum_{b,p} g[b,o,p] f(in[b,i,p]);  gb[o] = sum_{b,p} g[b,o,p]   (weight/bias grads of that conv).
 * ws: cfd_chan_wgrad_workspace_bytes().  Overwrites gw, gb.                                                */
size_t cfd_chan_wgrad_workspace_bytes(int B, int Ci, int Co, int HW);
int cfd_chan_wgrad(const float* g, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW,
                   int act_in, void* stream);

/* Fno2d input assembly + fc0 (fno2d.py:189-217): features [u, v, mask, grid_x, grid_y, props...] -> (B,C,H,W).
 * inputs (B,in_chan,H,W); mask (B,1,H,W) or NULL (= ones, fno2d.py:189-191); case_params (B,P); w (C, in_chan+3+P). */
int cfd_fno_stem_fwd(const cfd_plan* plan, const float* inputs, const float* mask, const float* case_params,
                     const float* w, const float* bias, float* out, int B, int in_chan, int P, int C, void* stream);
size_t cfd_fno_stem_bwd_workspace_bytes(const cfd_plan* plan, int B, int in_chan, int P, int C);
int cfd_fno_stem_bwd(const cfd_plan* plan, const float* g, const float* inputs, const float* mask,
                     const float* case_params, float* gw, float* gb, void* ws, int B, int in_chan, int P, int C,
                     void* stream);

/* Projection head + mask + MseLoss sums (fno2d.py:228-237, src/models/loss.py:22-37):
 *   preds[b,c,p] = mask[b,p] * (b2[c] + sum_j w2[c,j] gelu(b1[j] + sum_i w1[j,i] f(a[b,i,p])))
 *   if label: sums[0..3] = { sum (preds - label*mask)^2, sum |preds - label*mask|, sum (label*mask)^2, count }
 * a: (B,C,HW); w1: (Hd,C); w2: (Co,Hd); Hd multiple of 16, <= 128; Co <= 2.  label/sums/ws may be NULL together. */
size_t cfd_fno_head_workspace_bytes(int B, int C, int Hd, int Co, int HW);
int cfd_fno_head_fwd(const float* a, const float* mask, const float* label, const float* w1, const float* b1,
                     const float* w2, const float* b2, float* preds, float* sums, void* ws, int B, int C, int Hd,
                     int Co, int HW, int act_in, void* stream);

/* Backward of the head.  Upstream gradient on preds is  gpreds_ext[b,c,p] (may be NULL)
 *   + coef[0] * 2 (preds - label*mask) + coef[1] * sign(preds - label*mask),   coef = 2 device floats
 * (nmse: coef[0] = 1/sums[2]; mse: coef[0] = 1/count; mae: coef[1] = 1/count -- written by cfd_loss_coef).
 * Outputs: ga (B,C,HW) = d/da (already multiplied by f'(a) when act_in=1); gw1,gb1,gw2,gb2 overwritten.     */
int cfd_fno_head_bwd(const float* a, const float* mask, const float* label, const float* preds,
                     const float* gpreds_ext, const float* coef, const float* w1, const float* b1, const float* w2,
                     float* ga, float* gw1, float* gb1, float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co,
                     int HW, int act_in, void* stream);

/* Both directions of the head in ONE pass for a training step whose loss is mse / nmse / mae: preds, sums[0] = sum d^2,
 * sums[1] = sum |d| (sums[2], sums[3] and coef are INPUTS here: cfd_label_energy_coef), ga and the four parameter
 * gradients.  Same results as cfd_fno_head_fwd + cfd_loss_coef + cfd_fno_head_bwd up to summation order; the hidden
 * layer's GELU is evaluated once instead of twice (fno2d.py:228-237 + the head part of train_auto.py:255).       */
int cfd_fno_head_train(const float* a, const float* mask, const float* label, const float* coef, const float* w1,
                       const float* b1, const float* w2, const float* b2, float* preds, float* sums, float* ga, float* gw1,
                       float* gb1, float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co, int HW, int act_in,
                       void* stream);

/* MseLoss on arbitrary tensors (src/models/loss.py:22-37): sums = {sum (p-l)^2, sum |p-l|, sum l^2, n}.
 * ws: cfd_loss_workspace_bytes(n).                                                                         */
size_t cfd_loss_workspace_bytes(size_t n);
int cfd_masked_loss_sums(const float* preds, const float* labels, float* sums, void* ws, size_t n, void* stream);
/* Gradient of the three sums: gp = gs[0]*2(p-l) + gs[1]*sign(p-l);  gl = -gp + gs[2]*2l.  gs: 3+ device floats.
 * gp / gl may be NULL.                                                                                        */
int cfd_loss_sums_bwd(const float* preds, const float* labels, const float* gsums, float* gp, float* gl, size_t n,
                      void* stream);
/* scores[0..3] = {mse, rmse, mae, nmse} from sums (loss.py:27-35).                                           */
int cfd_loss_scores(const float* sums, float* scores, void* stream);
/* gsums[4] = d(scores)/d(sums) applied to the upstream gradients of mse / rmse / mae / nmse (device scalars; NULL = no gradient):
 * the backward pass of cfd_loss_scores in one launch, with autograd's fp32 operation order for loss.py:27-35.  gsums[3] = 0.   */
int cfd_loss_scores_bwd(const float* sums, const float* g_mse, const float* g_rmse, const float* g_mae, const float* g_nmse,
                        float* gsums, void* stream);
/* MseLoss.forward / its backward as one autograd node (src/models/loss.py:22-37): sums (4 floats, as cfd_masked_loss_sums) AND scores
 * (mse, rmse, mae, nmse) from two launches; the gradients on preds / labels (either may be NULL) from the upstream gradients of the four
 * scores (device scalars or NULL) in ONE launch.  Same fp32 operations as the four entry points above in the same order.              */
int cfd_mse_loss_fwd(const float* preds, const float* labels, float* sums, float* scores, void* ws, size_t n, void* stream);
int cfd_mse_loss_bwd(const float* preds, const float* labels, const float* sums, const float* g_mse, const float* g_rmse,
                     const float* g_mae, const float* g_nmse, float* gp, float* gl, size_t n, void* stream);
/* The same with STRIDED label rows (ABI 601): labels (rows, cols) with row stride ldl >= cols elements -- the channel slice label[:, 0] of a
 * (B, C, H, W) tensor viewed as (B, H W), which MseLoss receives from the Auto-DeepONet family (src/models/auto_deeponet.py:137-142) -- so
 * that no contiguous copy is made first.  preds and gp are contiguous (rows, cols); the label gradient is not produced.  rows cols < 2^31. */
int cfd_mse_loss_fwd_ld(const float* preds, const float* labels, float* sums, float* scores, void* ws, size_t rows, size_t cols, size_t ldl,
                        void* stream);
int cfd_mse_loss_bwd_ld(const float* preds, const float* labels, const float* sums, const float* g_mse, const float* g_rmse,
                        const float* g_mae, const float* g_nmse, float* gp, size_t rows, size_t cols, size_t ldl, void* stream);
/* out (rows, ka + kb) = [a | b] with a (rows, ka) at row stride lda and b (rows, kb) at row stride ldb (ABI 601): the branch input
 * [u.flatten(), case_params] of the Auto-DeepONet family (src/models/auto_deeponet.py:109-116) from the field's channel slice and the case
 * parameters in one launch.  rows (ka + kb) < 2^31.                                                                                   */
int cfd_rows_concat2(const float* a, size_t lda, size_t ka, const float* b, size_t ldb, size_t kb, float* out, size_t rows, void* stream);
/* coef for cfd_fno_head_bwd: which = 0 mse, 1 nmse, 2 mae; scaled by `upstream` (d objective / d loss).       */
int cfd_loss_coef(const float* sums, float* coef, int which, float upstream, void* stream);
/* The same coefficients BEFORE any prediction exists: d mse|nmse|mae / d preds need only the element count and
 * sum (label * mask)^2 (loss.py:27-35), which this computes from the labels (sums[2], sums[3], coef[0..1]; mask (B,HW) may
 * be NULL; ws: cfd_label_energy_workspace_bytes()).  Used by cfd_fno_forward_train.                              */
size_t cfd_label_energy_workspace_bytes(void);
int cfd_label_energy_coef(const float* label, const float* mask, float* sums, float* coef, void* ws, int B, int out_chan,
                          int HW, int which, float upstream, void* stream);

/* nn.GELU() (exact erf, fno2d.py:147) as a standalone pass -- only used by the stand-alone FnoBlock module;
 * inside Fno2d the activation is fused into the consumers.  bwd: gx = gy * gelu'(x).                          */
int cfd_gelu_fwd(const float* x, float* y, size_t n, void* stream);
int cfd_gelu_bwd(const float* x, const float* gy, float* gx, size_t n, void* stream);

/* torch.optim.Adam step on one flat fp32 buffer (train_auto.py:213,256; complex params as (re,im) pairs --
 * torch's view_as_real handling).  step >= 1.                                                               */
int cfd_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same step for n parameter tensors in one launch per 80 tensors (host tables of device pointers and element counts).  lr_dev /
 * step_dev: NULL, or device scalars that override lr / step (torch.optim.Adam(capturable=True) keeps both on the device, so that a
 * captured HIP graph replays with the current values); bias corrections in fp32.  fp32 real tensors only.                      */
int cfd_adam_multi(int n, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                   const size_t* numel, const float* lr_dev, float lr, const float* step_dev, float step, float beta1, float beta2,
                   float eps, float weight_decay, float grad_scale, void* stream);

/* dst[k][i] = scale * src[k][i] for n fp32 tensors in one launch per 80 (host tables of device pointers and element counts).  The
 * data-parallel gradient exchange of the autograd training paths (a-11; no reference symbol): every parameter gradient, pre-scaled by
 * 1 / world, into one flat all-reduce buffer.                                                                                      */
int cfd_scale_copy_multi(int n, const float* const* src, float* const* dst, const size_t* numel, float scale, void* stream);

/* ---- dense layers of the DeepONet family (fp32 MFMA GEMMs) ----------------------------------------------*/

/* c (M,N) = op(a) op(b); trans_a: a stored (K,M) else (M,K); trans_b: b stored (N,K) else (K,N); row-major, leading
 * dimensions in elements.                                                                                    */
size_t cfd_gemm_workspace_bytes(int M, int N, int K); /* split-K partial tiles of skinny products; 0 = none needed */
int cfd_gemm(const float* a, const float* b, float* c, void* ws, int M, int N, int K, int lda, int ldb, int ldc,
             int trans_a, int trans_b, void* stream);

/* nn.Linear + activation (Ffn, src/models/ffn.py:23-31): y = act(x w^T + bias); x (M,K), w (N,K), y (M,N).
 * act: 0 none, 1 relu, 2 tanh, 3 gelu (exact erf), 4 swish (get_act_fn, src/models/act_fn.py:5-18).
 * preact (M,N) receives x w^T + bias when non-NULL (required for gelu / swish, whose derivative needs it).
 * ws: cfd_linear_fwd_workspace_bytes() bytes (may be NULL when that is 0).                                    */
size_t cfd_linear_fwd_workspace_bytes(int M, int K, int N);
int cfd_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* preact, void* ws, int M, int K,
                   int N, int act, void* stream);
/* Backward of the above: gx (M,K), gw (N,K), gb (N) from gy (M,N); y = layer output (relu / tanh), preact (gelu /
 * swish).  gx and gb may be NULL.  ws: cfd_linear_bwd_workspace_bytes().                                    */
size_t cfd_linear_bwd_workspace_bytes(int M, int K, int N);
int cfd_linear_bwd(const float* gy, const float* x, const float* w, const float* y, const float* preact, float* gx,
                   float* gw, float* gb, void* ws, int M, int K, int N, int act, void* stream);
/* The same; with in_act != 0, x is the OUTPUT of a layer with that activation (in_preact: its pre-activations, gelu / swish only) and
 * gx leaves as that layer's dZ = (gz w) * in_act'(x): the caller runs that layer's backward with act = 0 and this dZ as its gy -- one
 * pass over the (M, K) gradient less per layer of a Linear chain (src/models/ffn.py:12-35). */
int cfd_linear_bwd_ex(const float* gy, const float* x, const float* w, const float* y, const float* preact, float* gx, float* gw,
                      float* gb, void* ws, int M, int K, int N, int act, int in_act, const float* in_preact, void* stream);

/* DeepONet output (src/models/auto_deeponet.py:127-135, src/models/deeponet.py:204-205):
 *   preds[b,k] = sum_p branch[b,p] trunk[k,p] + bias[0] + (u ? u[b*HW + (qidx ? qidx[k] : k)] : 0)
 * branch (B,P); trunk (Kq,P); u (B,HW) = the u channel of the input frame (residual) or NULL; qidx (Kq) int32 flat
 * lattice indices row*W+col of the query points, NULL = the full lattice in row-major order.                  */
int cfd_deeponet_inner_fwd(const float* branch, const float* trunk, const float* bias, const float* u, const int* qidx,
                           float* preds, int B, int P, int Kq, int HW, void* stream);
/* The same with a row stride for u (ldu >= HW floats): the residual field read in place from the leading columns of the branch
 * net's input matrix [u.flatten(), case_params] (auto_deeponet.py:111-116).                                          */
int cfd_deeponet_inner_fwd_ex(const float* branch, const float* trunk, const float* bias, const float* u, int ldu, const int* qidx,
                              float* preds, int B, int P, int Kq, int HW, void* stream);
/* gbranch (B,P) = g trunk; gtrunk (Kq,P) = g^T branch; gbias (1) = sum g.  Any output may be NULL.             */
size_t cfd_deeponet_inner_bwd_workspace_bytes(int B, int P, int Kq);
int cfd_deeponet_inner_bwd(const float* gpreds, const float* branch, const float* trunk, float* gbranch, float* gtrunk,
                           float* gbias, void* ws, int B, int P, int Kq, void* stream);

/* Stand-alone activation get_act_fn(name) (src/models/act_fn.py:8-18) on n contiguous floats; act 0 none, 1 relu, 2 tanh,
 * 3 gelu, 4 swish.  Backward: gx = gy * act'(x), reading y = act(x) for relu / tanh and x for gelu / swish (the other may
 * be NULL). */
/* A whole stack of L <= 16 Linear(+activation) layers whose widths are all <= 128 as ONE kernel per direction (Ffn,
 * src/models/ffn.py:12-35; Auto-DeepONet's branch / trunk nets, src/models/auto_deeponet.py:52-60): y_0 = x (R, dims[0]);
 * z_l = y_{l-1} w_l^T + b_l, y_l = act(z_l), no activation after the last layer unless act_last.  w / b / y / z: host arrays of L
 * device pointers (w_l: (dims[l+1], dims[l]) as nn.Linear.weight; b may be NULL or hold NULLs; y_l (R, dims[l+1]) is written by
 * the forward pass and read by the backward pass; z_l the same for the pre-activations, needed for act = 3 gelu / 4 swish only).
 * Backward: gy (R, dims[L]) -> gw_l, gb_l (overwritten) and, if gx != NULL, gx (R, dims[0]); ws: cfd_ffn_stack_bwd_workspace_bytes().
 * Exact fp32 (v_mfma_f32_16x16x4_f32).                                                                                   */
int cfd_ffn_stack_fwd(const float* x, const float* const* w, const float* const* b, float* const* y, float* const* z, int R,
                      const int* dims, int L, int act, int act_last, void* stream);
size_t cfd_ffn_stack_bwd_workspace_bytes(int R, const int* dims, int L);
int cfd_ffn_stack_bwd(const float* x, const float* gy, const float* const* w, float* const* y, float* const* z,
                      float* const* gw, float* const* gb, float* gx, void* ws, int R, const int* dims, int L, int act,
                      int act_last, void* stream);

/* Up to 3 independent stacks in ONE launch per direction (the branch and trunk nets of a DeepONet variant -- src/models/auto_deeponet.py:52-60,
 * auto_edeeponet.py:37-39 -- are 32 + 269 row tiles at BASELINE configs[3]: each too small to fill the GPU, each at its latency floor).
 * One element per stack, fields as the arguments of cfd_ffn_stack_fwd / _bwd (gy .. ws are read by cfd_ffn_stacks_bwd only; ws:
 * cfd_ffn_stack_bwd_workspace_bytes(R, dims, L) bytes per stack).  Same arithmetic and summation orders as the single-stack calls,
 * which are these with n = 1. */
typedef struct cfd_ffn_stack_args {
    const float* x;
    const float* const* w;
    const float* const* b;
    float* const* y;
    float* const* z;
    int R;
    const int* dims;
    int L, act, act_last;
    const float* gy;
    float* const* gw;
    float* const* gb;
    float* gx;
    void* ws;
} cfd_ffn_stack_args;
int cfd_ffn_stacks_fwd(int n, const cfd_ffn_stack_args* stacks, void* stream);
int cfd_ffn_stacks_bwd(int n, const cfd_ffn_stack_args* stacks, void* stream);

int cfd_act_fwd(const float* x, float* y, size_t n, int act, void* stream);
int cfd_act_bwd(const float* gy, const float* y, const float* x, float* gx, size_t n, int act, void* stream);

/* NormAct (src/models/act_fn.py:21-47): per sample (all dims but the first; x viewed as (S,L)): y = act((x-mean)/std) * std
 * + mean with the unbiased std and no epsilon.  stats (S,2) keeps (mean, std) for the backward pass.  act 1..4. */
int cfd_normact_fwd(const float* x, float* y, float* stats, int S, long L, int act, void* stream);
int cfd_normact_bwd(const float* x, const float* gy, const float* stats, float* gx, int S, long L, int act, void* stream);

/* Non-autoregressive DeepONet (src/models/deeponet.py:184-205):
 *   trunk input  out[b,k,p] = ft[b,p] + fxy[k,p];   backward: gft = sum_k g, gfxy = sum_b g (either may be NULL)
 *   output       preds[b,k] = sum_p branch[b,p] trunk[b,k,p] + bias[0]                                          */
int cfd_bcast_add_fwd(const float* ft, const float* fxy, float* out, int B, int K, int P, void* stream);
int cfd_bcast_add_bwd(const float* g, float* gft, float* gfxy, int B, int K, int P, void* stream);
int cfd_rowdot_fwd(const float* branch, const float* trunk, const float* bias, float* preds, int B, int K, int P,
                   void* stream);
size_t cfd_rowdot_bwd_workspace_bytes(void);
int cfd_rowdot_bwd(const float* g, const float* branch, const float* trunk, float* gbranch, float* gtrunk, float* gbias,
                   void* ws, int B, int K, int P, void* stream);

/* ---- convolution stack of the U-Net / ResNet baselines (src/models/unet.py, src/models/resnet.py) ---------*/

/* out (B,Co,H,W) = nn.Conv2d(Ci, Co, ks, padding=ks/2, padding_mode="replicate")(in); w (Co,Ci,ks,ks); ks odd <= 7
 * (unet.py:20-27 ks=3, resnet.py:35-41 ks=7, unet.py:105 ks=1); bias may be NULL.  ws: cfd_conv2d_fwd_workspace_bytes()
 * bytes (weight fragments and split-K partials of the k = 3 / 7 kernels; 0 for other kernel sizes), or NULL: k = 3 / 7 then run
 * on the slower exact-fp32 gather kernel.  Both routes are fp32-exact class (the k = 3 / 7 kernels multiply three-piece bf16
 * operands, six MFMAs per product).  The 2x2 transposed conv needs `out` / `gout` 8-byte aligned.                        */
size_t cfd_conv2d_fwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks);
int cfd_conv2d_fwd(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int Ci, int Co, int H,
                   int W, int ks, void* stream);
/* The same forward, also emitting per-channel records for the training-mode BatchNorm that follows the conv (unet.py:20-30:
 * Conv2d -> BatchNorm2d): stats (Co, slots, 4) floats = (m, m2, n, -) = mean of out - bias, sum of squared deviations from it and
 * pixel count of each slot (disjoint pixel sets; running-mean updates, so that nothing cancels when |mean| >> std), slots =
 * cfd_conv2d_fwd_stats_slots() (0: this layer cannot emit them).  Consumed by cfd_batchnorm_fwd_stats (one launch instead of the
 * statistics pass + the normalising pass).                                                                               */
int cfd_conv2d_fwd_stats_slots(int B, int Ci, int Co, int H, int W, int ks);
int cfd_conv2d_fwd_stats(const float* in, const float* w, const float* bias, float* out, void* ws, float* stats, int B, int Ci,
                         int Co, int H, int W, int ks, void* stream);
/* Weights prepared ahead of the calls that use them.  The k = 3 / 7 kernels read a layer's weights as MFMA fragments of three bf16
 * pieces; cfd_conv2d_fwd / cfd_conv2d_bwd make them with a small launch of their own in every call.  A model that runs many
 * layers per step (unet.py:153-223: 18 convolutions, resnet.py:145-198: 14) makes them for ALL layers in one launch instead:
 * wfrag[i] = cfd_conv2d_wfrag_bytes(Ci[i], Co[i], ks[i], transposed[i]) bytes (0: the layer has no fragment form -- kernel size
 * other than 3 / 7), transposed = 0 for the forward pass, 1 for the input-gradient pass; the form depends on neither batch nor
 * grid size.  The fragments are a pure function of the weights: remake them after every change of the weights (optimizer step,
 * load_state_dict).  cfd_conv2d_fwd_ex / cfd_conv2d_bwd_ex are the calls above with the optional extras spelled out -- stats (NULL
 * or as in cfd_conv2d_fwd_stats) and wfrag / wfrag_t (NULL or the prepared fragments; ignored on layers that run elsewhere).  */
size_t cfd_conv2d_wfrag_bytes(int Ci, int Co, int ks, int transposed);
int cfd_conv2d_wprep_batch(int n, const float* const* w, void* const* wfrag, const int* Ci, const int* Co, const int* ks,
                           const int* transposed, void* stream);
int cfd_conv2d_fwd_ex(const float* in, const float* w, const float* bias, float* out, void* ws, float* stats, const void* wfrag,
                      int B, int Ci, int Co, int H, int W, int ks, void* stream);
/* gin (B,Ci,H,W), gw (Co,Ci,ks,ks), gb (Co) from gout; any output may be NULL.  ws: cfd_conv2d_bwd_workspace_bytes(). */
size_t cfd_conv2d_bwd_workspace_bytes(int B, int Ci, int Co, int H, int W, int ks);
int cfd_conv2d_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws, int B,
                   int Ci, int Co, int H, int W, int ks, void* stream);
int cfd_conv2d_bwd_ex(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                      const void* wfrag_t, int B, int Ci, int Co, int H, int W, int ks, void* stream);
/* nn.Conv2d(k, padding = k / 2) with ZERO padding (the CNN branch of src/models/auto_deeponet_cnn.py:17-33), k = 3 / 5 / 7, for the
 * shapes cfd_conv2d_zeropad_supported() accepts (others: zero-pad the input by k / 2, run cfd_conv2d_fwd on the larger grid, crop).
 * ws: cfd_conv2d_fwd_workspace_bytes() / cfd_conv2d_bwd_workspace_bytes() of the same shape. */
int cfd_conv2d_zeropad_supported(int B, int Ci, int Co, int H, int W, int ks);
int cfd_conv2d_zeropad_fwd(const float* in, const float* w, const float* bias, float* out, void* ws, int B, int Ci, int Co, int H, int W,
                           int ks, void* stream);
int cfd_conv2d_zeropad_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws, int B,
                           int Ci, int Co, int H, int W, int ks, void* stream);

/* y = [relu](nn.BatchNorm2d(x)) (unet.py:28-30).  training: batch statistics, saved in save_mean / save_rstd for the
 * backward pass, running_mean / running_var updated in place (momentum; unbiased variance) when non-NULL;
 * otherwise the running statistics normalise.  x, y: (B,C,HW).  ws: cfd_batchnorm_workspace_bytes(C).           */
size_t cfd_batchnorm_workspace_bytes(int C);
int cfd_batchnorm_fwd(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* y,
                      float* save_mean, float* save_rstd, void* ws, int B, int C, int HW, float eps, float momentum,
                      int training, int relu, void* stream);
/* training-mode forward from the records of cfd_conv2d_fwd_stats: stats (C, slots, 4), shift (C) = the conv bias or NULL */
int cfd_batchnorm_fwd_stats(const float* x, const float* gamma, const float* beta, float* run_mean, float* run_var, float* y,
                            float* save_mean, float* save_rstd, const float* stats, int slots, const float* shift, int B, int C,
                            int HW, float eps, float momentum, int relu, void* stream);
int cfd_batchnorm_bwd(const float* gy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_rstd, float* gx, float* ggamma, float* gbeta, void* ws, int B, int C, int HW,
                      int training, int relu, void* stream);

/* nn.MaxPool2d(2) (unet.py:59) on nimg = B*C images; the gradient goes to the first maximum of each window.     */
int cfd_maxpool2_fwd(const float* x, float* y, int nimg, int H, int W, void* stream);
int cfd_maxpool2_bwd(const float* x, const float* gy, float* gx, int nimg, int H, int W, void* stream);
/* The same with a second gradient of x summed in: add (B, C, H, W) with a batch stride of add_batch_stride floats (the skip
 * connection's share of the decoder concatenation's gradient, unet.py:80-88, read in place). */
int cfd_maxpool2_bwd_add(const float* x, const float* gy, const float* add, size_t add_batch_stride, float* gx, int B, int C, int H,
                         int W, void* stream);

/* nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) (unet.py:74-76, UNet(bilinear=True)) on nimg = B*C
 * images: x (nimg,H,W) -> y (nimg,2H,2W); the backward is a gather (no atomics): gy (nimg,2H,2W) -> gx (nimg,H,W).  */
int cfd_upsample2_bilinear_fwd(const float* x, float* y, int nimg, int H, int W, void* stream);
int cfd_upsample2_bilinear_bwd(const float* gy, float* gx, int nimg, int H, int W, void* stream);

/* nn.ConvTranspose2d(Ci, Co, kernel_size=2, stride=2) (unet.py:80): in (B,Ci,H,W), w (Ci,Co,2,2), out (B,Co,2H,2W). */
int cfd_convt2_fwd(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int H, int W,
                   void* stream);
size_t cfd_convt2_bwd_workspace_bytes(int B, int Ci, int Co, int H, int W);
int cfd_convt2_bwd(const float* gout, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws, int B,
                   int Ci, int Co, int H, int W, void* stream);
/* The same with the 2H x 2W tensor addressed as a channel slice of a wider one (out / gout = the slice's first channel of image 0,
 * *_bstride = elements between consecutive images, 0 = dense): unet.py:80-88 concatenates the transposed convolution's output behind
 * the skip tensor -- written there directly and its gradient read from there, a pass over it is saved in each direction.  Strided
 * tensors run on the matrix-pipe kernels only: CFD_ERR_UNSUPPORTED where those do not apply (W % 4, H W % 8, alignment).        */
int cfd_convt2_fwd_ex(const float* in, const float* w, const float* bias, float* out, long out_bstride, int B, int Ci, int Co, int H,
                      int W, void* stream);
int cfd_convt2_bwd_ex(const float* gout, long gout_bstride, const float* in, const float* w, float* gin, float* gw, float* gb, void* ws,
                      int B, int Ci, int Co, int H, int W, void* stream);

/* out = (x + resid[:, :C]) * mask (unet.py:206-208): x, out (B,C,HW); resid (B,Cr,HW) or NULL; mask (B,HW) or NULL. */
int cfd_residual_mask(const float* x, const float* resid, const float* mask, float* out, int B, int C, int Cr, int HW,
                      void* stream);

/* nn.Dropout (resnet.py:45,76): y = x * keep / (1-p) with keep from a counter-based hash of (seed, index).  The same
 * call with the upstream gradient as x is the backward pass.  (torch's Philox stream is not reproducible here.)  */
int cfd_dropout(const float* x, float* y, size_t n, float p, unsigned long long seed, void* stream);
/* y = gelu(dropout(x)) and gx = d/dx of it applied to gy, one pass each (resnet.py:70-77: conv1 -> nn.Dropout -> nn.GELU); p = 0 is
 * the plain GELU.  Value for value cfd_dropout followed by cfd_gelu_fwd / cfd_gelu_bwd followed by cfd_dropout.  n % 4 == 0 and
 * 16-byte aligned tensors, else CFD_ERR_UNSUPPORTED.                                                                        */
int cfd_dropout_gelu_fwd(const float* x, float* y, size_t n, float p, unsigned long long seed, void* stream);
int cfd_dropout_gelu_bwd(const float* x, const float* gy, float* gx, size_t n, float p, unsigned long long seed, void* stream);
/* The same with the stream's step counter in DEVICE memory: seed = splitmix64(base + *step) & (2^48 - 1), computed by the kernel, so
 * that a train step replayed from a captured graph (which increments *step inside the graph) draws a new mask each time.  With the
 * host computing the same expression and passing it as `seed` above, the masks are identical. */
int cfd_dropout_gelu_fwd_step(const float* x, float* y, size_t n, float p, unsigned long long base, const unsigned long long* step,
                              void* stream);
int cfd_dropout_gelu_bwd_step(const float* x, const float* gy, float* gx, size_t n, float p, unsigned long long base,
                              const unsigned long long* step, void* stream);

/* ---- whole Auto-FNO (Fno2d.forward, fno2d.py:178-242; loss.backward() at train_auto.py:255) ---------------*/
typedef struct {
    int B, H, W;
    int in_chan, out_chan, n_case_params;
    int hidden, num_layers, modes1, modes2, head;
} cfd_fno_shape;

typedef struct { /* device pointers, reference state_dict order (SURVEY.md 8b "Checkpoint ABI") */
    float* fc0_w;
    float* fc0_b;
    float* spec_w1[CFD_MAX_LAYERS]; /* blocks.{l}.conv0.weights1, c64 */
    float* spec_w2[CFD_MAX_LAYERS];
    float* w0_w[CFD_MAX_LAYERS];    /* blocks.{l}.w0.weight */
    float* w0_b[CFD_MAX_LAYERS];
    float* fc1_w;
    float* fc1_b;
    float* fc2_w;
    float* fc2_b;
} cfd_fno_params;

/* Bytes of caller-provided workspace: activations kept for backward + scratch.  training=0: forward only.   */
size_t cfd_fno_workspace_bytes(const cfd_plan* plan, const cfd_fno_shape* shape, int training);

/* preds (B,out_chan,H,W); sums (4 floats) written iff label != NULL.  With training=1 the workspace keeps what
 * cfd_fno_backward needs (same ws pointer must be passed to it, untouched in between).                      */
int cfd_fno_forward(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                    const float* inputs, const float* case_params, const float* mask, const float* label,
                    float* preds, float* sums, void* ws, int training, void* stream);

/* The same forward pass with the storage type of the activations BETWEEN kernels as an argument (BASELINE.json configs[4]: the
 * 200-step rollout of src/test_multistep.py:135-177 with bf16 storage): act_dtype 0 = fp32 (== cfd_fno_forward), 1 = bf16
 * (the lifting layer's output and every FnoBlock's pre-activation are rounded to bf16 when stored; inputs, predictions, kept
 * modes, weights, arithmetic and accumulation stay fp32).  cfd_fno_forward_ex(bf16) is the inference path (training must be 0);
 * bf16-storage TRAINING goes through cfd_fno_forward_train_ex / cfd_fno_backward_phase_ex below.  Grids up to 70 x 80.       */
size_t cfd_fno_workspace_bytes_ex(const cfd_plan* plan, const cfd_fno_shape* shape, int training, int act_dtype);
int cfd_fno_forward_ex(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                       const float* inputs, const float* case_params, const float* mask, const float* label,
                       float* preds, float* sums, void* ws, int training, int act_dtype, void* stream);

/* Training forward for a loss fixed in advance (`which` = 0 mse, 1 nmse, 2 mae; FnoTrainEngine): cfd_fno_forward(training = 1)
 * with the projection head run ONCE for both directions -- preds, sums[0..3], coef, d loss / d a_L (workspace) and the
 * gradients of fc1 / fc2 (grads) all leave one kernel, so the hidden layer's GELU is evaluated once per step instead of
 * twice.  Replaces `model(**batch)` + the head part of `loss.backward()` (train_auto.py:233-255); continue with
 * cfd_fno_backward_phase(1 .. num_layers + 1).                                                                 */
int cfd_fno_forward_train(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                          const cfd_fno_params* grads, const float* inputs, const float* case_params, const float* mask,
                          const float* label, float* preds, float* sums, float* coef, void* ws, int which, float upstream,
                          void* stream);

/* The training step with the storage type of the SAVED activations as an argument (the mixed-precision option of this
 * fork's other trainers, src/args.py:77-80, src/train_gencast.py:324-340): act_dtype 0 = fp32 (== the functions without _ex),
 * 1 = bf16 -- a_0 .. a_L are rounded to bf16 when stored and the backward pass reads those rounded values (it differentiates the
 * computation that was run); parameters, kept modes, gradients, accumulation and the optimiser stay fp32.  Workspace:
 * cfd_fno_workspace_bytes_ex(plan, shape, 1, act_dtype).  With act_dtype = 1 backward phase 0 does not exist (the head ran in
 * cfd_fno_forward_train_ex); phases 1 .. num_layers + 1 as below.                                                           */
int cfd_fno_forward_train_ex(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                             const cfd_fno_params* grads, const float* inputs, const float* case_params, const float* mask,
                             const float* label, float* preds, float* sums, float* coef, void* ws, int which, float upstream,
                             int act_dtype, void* stream);
int cfd_fno_backward_phase_ex(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                              const cfd_fno_params* grads, const float* inputs, const float* case_params,
                              const float* mask, const float* label, const float* preds, const float* gpreds_ext,
                              const float* coef, void* ws, int phase, int act_dtype, void* stream);

/* Round 6 -- the fused SINGLE-GPU training step with deferred work (FnoTrainEngine.train_step; the reference's loop is
 * src/train_auto.py:231-257: model(**batch) -> loss["nmse"].backward() -> Adam.step()).  Three tiny launches of the step above --
 * the label-energy pair in front of the head, the head's partial-sum reduction behind it, the lifting layer's combine in front of
 * Adam -- each cost ~5 us of dispatch floor for microseconds of work; with `flags` they ride in launches that exist anyway:
 *   CFD_TRAIN_DEFER_SCALE  nmse only: the head runs with the mse coefficient upstream / n (passed by value, `coef` unused) and ALSO sums
 *                          (label * mask)^2; every gradient of the pass is then short of the factor n / sum (label*mask)^2, which
 *                          cfd_fno_adam_step applies from sums[2], sums[3] (every gradient is linear in that scalar).  Until then the
 *                          gradient buffers hold the gradients of sum d^2 * upstream / n.
 *   CFD_TRAIN_DEFER_HEAD   the head's reduction rides in backward phase 1's FnoBlock kernel (fc1 / fc2 gradients and sums[] are final
 *                          after phase 1 instead of after the forward call)
 *   CFD_TRAIN_DEFER_STEM   backward phase num_layers + 1 launches nothing; cfd_fno_adam_step finishes the fc0 gradient in its own launch
 * A flag whose precondition does not hold for the shape (other widths / grids, bf16 storage, no FnoBlock) is ignored -- by all three
 * calls alike, they evaluate the same predicates -- so the caller passes the same `flags` to the forward, to every phase and to
 * cfd_fno_adam_step.  flags = 0 is exactly cfd_fno_forward_train_ex / cfd_fno_backward_phase_ex / cfd_adam_flat.  Data-parallel
 * training keeps flags = 0: a rank's gradients must be final and normalised by ITS labels before they are all-reduced.
 * cfd_fno_adam_step: param / grad / exp_avg / exp_avg_sq = the flat buffers (n floats) that `params` / `grads` point into.       */
#define CFD_TRAIN_DEFER_SCALE 1
#define CFD_TRAIN_DEFER_HEAD 2
#define CFD_TRAIN_DEFER_STEM 4
int cfd_fno_forward_train_f(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                            const cfd_fno_params* grads, const float* inputs, const float* case_params, const float* mask,
                            const float* label, float* preds, float* sums, float* coef, void* ws, int which, float upstream,
                            int act_dtype, int flags, void* stream);
int cfd_fno_backward_phase_f(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                             const cfd_fno_params* grads, const float* inputs, const float* case_params,
                             const float* mask, const float* label, const float* preds, const float* gpreds_ext,
                             const float* coef, float* sums, void* ws, int phase, int which, int act_dtype, int flags, void* stream);
int cfd_fno_adam_step(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params, const cfd_fno_params* grads,
                      const float* inputs, const float* case_params, const float* mask, const float* sums, void* ws, float* param,
                      float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, float grad_scale, int which, int act_dtype, int flags, void* stream);

/* grads: same layout as params, every tensor overwritten.  coef/gpreds_ext as in cfd_fno_head_bwd.           */
int cfd_fno_backward(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                     const cfd_fno_params* grads, const float* inputs, const float* case_params, const float* mask,
                     const float* label, const float* preds, const float* gpreds_ext, const float* coef, void* ws,
                     void* stream);

/* The same pass one phase at a time: 0 = head (+ loss gradient), 1 .. num_layers = the FnoBlocks in reverse order,
 * num_layers + 1 = lifting layer; phases in this order on one stream.  After a phase returns (is enqueued) the gradients
 * of its parameters are final, so a data-parallel trainer overlaps their all-reduce with the remaining phases
 * (the reference has no DP; src/train_auto.py:255 is its single backward call).                                 */
int cfd_fno_backward_phase(const cfd_plan* plan, const cfd_fno_shape* shape, const cfd_fno_params* params,
                           const cfd_fno_params* grads, const float* inputs, const float* case_params,
                           const float* mask, const float* label, const float* preds, const float* gpreds_ext,
                           const float* coef, void* ws, int phase, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CFDBENCH_AMD_H */
