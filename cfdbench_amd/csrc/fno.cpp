// Whole Auto-FNO forward / backward as one C call each (Fno2d.forward, src/models/fno/fno2d.py:178-242, and the
// autograd pass behind loss["nmse"].backward(), src/train_auto.py:255).  Only enqueues kernels on `stream`.
//
// Activation storage: block l keeps its PRE-activation a_{l+1} = spectral(h_l) + w0 h_l + b (h_0 = a_0 = fc0 output,
// h_l = gelu(a_l) for l >= 1); consumers apply GELU on load (act_in=1) so every activation crosses HBM once per
// producer/consumer instead of once more for a standalone GELU pass, and the backward pass reads the same buffers.
#include "cfd_common.h"
#include "cfd_tail.h"

namespace {

struct Layout {
    size_t n_act;    // floats of one (B,C,H,W) activation
    size_t n_modes;  // floats of one (B,C,2*m1,m2) complex tensor
    size_t off_acts, off_xh, off_z, off_gA, off_gB, off_gh, off_scratch, off_tmp;
    size_t head_off;  // byte offset of the training head's partial records inside the scratch region (0 in inference)
    size_t scratch_bytes, total_bytes;
    int n_acts, n_xh;
};

size_t max2(size_t a, size_t b) { return a > b ? a : b; }

// byte offset of the lifting layer's sum records inside the scratch region (behind the spectral and the 1x1 weight-gradient partials,
// which the tail workgroups of the same launch are still reading)
size_t stemg_offset(const cfd_plan* p, int B, int C, int HW) {
    return cfd_align_up(cfd_align_up(cfd_spectral_wgrad_workspace_bytes(p, B, C, C), 256) + cfd_chan_wgrad_workspace_bytes(B, C, C, HW), 256);
}

Layout make_layout(const cfd_plan* p, const cfd_fno_shape* s, int training, int dt = CFD_DT_F32) {
    Layout L{};
    const size_t esz = cfd_dt_size(dt);  // bytes of one stored activation
    const size_t HW = (size_t)s->H * s->W;
    const int C = s->hidden, B = s->B;
    L.n_act = (size_t)B * C * HW;
    L.n_modes = (size_t)B * C * 2 * s->modes1 * s->modes2 * 2;
    L.n_acts = training ? s->num_layers + 1 : 2;
    L.n_xh = training ? s->num_layers : 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += cfd_align_up(bytes, 256); return o; };
    L.off_acts = take(L.n_act * esz * L.n_acts);
    L.off_xh = take(L.n_modes * sizeof(float) * L.n_xh);
    L.off_z = take(L.n_modes * sizeof(float));
    L.off_tmp = dt == CFD_DT_BF16 ? take(L.n_act * sizeof(float)) : 0;  // fp32 result of the 1x1 conv (bf16 storage path)
    size_t scratch = cfd_fno_head_workspace_bytes(B, C, s->head, s->out_chan, (int)HW);
    if (training) {
        L.off_gA = take(L.n_act * sizeof(float));
        L.off_gB = take(L.n_act * sizeof(float));
        L.off_gh = take(L.n_modes * sizeof(float));
        // both weight-gradient partial buffers of a block are alive until the block's input-gradient kernel has reduced
        // them (cfd_tail.h): spectral partials first, the 1x1-conv partials behind them
        scratch = max2(scratch, cfd_align_up(cfd_spectral_wgrad_workspace_bytes(p, B, C, C), 256) +
                                    cfd_chan_wgrad_workspace_bytes(B, C, C, (int)HW));
        scratch = max2(scratch, cfd_fno_stem_bwd_workspace_bytes(p, B, s->in_chan, s->n_case_params, C));
        // the lifting layer's sums of k_block<.., STEMG> live behind the two weight-gradient partial regions of the last block phase
        scratch = max2(scratch, stemg_offset(p, B, C, (int)HW) + cfd_int_stemg_part_bytes(p, B, C));
        // round 6: the training head's partial records behind everything a block phase writes -- with CFD_TRAIN_DEFER_HEAD they are read
        // by backward phase 1's block kernel, AFTER that phase's weight-gradient producers have written their partials into the regions above
        L.head_off = cfd_align_up(stemg_offset(p, B, C, (int)HW) + cfd_int_stemg_part_bytes(p, B, C), 256);
        scratch = max2(scratch, L.head_off + cfd_fno_head_workspace_bytes(B, C, s->head, s->out_chan, (int)HW));
    }
    L.scratch_bytes = scratch;
    L.off_scratch = take(scratch);
    L.total_bytes = off;
    return L;
}

int check_shape(const char* fn, const cfd_plan* p, const cfd_fno_shape* s) {
    CFD_REQUIRE(p && s, CFD_ERR_INVALID_ARG, "%s: NULL plan/shape", fn);
    CFD_REQUIRE(p->H == s->H && p->W == s->W && p->m1 == s->modes1 && p->m2 == s->modes2, CFD_ERR_INVALID_ARG,
                "%s: plan is for %dx%d modes (%d,%d) but shape says %dx%d modes (%d,%d)", fn, p->H, p->W, p->m1, p->m2,
                s->H, s->W, s->modes1, s->modes2);
    CFD_REQUIRE(s->B >= 1, CFD_ERR_INVALID_ARG, "%s: empty batch", fn);
    CFD_REQUIRE(s->num_layers >= 0 && s->num_layers <= CFD_MAX_LAYERS, CFD_ERR_UNSUPPORTED, "%s: num_layers=%d (max %d)", fn,
                s->num_layers, CFD_MAX_LAYERS);
    CFD_REQUIRE(s->hidden >= 1 && s->hidden <= 32, CFD_ERR_UNSUPPORTED, "%s: hidden=%d (max 32)", fn, s->hidden);
    return CFD_OK;
}

// Which deferrals of `flags` apply to this shape (the forward call, every backward phase and cfd_fno_adam_step evaluate the same thing).
struct Deferred {
    bool scale, head, stem;
};
Deferred deferred(const cfd_plan* p, const cfd_fno_shape* s, const Layout& L, char* base, int which, int dt, int flags, const void* inputs,
                  const void* mask) {
    const int B = s->B, C = s->hidden, NL = s->num_layers;
    Deferred d{};
    d.scale = (flags & CFD_TRAIN_DEFER_SCALE) && which == 1;
    const float* gA = (const float*)(base + L.off_gA);
    const float* gB = (const float*)(base + L.off_gB);
    const float* z = (const float*)(base + L.off_z);
    // backward phase 1 = FnoBlock NL-1: gcur = gA, gnext = gB, aprev = a_{NL-1} when NL > 1
    const void* aprev = NL > 1 ? (const void*)(base + L.off_acts + (size_t)(NL - 1) * L.n_act * cfd_dt_size(dt)) : nullptr;
    d.head = (flags & CFD_TRAIN_DEFER_HEAD) && dt == CFD_DT_F32 && NL >= 1 && cfd_int_block_bwd_fused(p, B, C, gA, gB, aprev, z);
    d.stem = (flags & CFD_TRAIN_DEFER_STEM) && dt == CFD_DT_F32 && NL >= 1 &&
             cfd_int_stemg_ok(p, B, C, s->in_chan, s->n_case_params, inputs, mask, z);
    return d;
}

}  // namespace

extern "C" size_t cfd_fno_workspace_bytes(const cfd_plan* p, const cfd_fno_shape* s, int training) {
    if (!p || !s || s->B < 1) return 0;
    return make_layout(p, s, training).total_bytes;
}

extern "C" size_t cfd_fno_workspace_bytes_ex(const cfd_plan* p, const cfd_fno_shape* s, int training, int act_dtype) {
    if (!p || !s || s->B < 1 || (act_dtype != CFD_DT_F32 && act_dtype != CFD_DT_BF16)) return 0;
    return make_layout(p, s, training, act_dtype).total_bytes;
}

extern "C" int cfd_fno_forward(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm, const float* inputs,
                               const float* case_params, const float* mask, const float* label, float* preds,
                               float* sums, void* ws, int training, void* stream) {
    return cfd_fno_forward_ex(p, s, prm, inputs, case_params, mask, label, preds, sums, ws, training, CFD_DT_F32, stream);
}

// bf16 activation storage (act_dtype = 1, inference only): the activations between kernels -- the lifting layer's output, every
// FnoBlock's pre-activation -- are rounded to bf16 when stored and widened when loaded; inputs, predictions, kept modes,
// weights and all arithmetic stay fp32.  Halves the activation traffic of a rollout step (3.3 -> 1.7 MB per frame).
extern "C" int cfd_fno_forward_ex(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm, const float* inputs,
                                  const float* case_params, const float* mask, const float* label, float* preds,
                                  float* sums, void* ws, int training, int act_dtype, void* stream) {
    CFD_TRY(check_shape("cfd_fno_forward", p, s));
    CFD_REQUIRE(prm && inputs && preds && ws, CFD_ERR_INVALID_ARG, "cfd_fno_forward: NULL pointer");
    CFD_REQUIRE(!label || sums, CFD_ERR_INVALID_ARG, "cfd_fno_forward: label given without sums");
    CFD_REQUIRE(act_dtype == CFD_DT_F32 || act_dtype == CFD_DT_BF16, CFD_ERR_INVALID_ARG, "cfd_fno_forward: act_dtype %d (0 = fp32, 1 = bf16)", act_dtype);
    CFD_REQUIRE(act_dtype == CFD_DT_F32 || !training, CFD_ERR_UNSUPPORTED, "cfd_fno_forward: bf16 activation storage is an inference path (training = 0)");
    const int dt = act_dtype;
    const Layout L = make_layout(p, s, training, dt);
    char* base = (char*)ws;
    const int B = s->B, C = s->hidden, HW = s->H * s->W, NL = s->num_layers;
    const size_t esz = cfd_dt_size(dt);
    auto act_buf = [&](int l) { return (void*)(base + L.off_acts + (size_t)(training ? l : (l & 1)) * L.n_act * esz); };
    auto xh_buf = [&](int l) { return (float*)(base + L.off_xh) + (size_t)(training ? l : 0) * L.n_modes; };
    float* z = (float*)(base + L.off_z);
    void* scratch = base + L.off_scratch;

    // round 6: on 64 x 64 the lifting layer rides in the first forward transform (one launch, one activation-sized read less)
    const bool sd = dt == CFD_DT_F32 && NL >= 1 && cfd_int_dft_stem_ok(p, B, s->in_chan, s->n_case_params, C, inputs, mask, act_buf(0));
    if (!sd)
        CFD_TRY(cfd_int_fno_stem_fwd(p, inputs, mask, case_params, prm->fc0_w, prm->fc0_b, act_buf(0), B, s->in_chan,
                                     s->n_case_params, C, dt, stream));
    for (int l = 0; l < NL; ++l) {  // FnoBlock.forward, fno2d.py:106-112
        const int act = l > 0;
        if (l == 0 && sd)
            CFD_TRY(cfd_int_spectral_dft_stem(p, inputs, mask, case_params, prm->fc0_w, prm->fc0_b, (float*)act_buf(0), xh_buf(0), B,
                                              s->n_case_params, C, stream));
        else
            CFD_TRY(cfd_int_spectral_dft(p, act_buf(l), xh_buf(l), B * C, act, dt, stream));
        CFD_TRY(cfd_spectral_mix(p, xh_buf(l), prm->spec_w1[l], prm->spec_w2[l], z, B, C, C, 0, stream));
        if (dt == CFD_DT_F32) {
            CFD_TRY(cfd_fno_block_fwd(p, (const float*)act_buf(l), z, prm->w0_w[l], prm->w0_b[l], (float*)act_buf(l + 1), B, C, C, act, stream));
        } else {  // bf16 storage: 1x1 conv into an fp32 scratch tensor, inverse transform added to it, ONE rounding on the store
            float* tmp = (float*)(base + L.off_tmp);
            CFD_TRY(cfd_int_chanmix(act_buf(l), prm->w0_w[l], prm->w0_b[l], tmp, B, C, C, HW, act, 0, dt, stream));
            CFD_TRY(cfd_int_spectral_idft(p, z, tmp, nullptr, act_buf(l + 1), B * C, 1, dt, stream));
        }
    }
    CFD_TRY(cfd_int_fno_head_fwd(act_buf(NL), mask, label, prm->fc1_w, prm->fc1_b, prm->fc2_w, prm->fc2_b, preds, sums,
                                 scratch, B, C, s->head, s->out_chan, HW, NL > 0, dt, stream));
    return CFD_OK;
}

// Training forward with the loss known in advance (FnoTrainEngine): everything of cfd_fno_forward(training = 1), but the
// projection head runs ONCE for both directions -- predictions, loss sums, d loss / d a_L and the head's parameter
// gradients leave the same kernel (head.hip, FUSE), so backward phase 0 is already done when this returns and the caller
// continues with cfd_fno_backward_phase(1 .. L+1).  `which` = 0 mse, 1 nmse, 2 mae; `upstream` = d objective / d loss.
extern "C" int cfd_fno_forward_train(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                     const cfd_fno_params* g, const float* inputs, const float* case_params,
                                     const float* mask, const float* label, float* preds, float* sums, float* coef, void* ws,
                                     int which, float upstream, void* stream) {
    return cfd_fno_forward_train_ex(p, s, prm, g, inputs, case_params, mask, label, preds, sums, coef, ws, which, upstream, CFD_DT_F32, stream);
}

// act_dtype = 1: bf16-storage TRAINING (the fork's other trainers offer mixed precision, src/args.py:77-80): the saved
// activations a_0 .. a_L are rounded to bf16 when stored (half the bytes of everything the backward pass re-reads); parameters,
// kept modes, gradients, accumulation and the optimiser stay fp32.  The backward pass differentiates the computation that was
// actually run, i.e. it reads the ROUNDED activations (cfd_fno_backward_phase_ex with the same act_dtype).  Each FnoBlock runs as
// 1x1 conv (fp32 scratch) + inverse transform with addend, so a stored pre-activation is rounded exactly once.
extern "C" int cfd_fno_forward_train_ex(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                        const cfd_fno_params* g, const float* inputs, const float* case_params,
                                        const float* mask, const float* label, float* preds, float* sums, float* coef, void* ws,
                                        int which, float upstream, int act_dtype, void* stream) {
    return cfd_fno_forward_train_f(p, s, prm, g, inputs, case_params, mask, label, preds, sums, coef, ws, which, upstream, act_dtype, 0, stream);
}

// flags: CFD_TRAIN_DEFER_* (include/cfdbench_amd.h) -- the single-GPU training step with its three tiny launches folded into others
extern "C" int cfd_fno_forward_train_f(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                       const cfd_fno_params* g, const float* inputs, const float* case_params,
                                       const float* mask, const float* label, float* preds, float* sums, float* coef, void* ws,
                                       int which, float upstream, int act_dtype, int flags, void* stream) {
    CFD_TRY(check_shape("cfd_fno_forward_train", p, s));
    CFD_REQUIRE(which >= 0 && which <= 2, CFD_ERR_INVALID_ARG, "cfd_fno_forward_train: which must be 0 (mse), 1 (nmse), 2 (mae)");
    CFD_REQUIRE(prm && g && inputs && label && preds && sums && ws && (coef || ((flags & CFD_TRAIN_DEFER_SCALE) && which == 1)), CFD_ERR_INVALID_ARG,
                "cfd_fno_forward_train: NULL pointer");
    CFD_REQUIRE(act_dtype == CFD_DT_F32 || act_dtype == CFD_DT_BF16, CFD_ERR_INVALID_ARG, "cfd_fno_forward_train: act_dtype %d (0 = fp32, 1 = bf16)", act_dtype);
    const int dt = act_dtype;
    const Layout L = make_layout(p, s, 1, dt);
    char* base = (char*)ws;
    const int B = s->B, C = s->hidden, HW = s->H * s->W, NL = s->num_layers;
    const size_t esz = cfd_dt_size(dt);
    auto act_buf = [&](int l) { return (void*)(base + L.off_acts + (size_t)l * L.n_act * esz); };
    auto xh_buf = [&](int l) { return (float*)(base + L.off_xh) + (size_t)l * L.n_modes; };
    float* z = (float*)(base + L.off_z);
    float* gA = (float*)(base + L.off_gA);
    void* scratch = base + L.off_scratch;
    // the label's energy and the gradient coefficients: independent of the network (scratch is free until the head); with
    // side_stream bit 1 they run beside the lifting layer on the side stream and join in front of the head (off by default:
    // the fork / join pair costs more than the 16 us it hides -- side.cpp)
    const Deferred df = deferred(p, s, L, base, which, dt, flags, inputs, mask);
    hipStream_t side = cfd_side_fork((hipStream_t)stream, df.scale ? 0 : 1);
    if (!df.scale) CFD_TRY(cfd_label_energy_coef(label, mask, sums, coef, scratch, B, s->out_chan, HW, which, upstream, side));
    const bool sd = dt == CFD_DT_F32 && NL >= 1 && cfd_int_dft_stem_ok(p, B, s->in_chan, s->n_case_params, C, inputs, mask, act_buf(0));
    if (!sd)
        CFD_TRY(cfd_int_fno_stem_fwd(p, inputs, mask, case_params, prm->fc0_w, prm->fc0_b, act_buf(0), B, s->in_chan, s->n_case_params, C,
                                     dt, stream));
    for (int l = 0; l < NL; ++l) {  // FnoBlock.forward, fno2d.py:106-112
        const int act = l > 0;
        if (l == 0 && sd)
            CFD_TRY(cfd_int_spectral_dft_stem(p, inputs, mask, case_params, prm->fc0_w, prm->fc0_b, (float*)act_buf(0), xh_buf(0), B,
                                              s->n_case_params, C, stream));
        else
            CFD_TRY(cfd_int_spectral_dft(p, act_buf(l), xh_buf(l), B * C, act, dt, stream));
        CFD_TRY(cfd_spectral_mix(p, xh_buf(l), prm->spec_w1[l], prm->spec_w2[l], z, B, C, C, 0, stream));
        if (dt == CFD_DT_F32) {
            CFD_TRY(cfd_fno_block_fwd(p, (const float*)act_buf(l), z, prm->w0_w[l], prm->w0_b[l], (float*)act_buf(l + 1), B, C, C, act, stream));
        } else {
            float* tmp = (float*)(base + L.off_tmp);
            CFD_TRY(cfd_int_chanmix(act_buf(l), prm->w0_w[l], prm->w0_b[l], tmp, B, C, C, HW, act, 0, dt, stream));
            CFD_TRY(cfd_int_spectral_idft(p, z, tmp, nullptr, act_buf(l + 1), B * C, 1, dt, stream));
        }
    }
    CFD_TRY(cfd_side_join((hipStream_t)stream, side));
    // deferred normaliser: the mse coefficient by value, sum (label*mask)^2 and the count leave the head's reduction (sums[2], sums[3])
    const float count = (float)((double)B * s->out_chan * HW);
    HeadTail ht{};
    return cfd_int_fno_head_train_f(act_buf(NL), mask, label, df.scale ? nullptr : coef, upstream / count, 0.f, df.scale ? count : 0.f,
                                    prm->fc1_w, prm->fc1_b, prm->fc2_w, prm->fc2_b, preds, sums, gA, g->fc1_w, g->fc1_b, g->fc2_w, g->fc2_b,
                                    (char*)scratch + L.head_off, B, C, s->head, s->out_chan, HW, NL > 0, dt, stream, df.head ? &ht : nullptr);
}

// One phase of the backward pass: 0 = projection head (+ loss gradient), 1 .. L = FnoBlock L-phase (the blocks in reverse
// order), L+1 = lifting layer.  Phases must run in this order on one stream; the running input gradient alternates
// between two workspace buffers, so a phase finds its operands from its index alone.  After phase k the gradients of
// that phase's parameters are final -- a data-parallel trainer can start their all-reduce while later phases compute.
extern "C" int cfd_fno_backward_phase(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                      const cfd_fno_params* g, const float* inputs, const float* case_params,
                                      const float* mask, const float* label, const float* preds,
                                      const float* gpreds_ext, const float* coef, void* ws, int phase, void* stream) {
    return cfd_fno_backward_phase_ex(p, s, prm, g, inputs, case_params, mask, label, preds, gpreds_ext, coef, ws, phase, CFD_DT_F32, stream);
}

// act_dtype = 1 continues cfd_fno_forward_train_ex(act_dtype = 1): phases 1 .. L+1 read the bf16 activations that pass stored.  (Phase 0,
// the stand-alone head backward, exists for fp32 storage only: the bf16 path always runs the one-pass training head.)
extern "C" int cfd_fno_backward_phase_ex(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                         const cfd_fno_params* g, const float* inputs, const float* case_params,
                                         const float* mask, const float* label, const float* preds,
                                         const float* gpreds_ext, const float* coef, void* ws, int phase, int act_dtype, void* stream) {
    return cfd_fno_backward_phase_f(p, s, prm, g, inputs, case_params, mask, label, preds, gpreds_ext, coef, nullptr, ws, phase, 0, act_dtype, 0, stream);
}

// flags (CFD_TRAIN_DEFER_*): phase 1 carries the head's reduction left behind by cfd_fno_forward_train_f (which needs `sums` and `which`
// again: the head job writes the loss sums); phase L + 1 launches nothing when cfd_fno_adam_step finishes the lifting layer's gradient.
extern "C" int cfd_fno_backward_phase_f(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                        const cfd_fno_params* g, const float* inputs, const float* case_params,
                                        const float* mask, const float* label, const float* preds,
                                        const float* gpreds_ext, const float* coef, float* sums, void* ws, int phase, int which,
                                        int act_dtype, int flags, void* stream) {
    CFD_TRY(check_shape("cfd_fno_backward_phase", p, s));
    CFD_REQUIRE(prm && g && inputs && ws, CFD_ERR_INVALID_ARG, "cfd_fno_backward_phase: NULL pointer");
    CFD_REQUIRE(act_dtype == CFD_DT_F32 || act_dtype == CFD_DT_BF16, CFD_ERR_INVALID_ARG, "cfd_fno_backward_phase: act_dtype %d (0 = fp32, 1 = bf16)", act_dtype);
    const int dt = act_dtype;
    const Layout L = make_layout(p, s, 1, dt);
    char* base = (char*)ws;
    const int B = s->B, C = s->hidden, HW = s->H * s->W, NL = s->num_layers;
    const size_t esz = cfd_dt_size(dt);
    CFD_REQUIRE(phase >= 0 && phase <= NL + 1, CFD_ERR_INVALID_ARG, "cfd_fno_backward_phase: phase %d outside 0..%d", phase, NL + 1);
    auto act_buf = [&](int l) { return (void*)(base + L.off_acts + (size_t)l * L.n_act * esz); };
    auto xh_buf = [&](int l) { return (float*)(base + L.off_xh) + (size_t)l * L.n_modes; };
    float* z = (float*)(base + L.off_z);
    float* gA = (float*)(base + L.off_gA);
    float* gB = (float*)(base + L.off_gB);
    float* gh = (float*)(base + L.off_gh);
    void* scratch = base + L.off_scratch;
    if (phase == 0) {
        CFD_REQUIRE(dt == CFD_DT_F32, CFD_ERR_UNSUPPORTED, "cfd_fno_backward_phase: phase 0 with bf16 storage (the head ran in cfd_fno_forward_train_ex)");
        return cfd_fno_head_bwd((const float*)act_buf(NL), mask, label, preds, gpreds_ext, coef, prm->fc1_w, prm->fc1_b, prm->fc2_w, gA,
                                g->fc1_w, g->fc1_b, g->fc2_w, g->fc2_b, scratch, B, C, s->head, s->out_chan, HW, NL > 0,
                                stream);
    }
    const int done = phase - 1;  // blocks already processed: the gradient sits in gA after an even count
    float* gcur = (done & 1) ? gB : gA;
    float* gnext = (done & 1) ? gA : gB;
    // Round 5: where the fused FnoBlock kernel runs the last block phase (l = 0), it emits the six per-(entry, channel) sums the lifting
    // layer's gradient needs instead of storing g_0 for a pass that reads it back (cfd_tail.h: CfdStemG): one activation-sized write and
    // the k_chan_wgrad_stem launch less.  Both phases evaluate the same predicate.
    const bool stemg = dt == CFD_DT_F32 && NL >= 1 &&
                       cfd_int_stemg_ok(p, B, C, s->in_chan, s->n_case_params, inputs, mask, z);
    float* stem_part = (float*)((char*)scratch + stemg_offset(p, B, C, HW));
    const Deferred df = deferred(p, s, L, base, which, dt, flags, inputs, mask);
    CFD_REQUIRE(!df.head || sums, CFD_ERR_INVALID_ARG, "cfd_fno_backward_phase: CFD_TRAIN_DEFER_HEAD needs the `sums` of the forward call");
    if (phase == NL + 1) {
        if (stemg && df.stem) return CFD_OK;  // cfd_fno_adam_step's launch finishes the lifting layer's gradient
        if (stemg)
            return cfd_int_stemg_combine(p, stem_part, case_params, g->fc0_w, g->fc0_b, B, C, s->in_chan, s->n_case_params, stream);
        return cfd_fno_stem_bwd(p, gcur, inputs, mask, case_params, g->fc0_w, g->fc0_b, scratch, B, s->in_chan,
                                s->n_case_params, C, stream);
    }
    const int l = NL - phase;
    const int act = l > 0;
    // gcur = d loss / d a_{l+1}
    char* scratch2 = (char*)scratch + cfd_align_up(cfd_spectral_wgrad_workspace_bytes(p, B, C, C), 256);
    if (dt == CFD_DT_BF16) {
        // two passes for the input gradient (1x1 conv transposed into the fp32 scratch tensor, inverse transform + addend
        // [* gelu'(a_l), a_l read as bf16]); the weight-gradient producers reduce their own partial sums
        float* tmp = (float*)(base + L.off_tmp);
        CFD_TRY(cfd_spectral_dft(p, gcur, gh, B * C, 0, stream));
        CFD_TRY(cfd_int_spectral_mix_adj_wgrad(p, xh_buf(l), gh, prm->spec_w1[l], prm->spec_w2[l], z, g->spec_w1[l],
                                               g->spec_w2[l], scratch, B, C, C, stream, nullptr));
        CFD_TRY(cfd_int_chan_wgrad_dt(gcur, act_buf(l), g->w0_w[l], g->w0_b[l], scratch2, B, C, C, HW, act, dt, stream, nullptr));
        CFD_TRY(cfd_chanmix(gcur, prm->w0_w[l], nullptr, tmp, B, C, C, HW, 0, 1, stream));
        return cfd_int_spectral_idft_grad(p, z, tmp, act ? act_buf(l) : nullptr, gnext, B * C, dt, stream);
    }
    // the reductions of both weight gradients ride in front of the input-gradient kernel's launch (cfd_tail.h); whatever
    // a producer could not defer it has already reduced itself.  The 1x1 weight gradient needs only gcur and a_l; with
    // side_stream bit 2 it runs on the side stream (side.cpp) beside the transform and the mode-domain kernel.  OFF by default:
    // measured (profiles/r04a_side_stream_ab.txt) the two do run concurrently, but the latency-bound mode-domain kernel then
    // takes 53 us instead of 26 -- it needs the wave slots the streaming kernel occupies -- and the phase is no shorter.
    hipStream_t side = cfd_side_fork((hipStream_t)stream, 2);
    CfdReduceTail tail{};
    CFD_TRY(cfd_int_chan_wgrad(gcur, (const float*)act_buf(l), g->w0_w[l], g->w0_b[l], scratch2, B, C, C, HW, act, side, &tail.chan));
    CFD_TRY(cfd_spectral_dft(p, gcur, gh, B * C, 0, stream));
    CFD_TRY(cfd_int_spectral_mix_adj_wgrad(p, xh_buf(l), gh, prm->spec_w1[l], prm->spec_w2[l], z, g->spec_w1[l],
                                           g->spec_w2[l], scratch, B, C, C, stream, &tail.spec));
    CFD_TRY(cfd_side_join((hipStream_t)stream, side));  // the block kernel reduces the 1x1 partial sums
    if (df.head && phase == 1)  // the reduction cfd_fno_forward_train_f left behind (records at head_off: nothing of this phase touched them)
        tail.head = cfd_int_head_tail((char*)scratch + L.head_off, g->fc1_w, g->fc1_b, g->fc2_w, g->fc2_b, sums, B, C, s->out_chan, HW,
                                      df.scale ? (float)((double)B * s->out_chan * HW) : 0.f);
    tail.nblk = (tail.spec.part || tail.chan.part || tail.head.part) ? 128 : 0;
    const CfdStemG sg{(l == 0 && stemg) ? inputs : nullptr, mask, p->d_gx, p->d_gy, stem_part, s->in_chan};
    return cfd_int_fno_block_bwd_input(p, gcur, z, prm->w0_w[l], act ? (const float*)act_buf(l) : nullptr, gnext, B, C, C, stream, &tail, &sg);
}

extern "C" int cfd_fno_backward(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm,
                                const cfd_fno_params* g, const float* inputs, const float* case_params,
                                const float* mask, const float* label, const float* preds, const float* gpreds_ext,
                                const float* coef, void* ws, void* stream) {
    CFD_TRY(check_shape("cfd_fno_backward", p, s));
    for (int phase = 0; phase <= s->num_layers + 1; ++phase)
        CFD_TRY(cfd_fno_backward_phase(p, s, prm, g, inputs, case_params, mask, label, preds, gpreds_ext, coef, ws, phase,
                                       stream));
    return CFD_OK;
}

// The optimiser launch of the fused training step: Adam over the flat buffers, the deferred nMSE normaliser (sums[3] / sums[2] on top of
// grad_scale) and the lifting layer's gradient rows (see include/cfdbench_amd.h).  `params` / `grads` point into `param` / `grad`.
extern "C" int cfd_fno_adam_step(const cfd_plan* p, const cfd_fno_shape* s, const cfd_fno_params* prm, const cfd_fno_params* g,
                                 const float* inputs, const float* case_params, const float* mask, const float* sums, void* ws,
                                 float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, float grad_scale, int which, int act_dtype,
                                 int flags, void* stream) {
    CFD_TRY(check_shape("cfd_fno_adam_step", p, s));
    CFD_REQUIRE(prm && g && ws && param && grad, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: NULL pointer");
    CFD_REQUIRE(act_dtype == CFD_DT_F32 || act_dtype == CFD_DT_BF16, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: act_dtype %d", act_dtype);
    const Layout L = make_layout(p, s, 1, act_dtype);
    char* base = (char*)ws;
    const Deferred df = deferred(p, s, L, base, which, act_dtype, flags, inputs, mask);
    CFD_REQUIRE(!df.scale || sums, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: CFD_TRAIN_DEFER_SCALE needs the `sums` of the forward call");
    StemAdamJob job{};
    if (df.stem) {
        const int B = s->B, C = s->hidden, HW = s->H * s->W;
        CFD_REQUIRE(case_params || s->n_case_params == 0, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: NULL case_params");
        const float* gw = (const float*)g->fc0_w;
        const float* gb = (const float*)g->fc0_b;
        CFD_REQUIRE(gw >= grad && gb >= grad && (size_t)(gw - grad) < n && (size_t)(gb - grad) < n, CFD_ERR_INVALID_ARG,
                    "cfd_fno_adam_step: grads->fc0 does not point into the flat gradient buffer");
        CFD_REQUIRE((const float*)prm->fc0_w - param == gw - grad && (const float*)prm->fc0_b - param == gb - grad, CFD_ERR_INVALID_ARG,
                    "cfd_fno_adam_step: params and grads are laid out differently");
        const int spl = cfd_int_stemg_splits(p, B);
        job = StemAdamJob{(const float*)(base + L.off_scratch + stemg_offset(p, B, C, HW)), case_params, B * spl, spl, s->n_case_params,
                          s->in_chan, C, (long)(gw - grad), (long)(gb - grad)};
    }
    return cfd_int_adam_flat_f(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                               df.scale ? sums : nullptr, df.stem ? &job : nullptr, stream);
}
