// Process-wide dispatch overrides (which kernel variant a launch helper picks).  The environment is read ONCE, the
// first time any knob is looked up (the first versions called getenv() on every launch); afterwards cfd_tune_set() is
// the only way to change a knob (tests and timing tools use it to reach every route).  -1 = the built-in choice.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "cfd_common.h"

namespace {
struct Knob {
    int id;  // its CFD_TUNE_* index: read_env() refuses a table that is out of step with the enum
    const char* name;
    const char* env;
    std::atomic<int> value;
};
// (the entries follow the CFD_TUNE_* enum of cfd_common.h in order)
Knob g_knobs[CFD_TUNE_COUNT] = {
    {CFD_TUNE_MIX_NWV, "mix_nwv", "CFD_MIX_NWV", {-1}},            // waves per k_mix_lds workgroup; 0 = the lane = mode kernel
    {CFD_TUNE_WGRAD_WG, "wgrad_wg", "CFD_WGRAD_WG", {-1}},          // workgroups the tiled spectral weight gradient aims at
    {CFD_TUNE_FUSED_VARIANT, "fused_variant", "CFD_FUSED_VARIANT", {-1}},  // 0 = adjoint mix and weight gradient as two launches
    {CFD_TUNE_BLOCK_FUSE, "block_fuse", "CFD_BLOCK_FUSE", {-1}},      // 0 = FnoBlock backward without the fused 1x1 weight gradient
    {CFD_TUNE_GENERAL_B3, "general_b3", "CFD_GENERAL_B3", {-1}},      // 0 = grids other than 64 x 64 on the exact-fp32 generic transforms
    {CFD_TUNE_HEAD_BLOCKS, "head_blocks", "CFD_HEAD_BLOCKS", {-1}},    // workgroup cap of the head kernels (tests: several tiles per workgroup at small sizes)
    {CFD_TUNE_EXACT_FP32, "exact_fp32", "CFD_EXACT_FP32", {-1}},      // 1 = every DFT / inverse DFT on the exact-fp32 kernels (fp32 MFMA / FMA), the fused
                                                 // FnoBlock kernel replaced by its two exact passes: the price of the split-bf16 transforms
                                                 // on the record (bench.py).  The 1x1 weight gradient and the head have no exact-fp32 build.
    {CFD_TUNE_CONV6_GRID, "conv6_grid", "CFD_CONV6_GRID", {-1}},      // persistent workgroups of a conv6 forward / input-gradient launch (default 512 = two per CU;
                                                 // tests: 2, so that small shapes walk several tiles per workgroup)
    {CFD_TUNE_CONV6_WGRAD_MUL, "conv6_wgrad_mul", "CFD_CONV6_WGRAD_MUL", {-1}},  // workgroups of a conv6 weight-gradient launch in units of conv6_grid (default 1)
    {CFD_TUNE_CONVT_MFMA, "convt_mfma", "CFD_CONVT_MFMA", {-1}},      // 0 = ConvTranspose2d(2, 2) on the fp32 VALU kernels of conv.hip instead of convt6.hip
    {CFD_TUNE_CONV1_MFMA, "conv1_mfma", "CFD_CONV1_MFMA", {-1}},      // 0 = 1x1 convolutions on the general gather kernels instead of the streamed matrix-pipe kernels of conv1.hip
                                                 // (on by default since round 4: full GPU suite green with it, U-Net step -1 %: profiles/r04c_conv1_mfma.txt)
    {CFD_TUNE_SIDE_STREAM, "side_stream", "CFD_SIDE_STREAM", {-1}},    // mask of the side-stream users (side.cpp): 1 = label energy, 2 = 1x1 weight gradient; default 0 = none (measured a net loss)
    {CFD_TUNE_ACT_PIECES, "act_pieces", "CFD_ACT_PIECES", {-1}},      // bf16 pieces of the ACTIVATION operand of the FNO contractions: 3 (default since round 5: fp32-exact class, six
                                                 // MFMAs per product; cfd_common.h) or 2 (2^-16 per product, the round 1-4 default: ~8 % faster)
    {CFD_TUNE_BLOCK_GEN, "block_gen", "CFD_BLOCK_GEN", {-1}},        // 0 = the FnoBlock of grids other than 64-wide / H % 16 == 0 (66 x 65) as two passes instead of the fused kernel
    {CFD_TUNE_GEMM_TILE, "gemm_tile", "CFD_GEMM_TILE", {-1}},        // block tile of the fp32 GEMM (dense.hip): 128 = 128 x 128, 1 = 128 rows x all columns, 8 = 128 x 64 on eight waves;
                                                 // default 64 x 64 (the fastest on every product of the benchmark)
    {CFD_TUNE_GEMM_SPLITS, "gemm_splits", "CFD_GEMM_SPLITS", {-1}},    // > 0: forced split-K count of every fp32 GEMM that takes a workspace (timing sweeps: tools/exp/gemm_shapes.py)
    {CFD_TUNE_BLOCK_WIDE, "block_wide", "CFD_BLOCK_WIDE", {-1}},      // fused FnoBlock at 25 .. 32 channels: 0 = never (two passes), 1 = the round-5 pair of (8,2,4) workgroups per entry wherever
                                                 // the shape allows, 2 = the round-6 single (8,4,4) workgroup (one source-chunk buffer) wherever the shape allows; default: the
                                                 // single workgroup where it beats the two passes (64-wide grids always; general grids from 128 entries)
    {CFD_TUNE_HEAD_WAVES, "head_waves", "CFD_HEAD_WAVES", {-1}},      // waves per workgroup of the one-pass training head (k_head_bwd<.., NWV>): 4 or 8; default: 8 at 21 .. 32 channels, else 4
    {CFD_TUNE_STEM_FUSE, "stem_fuse", "CFD_STEM_FUSE", {-1}},        // 0 = the lifting layer's gradient as its own pass over a stored g_0 (k_chan_wgrad_stem) instead of sums emitted by FnoBlock 0's input-gradient kernel; 3 = the sums on the general grids (66 x 65) too (slower there: off by default)
    {CFD_TUNE_MODE_MFMA, "mode_mfma", "CFD_MODE_MFMA", {-1}},        // mode mixing / adjoint / spectral weight gradient on the fp32 matrix pipe (modes.hip, round 6): 0 = never (the batch-in-lanes
                                                 // VALU kernels of spectral.hip), 1 = wherever the shape allows (tests: small batches); default: 20 channels and >= 128 entries
    {CFD_TUNE_MODE_BC, "mode_bc", "CFD_MODE_BC", {-1}},            // batch entries per chunk of the modes.hip kernels (tests: several chunks and ragged stages at small batches)
    {CFD_TUNE_STEM_DFT, "stem_dft", "CFD_STEM_DFT", {-1}},          // the lifting layer fused into the first forward transform (k_dft_fwd64_b3<.., STEM>, round 6; 64 x 64, in_chan 2): default below 128 batch entries (rollouts), 1 = always, 0 = never
    {CFD_TUNE_GEMM_B3, "gemm_b3", "CFD_GEMM_B3", {-1}},            // 0 = tall Linear products (M >= 4096, N, K <= 1024) stay on the fp32 MFMA kernel k_gemm instead of the three-piece bf16 kernel k_rowgemm6 (round 6); 2 = k_rowgemm6 at any row count (tests)
};
std::once_flag g_once;
void read_env() {
    for (int i = 0; i < CFD_TUNE_COUNT; ++i)
        if (g_knobs[i].id != i) {
            fprintf(stderr, "cfdbench_amd: tune.cpp's knob table is out of step with the CFD_TUNE_* enum at entry %d (%s)\n", i, g_knobs[i].name ? g_knobs[i].name : "?");
            abort();
        }
    for (auto& k : g_knobs)
        if (const char* e = getenv(k.env)) k.value.store(atoi(e), std::memory_order_relaxed);
}
}  // namespace

int cfd_tune_get(int which) {
    if (which < 0 || which >= CFD_TUNE_COUNT) return -1;
    std::call_once(g_once, read_env);
    return g_knobs[which].value.load(std::memory_order_relaxed);
}

int cfd_act_pieces() { return cfd_tune_get(CFD_TUNE_ACT_PIECES) == 2 ? 2 : 3; }  // round 5: fp32-exact class is the default

extern "C" int cfd_tune_set(const char* name, int value) {
    CFD_REQUIRE(name != nullptr, CFD_ERR_INVALID_ARG, "cfd_tune_set: NULL name");
    std::call_once(g_once, read_env);
    for (auto& k : g_knobs)
        if (strcmp(k.name, name) == 0) {
            k.value.store(value, std::memory_order_relaxed);
            return CFD_OK;
        }
    cfd_set_error("cfd_tune_set: unknown knob '%s'", name);
    return CFD_ERR_INVALID_ARG;
}
