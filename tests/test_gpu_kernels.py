"""MI355X: every C-ABI entry point of the shipped library against the oracle (same checks as the CPU emulator run,
tests/test_emul_kernels.py, at larger sizes)."""
import numpy as np
import pytest

from tests import kernel_checks as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from tests.backends import TorchBackend
    return TorchBackend()


def _assert_all(res, tol=K.TOL):
    bad = {k: v for k, v in res.items() if not (v < tol)}
    assert not bad, f"parity failures (tol {tol}): {bad}; all: {res}"


def test_library_is_the_hip_extension(be):
    from cfdbench_amd import _lib
    assert _lib.lib_path().exists()
    assert be.api.missing == [] and be.api.version() >= 100


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 3, 5, 64, 64), (3, 4, 3, 66, 65), (4, 20, 20, 64, 64), (1, 32, 32, 64, 64)])
def test_spectral_fwd_bwd(be, B, Cin, Cout, H, W):
    _assert_all(K.check_spectral(be, B, Cin, Cout, H, W))


@pytest.mark.parametrize("H,W", [(66, 65), (32, 64), (48, 64), (70, 80), (20, 24), (65, 33), (64, 64)])
@pytest.mark.parametrize("general", [1, 0])
def test_spectral_general_widths(be, H, W, general):
    """Grids other than 64 x 64: the general-width split-bf16 transforms (columns dealt as y = 16 j + n) and, with the
    general_b3 knob at 0, the exact-fp32 generic kernels they replace.  Many images per wave (persistent loop + ring)."""
    with K.tuned(be, general_b3=general):
        _assert_all(K.check_spectral(be, 23, 5, 4, H, W, m1=min(12, H // 2), m2=min(12, W // 2)))
        _assert_all(K.check_idft_epilogues(be, 41, H, W, m1=min(12, H // 2), m2=min(12, W // 2)))


@pytest.mark.parametrize("B,Cin,Cout", [(37, 20, 20), (9, 12, 7), (33, 24, 24), (70, 32, 32), (10, 5, 20), (256, 20, 20)])
def test_mix_and_spectral_wgrad(be, B, Cin, Cout):
    _assert_all(K.check_mix_wgrad(be, B, Cin, Cout))


@pytest.mark.parametrize("B,C,nwv,want_wg,fused", [(70, 32, "4", "256", "1"), (37, 20, "2", "36", "1"), (300, 20, "8", "80", "0"),
                                                   (256, 20, "0", "256", "1")])
def test_mix_and_spectral_wgrad_kernel_routes(be, B, C, nwv, want_wg, fused):
    """Every route of the mode-domain entry points at sizes the defaults would not pick: k_mix_lds workgroup shapes
    (mix_nwv knob, 0 = the lane = mode kernel), weight-gradient chunk sizes (wgrad_wg), fused vs two launches (cfd_tune_set)."""
    with K.tuned(be, mix_nwv=int(nwv), wgrad_wg=int(want_wg), fused_variant=int(fused), mode_mfma=0):
        _assert_all(K.check_mix_wgrad(be, B, C, C))


@pytest.mark.parametrize("B,bc,C", [(256, -1, 20), (300, -1, 20), (128, 16, 20), (37, -1, 20), (200, 24, 20), (21, 5, 20), (512, -1, 20),
                                    (256, -1, 32), (130, -1, 32), (41, 9, 32)])
def test_mix_and_spectral_wgrad_on_the_matrix_pipe(be, B, bc, C):
    """modes.hip (round 6): mixing, adjoint and spectral weight gradient of 20 channels as real GEMMs on v_mfma_f32_16x16x4_f32 --
    the default from 128 entries, forced here at every size (mode_mfma = 1); mode_bc shrinks the chunk so that small batches reach
    several chunks, ragged last stages and chunks whose last K-step is partly zeros."""
    with K.tuned(be, mode_mfma=1, mode_bc=bc):  # (32 channels: mix and adjoint only -- the weight gradient stays on the VALU kernels there)
        _assert_all(K.check_mix_wgrad(be, B, C, C))


def test_matrix_pipe_mode_kernels_agree_with_the_valu_kernels(be):
    """Both routes are fp32-exact class: their results differ only in summation order (1e-13 relative)."""
    with K.tuned(be, mode_mfma=1):
        a = K.check_mix_wgrad(be, 256, 20, 20, seed=3)
    with K.tuned(be, mode_mfma=0):
        b = K.check_mix_wgrad(be, 256, 20, 20, seed=3)
    _assert_all(a)
    _assert_all(b)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(5, 20, 20, 64, 64), (3, 6, 7, 32, 64), (2, 3, 5, 66, 65), (3, 32, 32, 64, 64), (2, 14, 9, 48, 64),
                                            # round 4, the general fused kernel (pitch != 64, tail columns, ragged / five row tiles)
                                            (7, 20, 20, 66, 65), (300, 20, 20, 66, 65), (2, 6, 7, 50, 64), (3, 20, 20, 66, 67), (2, 5, 4, 70, 68),
                                            (2, 12, 20, 80, 65), (3, 16, 9, 33, 66), (2, 24, 24, 66, 65)])
def test_fused_block(be, B, Cin, Cout, H, W):
    _assert_all(K.check_block(be, B, Cin, Cout, H, W))


def test_general_fused_block_equals_its_two_passes(be):
    """block_gen = 0 routes the 66 x 65 FnoBlock through k_chanmix_b3 + k_idft_g; both routes hold the oracle."""
    with K.tuned(be, block_gen=0):
        _assert_all(K.check_block(be, 2, 20, 20, 66, 65))


@pytest.mark.parametrize("Bbig,Bsmall,C,H,W", [(160, 7, 20, 64, 64), (150, 40, 8, 64, 64), (148, 100, 32, 32, 64),
                                               (300, 64, 20, 66, 65), (300, 17, 20, 66, 65), (290, 100, 6, 50, 66)])  # five / odd tile counts: uneven splits
def test_fused_block_batch_split_is_bitwise_neutral(be, Bbig, Bsmall, C, H, W):
    res = K.check_block_batch_split(be, Bbig, Bsmall, C, H, W)
    assert res["fwd_bitwise"] == 0.0 and res["bwd_bitwise"] == 0.0, res


@pytest.mark.parametrize("B,C,L,H,W", [(3, 20, 2, 66, 65), (2, 32, 4, 66, 65), (2, 20, 4, 64, 64), (2, 8, 1, 48, 64)])
def test_bf16_activation_storage_forward(be, B, C, L, H, W):
    """BASELINE configs[4] storage format: cfd_fno_forward_ex(act_dtype = bf16) against the oracle with the same rule (the
    lifting layer's output and every pre-activation rounded to bf16 once, where stored)."""
    res = K.check_fno_bf16_storage(be, B, C, L, H, W)
    info = {k: v for k, v in res.items() if k.startswith("info:")}
    print(f"bf16 storage {B}x{C}x{H}x{W} L={L}:", {k: f"{v:.2e}" for k, v in res.items()})
    assert res.pop("bf16_loss") < 1e-5  # fp32 loss sums
    _assert_all({k: v for k, v in res.items() if not k.startswith("info:")}, tol=1e-7)
    assert 1e-7 < info["info:bf16_vs_f32"] < 1e-3


@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
def test_idft_epilogues(be, H, W):
    _assert_all(K.check_idft_epilogues(be, 37, H, W))


@pytest.mark.parametrize("Ci,Co,HW,act", [(20, 20, 4096, True), (6, 8, 66 * 65, False), (32, 32, 4096, True),
                                          (20, 20, 4290, True), (13, 20, 1001, False)])
def test_chanmix_and_wgrad(be, Ci, Co, HW, act):
    _assert_all(K.check_chanmix(be, 3, Ci, Co, HW, act))


@pytest.mark.parametrize("H,W,border,C", [(64, 64, False, 20), (66, 65, True, 20), (64, 64, True, 32), (66, 65, False, 6)])
def test_stem(be, H, W, border, C):
    _assert_all(K.check_stem(be, 3, H, W, 5, C, border))


@pytest.mark.parametrize("C,HW,act,which,ext", [(20, 4096, True, "nmse", False), (6, 4290, False, "mse", True),
                                                (32, 4096, True, "mae", False), (20, 4290, True, "nmse", True),
                                                (8, 1000, True, "nmse", False)])
def test_head(be, C, HW, act, which, ext):
    res = K.check_head(be, 3, C, HW, act, which, ext)
    _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")})
    assert res["sums"] < 1e-5 and res["scores"] < 1e-5


@pytest.mark.parametrize("C,HW,act,which,cap", [(20, 4096, True, "nmse", -1), (6, 4290, False, "mse", -1), (32, 4096, True, "mae", -1),
                                                 (20, 4290, True, "nmse", 3), (8, 1000, True, "nmse", 1), (20, 64, True, "mse", -1)])
def test_head_train_one_pass(be, C, HW, act, which, cap):
    """Both directions of the head in one kernel (cfd_fno_head_train): the same references as test_head."""
    with K.tuned(be, head_blocks=cap):
        res = K.check_head_train(be, 3, C, HW, act, which)
    assert res.pop("sums") < 1e-5
    _assert_all(res)


@pytest.mark.parametrize("cap", [1, 3, 16])
def test_head_several_tiles_per_workgroup(be, cap):
    """The head kernels' persistent loops (several tiles per workgroup: staged planes and f'(a) handed from tile to tile)."""
    with K.tuned(be, head_blocks=cap):
        res = K.check_head(be, 5, 20, 4096, True, "nmse", False)
    _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")})


def test_loss_and_adam(be):
    res = K.check_loss_and_adam(be, n=1_000_003)
    assert res["sums"] < 1e-5
    assert res["adam_delta"] < 1e-9


@pytest.mark.parametrize("B,C,L,H,W,border", [(2, 8, 2, 64, 64, False), (2, 6, 2, 66, 65, True), (4, 20, 4, 64, 64, False),
                                              (2, 20, 4, 66, 65, True), (1, 32, 3, 64, 64, True)])
def test_fno_whole_model_vs_oracle(be, B, C, L, H, W, border):
    res = K.check_fno_vs_oracle(be, B, C, L, H, W, border=border)
    loss_err = res.pop("nmse_loss")
    assert loss_err < 1e-5
    _assert_all(res, 1e-9)
    assert max(res.values()) < K.NORTH_STAR_TOL


@pytest.mark.parametrize("B,C,L,H,W", [(3, 20, 2, 64, 64), (2, 6, 2, 66, 65)])
def test_exact_fp32_transform_route(be, B, C, L, H, W):
    """cfd_tune_set("exact_fp32", 1): every DFT / inverse DFT on the exact-fp32 kernels, the fused FnoBlock replaced by its two
    passes -- the route bench.py times beside the split-bf16 default.  Same oracle, and tighter where only transforms differ."""
    with K.tuned(be, exact_fp32=1):
        res = K.check_fno_vs_oracle(be, B, C, L, H, W, border=True)
        assert res.pop("nmse_loss") < 1e-5
        _assert_all(res, 1e-9)
        sp = K.check_spectral(be, 5, 4, 3, H, W)
        _assert_all(sp, 1e-12)
        _assert_all(K.check_block(be, 3, C, C, H, W), 1e-12)


def test_mfma_operand_layout_is_transpose_detecting(be):
    """A = I against an ASYMMETRIC second operand: a swapped row/column map in any of the chained MFMA stages would
    show up as a transposed spectrum (cdna_hip_programming.md section 3 'Always A=I-check with asymmetric B')."""
    import torch
    api = be.api
    H = W = 64
    plan = api.plan_create(H, W, 12, 12)
    try:
        x = np.zeros((1, 1, H, W), np.float32)
        x[0, 0, 3, 5] = 1.0  # delta -> X^[k,l] = exp(-2 pi i (3k/H + 5l/W)): asymmetric in (k,l)
        dx = be.dev(x)
        xh = be.zeros((1, 1, 24, 12), np.complex64)
        api.call("cfd_spectral_dft", plan, be.ptr(dx), be.ptr(xh), 1, 0, be.stream)
        be.sync()
        from oracle import fno_oracle as O
        k = O.kept_rows(H, 12)[:, None]
        l = np.arange(12)[None, :]
        ref = np.exp(-2j * np.pi * (3 * k / H + 5 * l / W))
        assert np.max(np.abs(be.host(xh)[0, 0] - ref)) < 1e-5
    finally:
        api.plan_destroy(plan)


def test_error_paths(be):
    from cfdbench_amd._capi import CfdError
    with pytest.raises(CfdError):
        be.api.plan_create(64, 64, 40, 12)
    with pytest.raises(CfdError):
        be.api.plan_create(64, 200, 12, 12)
    plan = be.api.plan_create(64, 64, 12, 12)
    with pytest.raises(CfdError):
        be.api.call("cfd_spectral_dft", plan, None, None, 4, 0, be.stream)
    be.api.plan_destroy(plan)


@pytest.mark.parametrize("M,N,Kd,ta,tb", [(512, 100, 4295, 0, 1), (4290, 100, 100, 0, 0), (100, 4295, 512, 1, 0), (100, 100, 4290, 1, 1), (70, 45, 37, 0, 0)])
def test_gemm(be, M, N, Kd, ta, tb):
    _assert_all(K.check_gemm(be, M, N, Kd, ta, tb))


@pytest.mark.parametrize("tile", [64, 128, 1, 8])
@pytest.mark.parametrize("M,N,Kd,ta,tb", [(4290, 100, 100, 0, 0), (70001, 200, 200, 0, 1), (300, 515, 4000, 1, 0), (130, 129, 77, 1, 1), (70, 45, 37, 0, 0)])
def test_gemm_both_block_tiles(be, M, N, Kd, ta, tb, tile):
    """Every block tile of k_gemm (64 x 64, 128 x 128, 1 = 128 rows x all columns, 8 = 128 x 64 on eight waves) on every storage form, whatever launch_gemm would
    pick for the shape."""
    with K.tuned(be, gemm_tile=tile):
        _assert_all(K.check_gemm(be, M, N, Kd, ta, tb))


@pytest.mark.parametrize("M,K_in,N,in_act", [(4290, 200, 200, "relu"), (300, 100, 100, "gelu"), (129, 33, 70, "swish"), (2050, 5, 530, "tanh")])
def test_linear_input_gradient_leaves_as_the_previous_layers_dz(be, M, K_in, N, in_act):
    res = K.check_linear_chain_bwd(be, M, K_in, N, in_act)
    assert res.pop("differs_from_two_passes") == 0 and res.pop("gw_differs") == 0
    _assert_all(res)


@pytest.mark.parametrize("specs", [[(512, [100] * 8, "relu", False, True), (4290, [2] + [100] * 8, "relu", False, False)], [(128, [100] * 8, "relu", True, True), (128, [100] * 8, "relu", True, True), (4096, [2] + [100] * 8, "relu", False, False)], [(70, [5, 12, 20, 7], "swish", False, True), (333, [3, 128, 128, 1], "gelu", True, True)]])
def test_ffn_stacks_in_one_launch_equal_the_single_calls(be, specs):
    """cfd_ffn_stacks_fwd / _bwd (the branch and trunk stacks of a DeepONet variant as one launch per direction) == the single-stack
    calls, bit for bit."""
    res = K.check_ffn_stacks(be, specs)
    assert all(v == 0 for v in res.values()), {k: v for k, v in res.items() if v}


@pytest.mark.parametrize("M,K_in,N,act", [(131072, 200, 200, "tanh"), (4290, 100, 100, "relu"), (129, 33, 70, "swish"), (2050, 5, 530, "none")])  # (the Auto-FFN layer with a smooth activation: among 26 M pre-activations a handful sit within an fp32 rounding of the ReLU kink)
@pytest.mark.parametrize("tile", [128, 1])
def test_linear_act_on_the_large_tiles(be, M, K_in, N, act, tile):
    with K.tuned(be, gemm_tile=tile):
        _assert_all(K.check_linear(be, M, K_in, N, act))


@pytest.mark.parametrize("R,dims,act,act_last,with_gx", [(4290, [2, 100, 100, 100, 100, 100, 100, 100, 100], "relu", False, False), (512, [100] * 8, "relu", False, True),
                                                            (70, [5, 12, 20, 7], "tanh", False, True), (333, [3, 128, 128, 1], "gelu", True, True), (65, [128, 64, 128], "swish", False, True),
                                                            (1, [7, 9, 4], "none", False, True)])
def test_ffn_stack(be, R, dims, act, act_last, with_gx):
    """A whole Linear(+activation) stack per kernel (cfd_ffn_stack_fwd / _bwd) against the fp64 layer-by-layer restatement."""
    _assert_all(K.check_ffn_stack(be, R, dims, act, act_last, with_gx))


@pytest.mark.parametrize("M,K_in,N,act", [(512, 4295, 100, "relu"), (4290, 2, 100, "relu"), (4290, 100, 100, "tanh"), (300, 100, 100, "gelu"), (129, 33, 70, "swish"), (66, 100, 16, "none"),
                                          (70001, 9, 100, "relu"), (2050, 5, 530, "none")])  # two-stage bias gradient: > 1024 chunks of rows; N > 512
def test_linear_act(be, M, K_in, N, act):
    _assert_all(K.check_linear(be, M, K_in, N, act))


@pytest.mark.parametrize("B,P,Kq,HW,with_q", [(512, 100, 4290, 4290, False), (37, 100, 1000, 4290, True), (3, 24, 256, 256, False)])
def test_deeponet_inner(be, B, P, Kq, HW, with_q):
    res = K.check_deeponet_inner(be, B, P, Kq, HW, with_q)
    assert res.pop("gbias") < 1e-5  # a relative error of one fp32 sum, not an nMSE
    _assert_all(res)


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(3, 11, 12, 64, 64, 3), (2, 24, 12, 33, 32, 3), (2, 96, 192, 5, 4, 3), (2, 192, 96, 8, 8, 3), (1, 8, 64, 20, 21, 7), (1, 64, 16, 17, 16, 7), (2, 12, 2, 16, 16, 1), (32, 32, 32, 36, 36, 5), (4, 8, 32, 68, 68, 5), (3, 32, 32, 8, 8, 5)])
def test_conv2d_replicate(be, B, Ci, Co, H, W, ks):
    _assert_all(K.check_conv2d(be, B, Ci, Co, H, W, ks))


@pytest.mark.parametrize("grid,B,Ci,Co,H,W,ks", [(2, 20, 12, 12, 64, 64, 3), (4, 37, 20, 40, 16, 16, 3), (2, 24, 8, 64, 30, 33, 7), (8, 11, 64, 16, 17, 16, 7),
                                                  (1, 130, 192, 96, 4, 4, 3), (2, 9, 3, 5, 2, 3, 3)])
def test_conv2d_persistent_workgroups_walk_many_tiles(be, grid, B, Ci, Co, H, W, ks):
    """conv6.hip with a handful of persistent workgroups (the conv6_grid knob): every workgroup walks many (tile, chunk)
    iterations -- register prefetch, deferred stores, accumulator hand-over, partial channel chunks, image groups of 8 with a
    ragged last group, a grid without interior (2 x 3: the full fold) -- at the tolerance of the exact-fp32 kernels."""
    with K.tuned(be, conv6_grid=grid):
        _assert_all(K.check_conv2d(be, B, Ci, Co, H, W, ks))


@pytest.mark.parametrize("grid,B,Ci,Co,H,W,ks", [(-1, 128, 12, 12, 64, 64, 3), (4, 64, 24, 48, 32, 32, 3), (-1, 32, 8, 64, 64, 64, 3), (-1, 3, 5, 7, 9, 10, 3)])
def test_conv_emits_batchnorm_statistics(be, grid, B, Ci, Co, H, W, ks):
    """cfd_conv2d_fwd_stats + cfd_batchnorm_fwd_stats against the oracle's conv -> training-mode BatchNorm (running statistics
    included), at a full-size U-Net layer, with few persistent workgroups, at a wide layer and at a tiny one (k = 3: the kernel sizes the
    reference follows with a BatchNorm, src/models/unet.py:20-30)."""
    with K.tuned(be, conv6_grid=grid):
        res = K.check_conv_bn_stats(be, B, Ci, Co, H, W, ks)
    assert res is not None
    _assert_all(res)


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(128, 12, 12, 64, 64, 3), (64, 24, 48, 32, 32, 3)])
def test_conv_batchnorm_statistics_with_mean_far_from_bias(be, B, Ci, Co, H, W, ks):
    """Output channels with |mean - bias| ~ 100 std: per-slot shifted records, no cancelling difference in the variance."""
    res = K.check_conv_bn_stats(be, B, Ci, Co, H, W, ks, offset=5.0, spread=0.1)
    assert res is not None
    # (y itself is bounded by the fp32 rounding of `out`, ~1e-7 of a mean that is 100 std: nMSE ~1e-8 whatever the statistics do)
    assert res["run_mean"] < 1e-10 and res["run_var"] < 1e-10 and res["out"] < 1e-10 and res["y"] < 1e-6, res


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(32, 8, 32, 64, 64, 5), (32, 32, 32, 32, 32, 5), (8, 32, 32, 4, 4, 5), (3, 11, 12, 33, 32, 3), (2, 16, 64, 20, 21, 7)])
def test_conv2d_zero_padding(be, B, Ci, Co, H, W, ks):
    res = K.check_conv2d_zeropad(be, B, Ci, Co, H, W, ks)
    if res is None:  # few pixel tiles: the input-gradient pass would split its channel chunks over workgroups -- route not taken
        assert (B, H) != (32, 64) and (B, H) != (3, 33)  # (the branch's first layers and a mid-size k = 3 layer must be covered)
        assert be.api.size("cfd_conv2d_zeropad_supported", B, Ci, Co, H, W, ks) == 0
    else:
        _assert_all(res)


def test_conv_weights_prepared_in_one_batch(be):
    """cfd_conv2d_wprep_batch: the fragments of all 18 3x3 layers of the configs[2] U-Net (both forms) and of ResNet's 7x7 layers
    from single launches; every layer then computes bit for bit what it computes preparing its own weights."""
    unet = [(16, 11, 12, 64, 64), (16, 12, 12, 64, 64), (16, 12, 24, 32, 32), (16, 24, 24, 32, 32), (16, 24, 48, 16, 16),
            (16, 48, 48, 16, 16), (16, 48, 96, 8, 8), (16, 96, 96, 8, 8), (16, 96, 192, 4, 4), (16, 192, 192, 4, 4), (16, 192, 96, 8, 8),
            (16, 96, 48, 16, 16), (16, 48, 24, 32, 32), (16, 24, 12, 64, 64)]
    assert K.check_conv_prepared(be, [(B, Ci, Co, H, W, 3) for B, Ci, Co, H, W in unet]) == 0
    assert K.check_conv_prepared(be, [(4, 11, 64, 64, 64, 7), (4, 64, 16, 64, 64, 7), (4, 16, 64, 33, 31, 7), (4, 64, 2, 64, 64, 7)], seed=39) == 0


@pytest.mark.parametrize("B,C,H,W,training,relu", [(4, 12, 64, 64, True, True), (3, 192, 4, 4, True, True), (2, 24, 33, 32, False, True), (3, 5, 7, 9, True, False)])
def test_batchnorm_relu(be, B, C, H, W, training, relu):
    _assert_all(K.check_batchnorm(be, B, C, H, W, training, relu))


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 24, 12, 32, 32), (2, 192, 96, 4, 4), (1, 3, 5, 7, 9)])
def test_pool_convtranspose_residual(be, B, Ci, Co, H, W):
    res = K.check_pool_convt_resid(be, B, Ci, Co, H, W)
    assert res.pop("pool") == 0.0 and res.pop("pool_bwd") == 0.0  # selections, not arithmetic: exact
    _assert_all(res)


@pytest.mark.parametrize("mfma", [-1, 0])
@pytest.mark.parametrize("B,Ci,Co,H,W", [(128, 24, 12, 32, 32), (128, 48, 24, 16, 16), (128, 96, 48, 8, 8), (128, 192, 96, 4, 4), (3, 50, 26, 9, 24),
                                         (5, 7, 3, 6, 5)])
def test_convtranspose_full_size(be, B, Ci, Co, H, W, mfma):
    """ConvTranspose2d(2, 2) forward / input gradient / weight + bias gradient at the four up-path levels of the configs[2] U-Net
    (batch 128) and two odd shapes, on the matrix-pipe kernels (convt6.hip) and on the fp32 VALU kernels they replace."""
    with K.tuned(be, convt_mfma=mfma):
        _assert_all(K.check_convt(be, B, Ci, Co, H, W))


@pytest.mark.parametrize("knob", [-1, 0])
@pytest.mark.parametrize("B,Ci,Co,H,W", [(128, 12, 2, 64, 64), (32, 11, 64, 64, 64), (32, 64, 2, 64, 64), (3, 50, 26, 9, 24), (5, 7, 3, 6, 5)])
def test_conv1x1_both_routes(be, B, Ci, Co, H, W, knob):
    """1x1 convolutions (U-Net OutConv, ResNet res_conv shapes + two odd ones): the streamed matrix-pipe kernels of conv1.hip
    (the default since round 4) and the general kernels (conv1_mfma = 0) against the oracle."""
    with K.tuned(be, conv1_mfma=knob):
        _assert_all(K.check_conv2d(be, B, Ci, Co, H, W, 1))


def test_convtranspose_into_a_channel_slice(be):
    """the transposed convolution written into / its gradient read from the trailing channels of a wider tensor == the dense calls"""
    for B, Ci, Co, H, W in [(128, 24, 12, 32, 32), (128, 192, 96, 4, 4), (16, 48, 24, 16, 16), (3, 7, 5, 6, 12)]:
        assert K.check_convt_strided(be, B, Ci, Co, H, W) == 0, (B, Ci, Co, H, W)


def test_adam_for_many_tensors_in_one_launch(be):
    assert K.check_adam_multi(be, sizes=tuple([3, 700, 100000, 12] * 25)) < 2e-6


def test_adam_for_many_tensors_aligned_and_unaligned_in_one_launch(be):
    assert K.check_adam_multi(be, sizes=(7, 1025, 300, 1, 9, 64), shift_odd=True) < 2e-6


def test_adam_flat_on_unaligned_buffers(be):
    """the 16-byte-unit Adam kernel and its one-element-per-thread form (buffers off the 16-byte grid) make the same update, bit for bit"""
    assert K.check_adam_flat_unaligned(be) == 0
    assert K.check_adam_flat_unaligned(be, n=926446) == 0  # the FNO engine's flat buffer: one trip of 905 workgroups / two of 2048


def test_multi_tensor_launches_beyond_the_workgroup_cap(be):
    """a tensor of more than 2048 x 256 elements: several trips per thread in cfd_adam_multi / cfd_scale_copy_multi"""
    assert K.check_adam_multi(be, sizes=(600001, 5, 2048 * 256 + 1)) < 2e-6
    assert K.check_scale_copy_multi(be, sizes=(600001, 3, 2048 * 256 + 1), scale=0.5) == 0


def test_loss_scores_and_their_gradient(be):
    """(mse, rmse, mae, nmse) from the sums tensor and d(scores)/d(sums), one launch each"""
    assert K.check_loss_scores_bwd(be) < 1e-6


@pytest.mark.parametrize("step", [0, 7, 123456789])
def test_dropout_seed_formed_on_the_device(be, step):
    """the dropout stream's step counter read from device memory gives the masks of the host-side seed formula (models/resnet.py)"""
    assert K.check_dropout_step(be, 4 * 1031, 0.2, 0x1234567890ABCDEF, step) == 0


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_dropout_gelu_one_pass(be, p):
    """gelu(dropout(x)) and its gradient in one pass each == the stand-alone passes bit for bit (p = 0: the plain GELU)"""
    bad, err = K.check_dropout_gelu(be, 32 * 64 * 64 * 64, p)
    assert bad == 0 and err < 1e-12


@pytest.mark.parametrize("B,C,H,W", [(2, 96, 4, 4), (3, 12, 32, 32), (1, 5, 7, 9), (2, 3, 1, 1), (1, 2, 33, 2)])
def test_upsample_bilinear_align_corners(be, B, C, H, W):
    _assert_all(K.check_upsample_bilinear(be, B, C, H, W))


@pytest.mark.parametrize("S,shape,act", [(8000, (100,), "relu"), (8, (1000, 100), "tanh"), (37, (333,), "gelu"), (5, (40, 24), "swish")])
def test_normact(be, S, shape, act):
    _assert_all(K.check_normact(be, S, shape, act))


@pytest.mark.parametrize("B,Kq,P", [(8, 1000, 100), (3, 37, 24), (5, 300, 200), (20, 77, 129)])
def test_broadcast_add_and_rowdot(be, B, Kq, P):
    res = K.check_bcast_rowdot(be, B, Kq, P)
    assert res.pop("gbias") < 1e-5
    _assert_all(res)


@pytest.mark.parametrize("act", ["relu", "tanh", "gelu", "swish"])
def test_standalone_activation(be, act):
    _assert_all(K.check_act(be, 1000, act))


def test_conv_split_k_and_gather_paths_agree(be):
    """A deep, narrow layer (4x4 images, 96 -> 192 channels): with the workspace the LDS-tiled kernel runs split-K over
    the channel chunks, without it the gather kernel; both within tolerance of the oracle and of each other."""
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = np.random.default_rng(91)
    B, Ci, Co, H, W, ks = 128, 96, 192, 4, 4, 3
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, ks, ks)) * 0.05).astype(np.float32)
    b = rng.standard_normal(Co).astype(np.float32)
    dx, dw, db = be.dev(x), be.dev(w), be.dev(b)
    nws = api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks)
    assert nws > 0, "this shape is expected to take the split-K path"
    ws = be.bytes(nws)
    y_split, y_gather = be.zeros((B, Co, H, W)), be.zeros((B, Co, H, W))
    api.call("cfd_conv2d_fwd", P(dx), P(dw), P(db), P(y_split), P(ws), B, Ci, Co, H, W, ks, be.stream)
    api.call("cfd_conv2d_fwd", P(dx), P(dw), P(db), P(y_gather), None, B, Ci, Co, H, W, ks, be.stream)
    be.sync()
    ref = CO.conv2d(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    assert K.nm(be.host(y_split), ref) < K.TOL and K.nm(be.host(y_gather), ref) < K.TOL
    assert K.nm(be.host(y_split), be.host(y_gather).astype(np.float64)) < K.TOL


def test_backward_phase_argument_checks(be):
    from cfdbench_amd._capi import CfdError, FnoShape
    import ctypes
    plan = be.api.plan_create(64, 64, 12, 12)
    try:
        shape = FnoShape(2, 64, 64, 2, 2, 5, 8, 2, 12, 12, 128)
        for phase in (-1, 4):  # valid phases of a 2-layer model: 0 .. 3
            with pytest.raises(CfdError):
                be.api.call("cfd_fno_backward_phase", plan, ctypes.byref(shape), None, None, None, None, None, None, None,
                            None, None, None, phase, be.stream)
    finally:
        be.api.plan_destroy(plan)


# fp32-exact class: every contraction kernel with the activation operand in THREE bf16 pieces (act_pieces = 3, cfd_common.h)
EXACT_TOL = 2e-13


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
def test_three_piece_activations_spectral_and_block(be, H, W):
    """act_pieces = 3: six-MFMA products in the transforms, the fused FnoBlock / its two-pass form and the 1x1 kernels hold the
    fp64 oracle to fp32 round-off (the default two-piece route: ~2e-11 .. 5e-11)."""
    with K.tuned(be, act_pieces=3):
        _assert_all(K.check_spectral(be, 2, 20 if W == 64 else 5, 20 if W == 64 else 5, H, W), EXACT_TOL)
        _assert_all(K.check_block(be, 1, 20, 20, H, W), EXACT_TOL)
        _assert_all(K.check_chanmix(be, 2, 20, 20, H * W, 1), EXACT_TOL)
        _assert_all(K.check_idft_epilogues(be, 3, H, W), EXACT_TOL)
        # the projection head: inference kernel, stand-alone backward, one-pass training kernel (three-piece fc1 weights, input planes
        # and hidden-layer gradients; single-buffered planes)
        for res in (K.check_head(be, 2, 20, H * W, 1), K.check_head_train(be, 2, 20, H * W, 1)):
            _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")}, 5e-13)  # (the loss sums are fp32 sums: ~1e-7)


@pytest.mark.gpu
def test_three_piece_activations_whole_model(be):
    """The whole model on the act_pieces = 3 route: every contraction of the step in fp32-exact class."""
    with K.tuned(be, act_pieces=3):
        res = K.check_fno_vs_oracle(be, 2, 20, 2, 64, 64)
        assert res.pop("nmse_loss") < 1e-5  # a ratio of fp32 sums
        _assert_all(res, 3e-12)  # fp32 round-off through the whole network (the default route: 3e-12 predictions, <= 2e-11 gradients)


def test_two_piece_activation_route_still_holds_its_tolerance(be):
    """act_pieces = 2 (the default of rounds 1-4, now the selectable `split2` route of bench.py): activation operands in two bf16
    pieces, rel. 2^-16 per product -- every contraction kernel and the whole model inside 1e-9 of the fp64 oracle.  (Everything else
    in this file runs the round-5 default, three pieces.)"""
    with K.tuned(be, act_pieces=2):
        _assert_all(K.check_spectral(be, 2, 20, 20, 64, 64), 1e-9)
        _assert_all(K.check_block(be, 1, 20, 20, 64, 64), 1e-9)
        _assert_all(K.check_block(be, 1, 20, 20, 66, 65), 1e-9)
        _assert_all(K.check_chanmix(be, 2, 20, 20, 64 * 64, 1), 1e-9)
        for res in (K.check_head(be, 2, 20, 64 * 64, 1), K.check_head_train(be, 2, 20, 64 * 64, 1)):
            _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")}, 1e-9)
        res = K.check_fno_vs_oracle(be, 2, 20, 2, 64, 64)
        assert res.pop("nmse_loss") < 1e-5
        _assert_all(res, 1e-9)


@pytest.mark.parametrize("wide", [1, 2])
@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
@pytest.mark.parametrize("Ci,Co", [(32, 32), (28, 25), (32, 20), (20, 32)])
def test_fused_block_for_25_to_32_channels(be, H, W, Ci, Co, wide):
    """The fused FnoBlock kernel at the reference's default width 32 (src/args.py:190), both directions, both piece counts.  block_wide = 1
    (round 5): the destination channels of an entry dealt to two workgroups of 16 (k_block: ND = 2; B = 3 leaves padding workgroups in the
    grid of 8 * ND block groups).  block_wide = 2 (round 6): ONE (8,4,4) workgroup per entry with a single source-chunk buffer -- the
    default where it beats the two passes."""
    with K.tuned(be, block_wide=wide):
        _assert_all(K.check_block(be, 3, Ci, Co, H, W), EXACT_TOL)
        with K.tuned(be, act_pieces=2):
            _assert_all(K.check_block(be, 1, Ci, Co, H, W), 1e-9)
    _assert_all(K.check_block(be, 2, Ci, Co, H, W), EXACT_TOL)  # the default dispatch


@pytest.mark.parametrize("wide", [1, 2])
def test_fused_block_width_32_batch_split_is_bitwise(be, wide):
    """Width 32: a sample's result does not depend on the batch it sits in (entries split by row tiles AND, with block_wide = 1, by
    destination group) -- as long as the same kernel runs: the DEFAULT dispatch picks by batch size (block_fused_ok)."""
    with K.tuned(be, block_wide=wide):
        res = K.check_block_batch_split(be, 300, 2, 32, 64, 64)
    assert res["fwd_bitwise"] == 0.0 and res["bwd_bitwise"] == 0.0, res


@pytest.mark.parametrize("C,HW", [(20, 64 * 64), (20, 66 * 65), (32, 64 * 64), (5, 36 * 40)])
def test_training_head_on_eight_waves(be, C, HW):
    """Round 5: the one-pass training head with 16 hidden units per wave on eight waves (`head_waves` = 8; the partial d/dh sums of waves
    w and w + 4 meet in one zeroed LDS slot through two ds_add_f32): same results class as the four-wave kernel, both piece counts, and
    bit-stable from call to call (two contributions per slot: the order of the two adds cannot matter)."""
    with K.tuned(be, head_waves=8):
        res = K.check_head_train(be, 2, C, HW, 1)
        _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")}, 5e-13)
        with K.tuned(be, act_pieces=2):
            res = K.check_head_train(be, 3, C, HW, 1)
            _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")}, 1e-9)


@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
@pytest.mark.parametrize("B,C", [(2, 20), (3, 8), (1, 24)])
def test_lifting_layer_gradient_from_the_block_kernels_sums(be, B, C, H, W):
    """Round 5: the input-gradient kernel of FnoBlock 0 emits the six per-(entry, channel) sums the fc0 gradient needs instead of storing
    g_0 (k_block<.., STEMG> + k_stem_grad_combine) -- whole model against the fp64 oracle with the fusion (default) and without it
    (`stem_fuse` = 0: the stored g_0 and the k_chan_wgrad_stem pass), several batch / width combinations (row splits, both wave shapes)."""
    with K.tuned(be, stem_fuse=3 if W != 64 else -1):  # (on the general grids the sums are built but off by default: slower there)
        res = K.check_fno_vs_oracle(be, B, C, 2, H, W, border=B == 3)  # (B = 3: a mask with zero rows / column; 66 x 65: ragged tile, tail column)
    assert res.pop("nmse_loss") < 1e-5
    _assert_all(res, 3e-12)
    with K.tuned(be, stem_fuse=0):
        res0 = K.check_fno_vs_oracle(be, B, C, 2, H, W, border=B == 3)
        assert res0.pop("nmse_loss") < 1e-5
        _assert_all(res0, 3e-12)


def test_scale_copy_multi_packs_many_tensors_into_one_buffer(be):
    """cfd_scale_copy_multi: the data-parallel gradient pack (engine.FlatGradExchange) -- 5 tensors and 170 tensors (three launches of <= 80)."""
    assert K.check_scale_copy_multi(be) == 0
    assert K.check_scale_copy_multi(be, sizes=tuple(1 + (37 * i) % 500 for i in range(170)), scale=0.125) == 0


@pytest.mark.parametrize("kw", [dict(B=4, C=20, L=4, H=64, W=64), dict(B=3, C=8, L=1, H=64, W=64, which="mae"), dict(B=300, C=20, L=2, H=64, W=64, steps=1),
                                dict(B=3, C=20, L=2, H=64, W=64, which="mse", flags=6), dict(B=2, C=20, L=2, H=66, W=65), dict(B=2, C=32, L=2, H=64, W=64),
                                dict(B=5, C=20, L=2, H=64, W=64, flags=1), dict(B=5, C=20, L=2, H=64, W=64, flags=2), dict(B=5, C=20, L=2, H=64, W=64, flags=4)])
def test_fused_train_step_with_deferred_launches(be, kw):
    """Round 6 (CFD_TRAIN_DEFER_*): the nMSE normaliser applied by Adam, the head's reduction riding in backward phase 1's block kernel, the
    fc0 combine riding in Adam's launch -- same parameters after two steps as the step with those five launches (fp32 round-off), same
    predictions bit for bit, and the (rescaled) gradient holds the oracle.  Every flag alone and all together."""
    res = K.check_fno_train_step_deferred(be, **kw)
    assert res.pop("sums") < 1e-6 and res.pop("preds") == 0.0
    _assert_all(res, 1e-11)


@pytest.mark.parametrize("B,H,W", [(256, 64, 64), (130, 66, 65), (64, 66, 65)])
def test_width_32_block_default_dispatch_at_full_size(be, B, H, W):
    """Width 32 at the sizes where the default dispatch switches between the single-workgroup fused kernel and the two passes
    (66 x 65: fused from 128 entries, two passes below; 64 x 64: always fused): both sides hold the oracle."""
    _assert_all(K.check_block(be, B, 32, 32, H, W))


@pytest.mark.parametrize("B,C,L,p,border", [(2, 20, 2, 5, False), (70, 20, 4, 5, True), (256, 20, 1, 5, False), (5, 32, 2, 8, True), (9, 12, 1, 0, True)])
def test_lifting_layer_fused_into_the_first_transform_is_bitwise(be, B, C, L, p, border):
    """k_dft_fwd64_b3<.., STEM> (round 6; default below 128 entries at 64 x 64): the wave of image (b, c) builds a_0[b, c] from the entry's
    u / v / mask planes, tables and case parameters by k_stem_fwd4's fmaf chain, folds it into the transform and stores it -- the whole
    model's predictions and gradients equal the two-launch route bit for bit."""
    res = K.check_stem_dft_fusion(be, B, C, L, p, border)
    assert all(v == 0.0 for v in res.values()), res


@pytest.mark.parametrize("M,K_in,N,act,in_act", [(300, 40, 24, "relu", None), (150, 100, 100, "gelu", "relu"), (260, 36, 230, "tanh", "tanh"), (129, 16, 16, "none", None), (70, 200, 52, "swish", "gelu")])
def test_linear_on_three_piece_bf16_operands(be, M, K_in, N, act, in_act):
    """k_rowgemm6 (round 6): forward product with bias / activation / pre-activation copy and the input gradient with the previous layer's
    activation derivative in the epilogue, weights pre-split into fragments; row counts that are no multiples of 128, column counts with a
    ragged last tile and a second pass (N > 208), K with a ragged last slab.  fp64 layer to the kernel-test tolerance; 1e-12 of the fp32-MFMA
    kernel's result."""
    res = K.check_linear_rowgemm6(be, M, K_in, N, act, in_act)
    assert res.pop("y_vs_fp32_kernel") < 1e-12 and res.pop("gx_vs_fp32_kernel") < 1e-12
    _assert_all(res)


@pytest.mark.parametrize("M,K_in,N,act,in_act", [(8192, 200, 200, "relu", "relu"), (4100, 160, 176, "gelu", None), (5000, 512, 512, "tanh", "tanh"), (6000, 100, 100, "tanh", "tanh"), (4500, 100, 112, "gelu", "tanh")])
def test_tall_linear_products_run_on_the_three_piece_kernel_by_default(be, M, K_in, N, act, in_act):
    """The shapes of the Auto-FFN / DeepONet / Auto-DeepONet-CNN layers (>= 4096 rows) take k_rowgemm6 without any knob."""
    res = K.check_linear_rowgemm6(be, M, K_in, N, act, in_act, force=False)
    assert res.pop("y_vs_fp32_kernel") < 1e-12 and res.pop("gx_vs_fp32_kernel") < 1e-12
    _assert_all(res)


def test_tall_linear_layer_at_the_cnn_models_row_count(be):
    """131072 x 512 x 512 (the Auto-DeepONet-CNN's output FFN): the weight gradient's row chunks are planned WITHOUT the column of ones here
    (K % 128 == 0) -- the workspace was once sized for the plan with it (fewer chunks): a write past its end."""
    res = K.check_linear_rowgemm6(be, 131072, 512, 512, "tanh", "tanh", force=False)  # (smooth activations: at 67 M pre-activations a ReLU mask differs between the fp64 layer and any fp32 kernel in a few elements)
    assert res.pop("y_vs_fp32_kernel") < 1e-12 and res.pop("gx_vs_fp32_kernel") < 1e-12
    _assert_all(res)


@pytest.mark.parametrize("rows,cols,ldl", [(7, 130, 260), (33, 4290, 8580), (1, 50, 77), (300, 3, 5)])
def test_mse_loss_with_strided_label_rows(be, rows, cols, ldl):
    """ABI 601: the loss entry points on label rows at a stride (a channel slice of a (B, C, H, W) label viewed as (B, H W)) == the contiguous
    entry points on a copy, bit for bit; scores to fp32 round-off of the fp64 values."""
    res = K.check_mse_loss_strided_labels(be, rows, cols, ldl)
    assert res.pop("sums_differ") == 0.0 and res.pop("scores_differ") == 0.0 and res.pop("gp_differ") == 0.0
    assert res["scores_rel"] < 1e-5


@pytest.mark.parametrize("rows,ka,lda,kb,ldb", [(5, 130, 260, 5, 5), (33, 4290, 8580, 8, 8), (1, 7, 7, 3, 9)])
def test_rows_concat2(be, rows, ka, lda, kb, ldb):
    assert K.check_rows_concat2(be, rows, ka, lda, kb, ldb)["differs"] == 0.0
