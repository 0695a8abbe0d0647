"""CPU: every kernel of the product library, compiled UNMODIFIED against the SIMT emulator (tests/emul), checked
against the oracle.  Catches index-math / layout bugs before a GPU minute is spent; the GPU twin of this file is
tests/test_gpu_kernels.py."""
import numpy as np
import pytest

from tests import kernel_checks as K
from tests.backends import NumpyBackend


@pytest.fixture(scope="module")
def be():
    return NumpyBackend()


def _assert_all(res, tol=K.TOL):
    bad = {k: v for k, v in res.items() if not (v < tol)}
    assert not bad, f"parity failures (tol {tol}): {bad}; all: {res}"


def test_symbols_exported(be):
    assert be.api.missing == []
    assert be.api.version() >= 100


@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
def test_spectral_fwd_bwd(be, H, W):
    _assert_all(K.check_spectral(be, 2, 3, 5, H, W))


def test_spectral_fwd_bwd_fused_route(be):
    """C = 20 at 64 x 64: adjoint mix + weight gradient in one launch, their reduction riding in the inverse transform's."""
    _assert_all(K.check_spectral(be, 1, 20, 20, 64, 64))


@pytest.mark.parametrize("B,Cin,Cout", [(3, 20, 20), (9, 12, 7), (2, 32, 32), (37, 20, 20), (16, 8, 24)])
def test_mix_and_spectral_wgrad(be, B, Cin, Cout):
    _assert_all(K.check_mix_wgrad(be, B, Cin, Cout))


@pytest.mark.parametrize("want_wg,nwv,B,C", [("36", "2", 27, 20), ("256", "1", 3, 32)])
def test_mix_and_spectral_wgrad_multi_step(be, want_wg, nwv, B, C):
    """The batch-in-lanes kernels (k_mix_lds, k_spec_wgrad_tile, the fused k_mixadj_wgrad) at chunk sizes that reach
    the software-pipelined loops of the weight gradient (several 8-entry steps per workgroup, ragged last step) and
    several workgroup shapes of the mixing kernel, with a small batch."""
    with K.tuned(be, mix_nwv=int(nwv), wgrad_wg=int(want_wg), mode_mfma=0):
        _assert_all(K.check_mix_wgrad(be, B, C, C))


@pytest.mark.parametrize("B,bc,C", [(3, -1, 20), (37, -1, 20), (45, 16, 20), (21, 5, 20), (38, -1, 32), (9, 4, 32)])
def test_mix_and_spectral_wgrad_on_the_matrix_pipe(be, B, bc, C):
    """modes.hip (round 6): the three mode-domain contractions of 20 channels as real GEMMs on the fp32 matrix pipe; forced at small
    batches (mode_mfma = 1), mode_bc shrinks the chunk: several chunks, ragged last stage, a last K-step that is partly zeros."""
    with K.tuned(be, mode_mfma=1, mode_bc=bc):  # (32 channels: mix and adjoint only -- the weight gradient stays on the VALU kernels there)
        _assert_all(K.check_mix_wgrad(be, B, C, C))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 20, 20, 64, 64), (1, 6, 7, 32, 64), (1, 3, 5, 66, 65), (1, 32, 12, 32, 64), (5, 20, 20, 64, 64),
                                            (3, 24, 21, 64, 64), (2, 20, 20, 66, 65), (3, 12, 16, 48, 64),
                                            # round 4, the general fused kernel: row pitch != 64, tail columns (E = W - 64 = 1 .. 4), a ragged
                                            # last row tile, five row tiles; every (waves, channels) configuration
                                            (2, 6, 7, 50, 64), (1, 20, 20, 66, 67), (1, 5, 4, 70, 68), (2, 12, 20, 80, 65), (1, 16, 9, 33, 66)])
def test_fused_block(be, B, Cin, Cout, H, W):
    _assert_all(K.check_block(be, B, Cin, Cout, H, W))


def test_general_fused_block_splits_entries_unevenly(be):
    """Five row tiles at 66 x 65 do not divide: few batch entries are split over ceil(5 / tiles-per-workgroup) workgroups, and an
    entry's result must not depend on that (290 entries: one workgroup each; 2 entries: five workgroups each)."""
    with K.tuned(be, block_gen=1):
        res = K.check_block_batch_split(be, 290, 2, 3, 66, 65)
    assert res["fwd_bitwise"] == 0.0 and res["bwd_bitwise"] == 0.0, res


def test_general_fused_block_equals_its_two_passes(be):
    """block_gen = 0 routes the 66 x 65 FnoBlock through k_chanmix_b3 + k_idft_g; both routes hold the oracle (above) and differ from
    each other by rounding only."""
    with K.tuned(be, block_gen=0):
        _assert_all(K.check_block(be, 2, 20, 20, 66, 65))


def test_bf16_activation_storage_forward(be):
    """cfd_fno_forward_ex(act_dtype = bf16) == the oracle with the same storage rule (one rounding per stored activation)."""
    res = K.check_fno_bf16_storage(be, 1, 4, 1, 34, 33)
    info = {k: v for k, v in res.items() if k.startswith("info:")}
    assert res.pop("bf16_loss") < 1e-5  # fp32 loss sums
    _assert_all({k: v for k, v in res.items() if not k.startswith("info:")}, tol=1e-7)
    assert 1e-7 < info["info:bf16_vs_f32"] < 1e-3  # the storage format is visible, and small


@pytest.mark.parametrize("H,W", [(64, 64), (66, 65)])
def test_idft_epilogues(be, H, W):
    _assert_all(K.check_idft_epilogues(be, 3, H, W))


@pytest.mark.parametrize("Ci,Co,HW,act", [(20, 20, 256, True), (6, 8, 66 * 5, False), (32, 32, 64, True), (20, 20, 4096, True), (32, 32, 4290, False)])
def test_chanmix_and_wgrad(be, Ci, Co, HW, act):
    _assert_all(K.check_chanmix(be, 2, Ci, Co, HW, act))


@pytest.mark.parametrize("H,W,border", [(64, 64, False), (66, 65, True)])
def test_stem(be, H, W, border):
    _assert_all(K.check_stem(be, 2, H, W, 5, 20, border))


@pytest.mark.parametrize("C,HW,act,which,ext", [(20, 256, True, "nmse", False), (6, 330, False, "mse", True),
                                                (32, 128, True, "mae", False)])
def test_head(be, C, HW, act, which, ext):
    res = K.check_head(be, 2, C, HW, act, which, ext)
    _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")})
    assert res["sums"] < 1e-5 and res["scores"] < 1e-5


@pytest.mark.parametrize("C,HW,act,which,cap", [(20, 192, True, "nmse", 2), (6, 330, False, "mse", -1), (32, 128, True, "mae", -1)])
def test_head_train_one_pass(be, C, HW, act, which, cap):
    with K.tuned(be, head_blocks=cap):
        res = K.check_head_train(be, 2, C, HW, act, which)
    assert res.pop("sums") < 1e-5
    _assert_all(res)


def test_head_several_tiles_per_workgroup(be):
    """The head kernels' persistent loops (tile after tile per workgroup, staged planes / f'(a) handed from one tile to the
    next): at small sizes every workgroup would get a single tile, so the workgroup count is capped through the knob."""
    with K.tuned(be, head_blocks=2):
        res = K.check_head(be, 3, 20, 192, True, "nmse", False)
    _assert_all({k: v for k, v in res.items() if k not in ("sums", "scores")})


def test_loss_and_adam(be):
    res = K.check_loss_and_adam(be)
    assert res["sums"] < 1e-5
    assert res["adam_delta"] < 1e-9


@pytest.mark.parametrize("C,L,H,W,border", [(5, 1, 64, 64, False), (6, 2, 34, 33, True), (20, 4, 64, 64, False), (32, 2, 66, 65, True)])
def test_fno_whole_model(be, C, L, H, W, border):
    res = K.check_fno_vs_oracle(be, 1, C, L, H, W, border=border)
    loss_err = res.pop("nmse_loss")
    assert loss_err < 1e-5
    _assert_all(res, 1e-9)


def test_error_paths(be):
    from cfdbench_amd._capi import CfdError
    with pytest.raises(CfdError):
        be.api.plan_create(64, 64, 40, 12)  # modes1 too large
    plan = be.api.plan_create(64, 64, 12, 12)
    with pytest.raises(CfdError):
        be.api.call("cfd_spectral_dft", plan, None, None, 4, 0, None)
    x = np.zeros((1, 64, 64), np.float32)
    with pytest.raises(CfdError):
        be.api.call("cfd_spectral_idft", plan, x.ctypes.data, None, None, x.ctypes.data, 1, 1, None)  # epi without addend
    be.api.call("cfd_spectral_dft", plan, x.ctypes.data, x.ctypes.data, 0, 0, None)  # empty batch is a no-op
    be.api.plan_destroy(plan)


@pytest.mark.parametrize("M,N,Kd,ta,tb", [(70, 45, 37, 0, 0), (33, 100, 19, 0, 1), (100, 21, 66, 1, 0), (17, 18, 5, 1, 1), (20, 30, 700, 1, 0), (40, 24, 330, 0, 1)])
def test_gemm(be, M, N, Kd, ta, tb):
    _assert_all(K.check_gemm(be, M, N, Kd, ta, tb))


@pytest.mark.parametrize("tile", [64, 128, 1, 8])
@pytest.mark.parametrize("M,N,Kd,ta,tb", [(150, 45, 37, 0, 0), (33, 140, 19, 0, 1), (130, 21, 66, 1, 0), (17, 18, 5, 1, 1)])
def test_gemm_both_block_tiles(be, M, N, Kd, ta, tb, tile):
    """Every block tile of k_gemm (64 x 64, 128 x 128, 1 = 128 rows x all columns, 8 = 128 x 64 on eight waves) on every storage form, whatever launch_gemm would
    pick for the shape."""
    with K.tuned(be, gemm_tile=tile):
        _assert_all(K.check_gemm(be, M, N, Kd, ta, tb))


@pytest.mark.parametrize("M,K_in,N,in_act", [(37, 21, 24, "relu"), (70, 130, 9, "gelu"), (20, 7, 5, "tanh")])
def test_linear_input_gradient_leaves_as_the_previous_layers_dz(be, M, K_in, N, in_act):
    res = K.check_linear_chain_bwd(be, M, K_in, N, in_act)
    assert res.pop("differs_from_two_passes") == 0 and res.pop("gw_differs") == 0
    _assert_all(res)


@pytest.mark.parametrize("specs", [[(70, [5, 12, 20, 7], "relu", False, True), (33, [3, 17, 9], "gelu", True, False)], [(20, [4, 100, 100], "relu", False, True), (50, [100, 100, 100], "relu", False, True), (17, [2, 33, 100], "tanh", True, False)]])
def test_ffn_stacks_in_one_launch_equal_the_single_calls(be, specs):
    """cfd_ffn_stacks_fwd / _bwd (the branch and trunk stacks of a DeepONet variant as one launch per direction) == the single-stack
    calls, bit for bit."""
    res = K.check_ffn_stacks(be, specs)
    assert all(v == 0 for v in res.values()), {k: v for k, v in res.items() if v}


@pytest.mark.parametrize("M,K_in,N,act", [(137, 21, 24, "relu"), (20, 7, 5, "gelu")])
@pytest.mark.parametrize("tile", [128, 1])
def test_linear_act_on_the_large_tiles(be, M, K_in, N, act, tile):
    with K.tuned(be, gemm_tile=tile):
        _assert_all(K.check_linear(be, M, K_in, N, act))


@pytest.mark.parametrize("R,dims,act,act_last,with_gx", [(70, [5, 12, 20, 7], "relu", False, True), (33, [3, 17, 9], "gelu", True, False),
                                                          (300, [100, 100, 100, 100, 100], "relu", False, True), (130, [7, 128, 64, 100], "tanh", True, True)])
def test_ffn_stack(be, R, dims, act, act_last, with_gx):
    """A whole Linear(+activation) stack per kernel (cfd_ffn_stack_fwd / _bwd) against the fp64 layer-by-layer restatement."""
    _assert_all(K.check_ffn_stack(be, R, dims, act, act_last, with_gx))


@pytest.mark.parametrize("M,K_in,N,act", [(37, 21, 24, "relu"), (9, 130, 100, "tanh"), (20, 7, 5, "gelu"), (5, 3, 70, "swish"), (66, 2, 16, "none"), (24, 520, 20, "relu"),
                                          (2300, 3, 100, "relu"), (2050, 5, 130, "none")])  # M >= 2048: the two-stage column sum of the bias gradient
def test_linear_act(be, M, K_in, N, act):
    _assert_all(K.check_linear(be, M, K_in, N, act))


@pytest.mark.parametrize("B,P,Kq,HW,with_q", [(3, 24, 256, 256, False), (5, 100, 77, 300, True)])
def test_deeponet_inner(be, B, P, Kq, HW, with_q):
    res = K.check_deeponet_inner(be, B, P, Kq, HW, with_q)
    assert res.pop("gbias") < 1e-5  # a relative error of one fp32 sum, not an nMSE
    _assert_all(res)


# conv6.hip (k = 3 / 7) with two persistent workgroups (the emulator build's CFD_CONV6_GRID): several tiles, two channel chunks
# and ragged image groups per workgroup in the (40, 20, 18) case; more shapes in test_gpu_kernels.py
@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(1, 5, 7, 9, 10, 3), (40, 20, 18, 5, 4, 3), (2, 2, 3, 8, 9, 7), (2, 4, 4, 6, 6, 1), (9, 3, 35, 2, 9, 7), (2, 5, 20, 9, 10, 5), (3, 32, 7, 6, 5, 5),
                                            (3, 12, 12, 64, 64, 3), (2, 24, 12, 33, 32, 3), (20, 96, 40, 4, 4, 3), (2, 16, 64, 20, 21, 7),
                                            (9, 64, 2, 17, 16, 7)])
def test_conv2d_replicate(be, B, Ci, Co, H, W, ks):
    _assert_all(K.check_conv2d(be, B, Ci, Co, H, W, ks))


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(3, 12, 12, 64, 64, 3), (9, 5, 20, 9, 10, 3), (2, 8, 40, 20, 21, 3), (5, 12, 18, 16, 16, 3)])
def test_conv_emits_batchnorm_statistics(be, B, Ci, Co, H, W, ks):
    """conv forward with the statistics epilogue + the one-launch BatchNorm that consumes them (several tiles per workgroup)."""
    with K.tuned(be, conv6_grid=3):
        res = K.check_conv_bn_stats(be, B, Ci, Co, H, W, ks)
    assert res is not None
    _assert_all(res)


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(3, 12, 12, 64, 64, 3), (9, 5, 20, 9, 10, 3)])
def test_conv_batchnorm_statistics_with_mean_far_from_bias(be, B, Ci, Co, H, W, ks):
    """Output channels with |mean - bias| ~ 100 std: the statistics records carry a per-slot sample as their shift, so the variance
    does not come out of a cancelling difference (with the bias as the only shift y missed by ~3e-4 relative here)."""
    with K.tuned(be, conv6_grid=3):
        res = K.check_conv_bn_stats(be, B, Ci, Co, H, W, ks, offset=5.0, spread=0.1)
    assert res is not None
    # (y itself is bounded by the fp32 rounding of `out`, ~1e-7 of a mean that is 100 std: nMSE ~1e-8 whatever the statistics do)
    assert res["run_mean"] < 1e-10 and res["run_var"] < 1e-10 and res["out"] < 1e-10 and res["y"] < 1e-6, res


@pytest.mark.parametrize("B,Ci,Co,H,W,ks", [(2, 5, 8, 9, 10, 5), (3, 8, 7, 6, 5, 3), (2, 3, 8, 8, 9, 7), (20, 12, 12, 16, 16, 3)])
def test_conv2d_zero_padding(be, B, Ci, Co, H, W, ks):
    """(one channel chunk per pass: with the emulator build's two persistent workgroups a second chunk would be split over workgroups,
    which the zero-padding route does not take -- cfd_conv2d_zeropad_supported)"""
    _assert_all(K.check_conv2d_zeropad(be, B, Ci, Co, H, W, ks))


def test_conv_weights_prepared_in_one_batch(be):
    """fragments of several layers (k = 3, 5 and 7, narrow and wide, forward and input-gradient forms) made by one
    cfd_conv2d_wprep_batch launch: the layers then compute bit for bit what they compute preparing their own"""
    layers = [(2, 3, 12, 16, 16, 3), (3, 12, 24, 9, 10, 3), (2, 40, 20, 8, 8, 3), (1, 8, 16, 12, 13, 7), (2, 17, 5, 20, 21, 7), (2, 9, 32, 10, 11, 5)]
    assert K.check_conv_prepared(be, layers) == 0
    assert K.check_conv_prepared(be, [(1, 3 + i % 3, 4 + i % 5, 6, 6, 3) for i in range(25)], seed=38) == 0  # 50 items: two launches
    assert be.api.size("cfd_conv2d_wfrag_bytes", 12, 12, 9, 0) == 0 and be.api.size("cfd_conv2d_wfrag_bytes", 12, 2, 1, 1) == 0


@pytest.mark.parametrize("B,C,H,W,training,relu", [(2, 3, 6, 7, True, True), (2, 3, 4, 4, False, True), (3, 2, 5, 5, True, False), (4, 12, 64, 64, True, True),
                                                    (9, 48, 4, 4, True, True)])
def test_batchnorm_relu(be, B, C, H, W, training, relu):
    _assert_all(K.check_batchnorm(be, B, C, H, W, training, relu))


@pytest.mark.parametrize("B,Ci,Co,H,W", [(1, 3, 5, 6, 8), (2, 2, 3, 5, 5), (3, 24, 12, 16, 16), (2, 50, 7, 4, 4)])
def test_pool_convtranspose_residual(be, B, Ci, Co, H, W):
    res = K.check_pool_convt_resid(be, B, Ci, Co, H, W)
    assert res.pop("pool") == 0.0 and res.pop("pool_bwd") == 0.0  # selections, not arithmetic: exact
    _assert_all(res)


@pytest.mark.parametrize("B,C,H,W", [(1, 2, 5, 7), (1, 1, 1, 1), (2, 1, 4, 2)])
def test_upsample_bilinear_align_corners(be, B, C, H, W):
    _assert_all(K.check_upsample_bilinear(be, B, C, H, W))


@pytest.mark.parametrize("S,shape,act", [(5, (24,), "relu"), (2, (7, 20), "tanh"), (3, (33,), "gelu"), (2, (5, 6), "swish"),
                                         (2, (45, 100), "relu"), (1, (4099,), "tanh")])  # several 16-byte rounds per thread incl. the 4-deep loop; odd length
def test_normact(be, S, shape, act):
    _assert_all(K.check_normact(be, S, shape, act))


@pytest.mark.parametrize("B,Kq,P", [(2, 9, 20), (3, 5, 7), (2, 150, 130)])  # (eight-row trips of every row group; two passes of 128 columns)
def test_broadcast_add_and_rowdot(be, B, Kq, P):
    res = K.check_bcast_rowdot(be, B, Kq, P)
    assert res.pop("gbias") < 1e-5
    _assert_all(res)


@pytest.mark.parametrize("act", ["relu", "tanh", "gelu", "swish"])
def test_standalone_activation(be, act):
    _assert_all(K.check_act(be, 1000, act))


def test_conv2d_random_shapes(be):
    """Seeded sweep over 45 random (batch, channels, image, kernel size, persistent-grid) combinations of the k = 3 / 7 convolution
    kernels (conv6.hip), degenerate images included (1 x W, H x 1, smaller than the kernel, fewer images than a group of 8).
    Found in round 3: the k = 7 weight gradient on 4 x 4 tiles staged a halo larger than its per-thread item budget."""
    import random
    rnd = random.Random(20260927)
    shapes = [(8, 16, 40, 1, 4, 7, 1), (3, 24, 40, 2, 1, 7, 2), (2, 17, 16, 3, 1, 7, 8)]  # the three cases that exposed it
    while len(shapes) < 45:
        shapes.append((rnd.choice([1, 2, 3, 5, 8, 9, 17]), rnd.choice([1, 2, 5, 8, 9, 16, 17, 24, 33]), rnd.choice([1, 2, 7, 12, 16, 17, 31, 40]),
                       rnd.choice([1, 2, 3, 4, 5, 8, 13, 16, 33]), rnd.choice([1, 2, 3, 4, 7, 8, 9, 17, 32, 40]), rnd.choice([3, 3, 7]),
                       rnd.choice([-1, 1, 2, 3, 8])))
    for i, (B, Ci, Co, H, W, ks, grid) in enumerate(shapes):
        with K.tuned(be, conv6_grid=grid):
            res = K.check_conv2d(be, B, Ci, Co, H, W, ks, seed=100 + i)
        bad = {k: v for k, v in res.items() if not (v < 1e-10)}
        assert not bad, ((B, Ci, Co, H, W, ks, grid), bad)


def test_conv1x1_on_the_matrix_pipe(be):
    """1x1 convolutions (U-Net OutConv, ResNet res_conv) on the streamed kernels of conv1.hip (the default since round 4)
    and on the general kernels (conv1_mfma = 0): forward, input gradient, weight + bias gradient against the oracle -- pixel counts that are no
    multiple of a tile or a k-step, several row / column groups, several k-step ranges, H W % 8 != 0 (weight gradient falls back)."""
    import random
    rnd = random.Random(41)
    R = rnd.choice
    shapes = [(3, 12, 2, 8, 8), (2, 11, 64, 4, 8), (1, 64, 2, 16, 16), (2, 50, 50, 2, 8), (5, 3, 7, 3, 5), (4, 16, 100, 4, 4), (8, 12, 2, 16, 32)]
    while len(shapes) < 16:
        shapes.append((R([1, 2, 3, 5]), R([1, 2, 7, 16, 17, 33, 64, 100]), R([1, 2, 3, 16, 17, 49, 64]), R([1, 2, 3, 4, 8]), R([1, 3, 4, 8, 16])))
    for i, (B, Ci, Co, H, W) in enumerate(shapes):
        for knob in (-1, 0) if i < 7 else (-1,):
            with K.tuned(be, conv1_mfma=knob):
                res = K.check_conv2d(be, B, Ci, Co, H, W, 1, seed=300 + i)
            bad = {k: v for k, v in res.items() if not (v < 1e-10)}
            assert not bad, ((B, Ci, Co, H, W, knob), bad)


def test_convtranspose_random_shapes(be):
    """Seeded sweep over the ConvTranspose2d(2, 2) kernels of convt6.hip: pixel counts that are no multiple of a 16-pixel tile or a
    32-pixel k-step, several row / column groups, several k-step ranges per launch, widths the MFMA weight gradient does not take
    (W % 8 != 0: fp32 kernel) -- and every shape again on the fp32 VALU kernels (convt_mfma = 0)."""
    import random
    rnd = random.Random(31)
    R = rnd.choice
    shapes = [(3, 20, 9, 1, 8), (4, 20, 9, 16, 16), (2, 50, 26, 8, 8), (1, 100, 3, 2, 16), (2, 3, 50, 4, 8), (8, 24, 12, 16, 32)]
    while len(shapes) < 20:
        shapes.append((R([1, 2, 3, 5]), R([1, 2, 7, 16, 17, 24, 48, 49, 96]), R([1, 2, 3, 4, 5, 12, 13, 24, 48]), R([1, 2, 3, 4, 8, 9]),
                       R([1, 3, 4, 8, 8, 16, 24])))
    for i, (B, Ci, Co, H, W) in enumerate(shapes):
        for mfma in (-1, 0) if i < 8 else (-1,):
            with K.tuned(be, convt_mfma=mfma):
                res = K.check_convt(be, B, Ci, Co, H, W, seed=200 + i)
            bad = {k: v for k, v in res.items() if not (v < 1e-10)}
            assert not bad, ((B, Ci, Co, H, W, mfma), bad)


def test_convtranspose_into_a_channel_slice(be):
    """the transposed convolution written into / its gradient read from the trailing channels of a wider tensor == the dense calls"""
    for B, Ci, Co, H, W in [(2, 20, 9, 4, 8), (3, 50, 26, 8, 8), (1, 7, 3, 2, 4)]:
        assert K.check_convt_strided(be, B, Ci, Co, H, W) == 0, (B, Ci, Co, H, W)


def test_adam_for_many_tensors_in_one_launch(be):
    assert K.check_adam_multi(be, sizes=(7, 1025, 300, 1)) < 2e-6


def test_adam_for_many_tensors_aligned_and_unaligned_in_one_launch(be):
    assert K.check_adam_multi(be, sizes=(7, 1025, 300, 1, 9, 64), shift_odd=True) < 2e-6


def test_adam_flat_on_unaligned_buffers(be):
    """the 16-byte-unit Adam kernel and its one-element-per-thread form (buffers off the 16-byte grid) make the same update, bit for bit"""
    assert K.check_adam_flat_unaligned(be) == 0


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
    bad, err = K.check_dropout_gelu(be, 4 * 1031, p)
    assert bad == 0 and err < 1e-12


def test_dense_and_norm_kernels_random_shapes(be):
    """The same kind of sweep for BatchNorm, the fused Linear stacks, the GEMM and the 1x1 conv / its weight gradient."""
    import random
    rnd = random.Random(7)
    R = rnd.choice
    for it in range(10):
        B, C, H, W = R([1, 2, 3, 5, 9]), R([1, 2, 3, 7, 12, 17, 24, 33, 48]), R([1, 2, 3, 4, 6, 9, 16]), R([1, 2, 3, 4, 5, 8, 9, 16])
        _assert_all(K.check_batchnorm(be, B, C, H, W, R([True, False]), R([True, False]), seed=it))
        dims = [R([1, 2, 3, 7, 16, 33, 100, 128]) for _ in range(R([2, 3, 4, 6]))]
        _assert_all(K.check_ffn_stack(be, R([1, 5, 16, 17, 33, 70, 129]), dims, R(["relu", "gelu", "tanh"]), R([True, False]), R([True, False]), seed=it))
        _assert_all(K.check_gemm(be, R([1, 5, 16, 17, 40, 100]), R([1, 3, 16, 21, 64, 100]), R([1, 2, 5, 19, 37, 130]), R([0, 1]), R([0, 1]), seed=it))
        _assert_all(K.check_chanmix(be, B, R([1, 2, 6, 8, 9, 16, 20, 24, 32]), R([1, 3, 8, 9, 16, 20, 25, 32]), R([1, 2, 5, 63, 64, 65, 130, 330]),
                                    R([True, False]), seed=it))


def test_fno_kernels_random_shapes(be):
    """Seeded sweep over grids (16 .. 70 x 16 .. 80, the 64-wide and the general-width kernels), kept modes and channel counts for the
    transforms, the mode-domain kernels, the FnoBlock and the inverse-transform epilogues."""
    import random
    rnd = random.Random(11)
    R = rnd.choice
    for it in range(6):
        H, W = R([16, 24, 32, 33, 48, 64, 65, 66, 70]), R([16, 17, 32, 33, 48, 64, 65, 72, 80])
        m1, m2 = min(R([1, 2, 4, 7, 8, 12]), H // 2), min(R([1, 2, 5, 8, 12]), W // 2)
        B, Ci, Co = R([1, 2, 3, 5]), R([1, 3, 8, 12, 20, 21, 24, 32]), R([1, 2, 7, 16, 20, 25, 32])
        _assert_all(K.check_spectral(be, B, Ci, Co, H, W, m1, m2, seed=it), 1e-9)
        _assert_all(K.check_block(be, B, Ci, Co, H, W, m1, m2, seed=it), 1e-9)
        _assert_all(K.check_idft_epilogues(be, 3 * B, H, W, m1, m2, seed=it), 1e-9)
        _assert_all(K.check_mix_wgrad(be, R([1, 3, 9, 17, 30]), Ci, Co, m1, m2, H, W, seed=it), 1e-9)


# fp32-exact class: every contraction kernel with the activation operand in THREE bf16 pieces (act_pieces = 3, cfd_common.h)
EXACT_TOL = 2e-13


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
        res = K.check_block_batch_split(be, 5, 2, 32, 64, 64)
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


@pytest.mark.parametrize("kw", [dict(B=2, C=20, L=2, H=64, W=64), dict(B=1, C=8, L=1, H=64, W=64, which="mae"),
                                dict(B=2, C=20, L=1, H=64, W=64, which="mse", flags=6), dict(B=1, C=20, L=1, H=66, W=65)])
def test_fused_train_step_with_deferred_launches(be, kw):
    """Round 6 (CFD_TRAIN_DEFER_*): the nMSE normaliser applied by Adam, the head's reduction riding in backward phase 1's block kernel, the
    fc0 combine riding in Adam's launch -- same parameters after two steps as the step with those five launches (fp32 round-off), same
    predictions bit for bit, and the (rescaled) gradient holds the oracle."""
    res = K.check_fno_train_step_deferred(be, **kw)
    assert res.pop("sums") < 1e-6 and res.pop("preds") == 0.0
    _assert_all(res, 1e-11)


@pytest.mark.parametrize("B,C,L,p,border", [(2, 20, 2, 5, False), (3, 8, 1, 8, True), (1, 32, 1, 0, True)])
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
