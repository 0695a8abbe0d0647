"""MI355X: parity at the BASELINE.json sizes -- the drop-in models against fixtures made by the REAL reference modules
(tests/golden, oracle/make_golden_fullsize.py) and against the fp64 oracle on the same seeded inputs.

  configs[1]  Fno2d hidden 20 / L 4 / modes 12, B = 256, 64x64   whole forward + backward (every prediction and gradient
              against the fp64 oracle; the reference's own fp32 outputs as fingerprints)
  configs[2]  UNet dim 12, p = 8, 64x64, train-mode BatchNorm
  configs[3]  AutoDeepONet branch 4295 / width 100 / depth 8+8, 66x65
  configs[4]  Fno2d hidden 32 / L 4, 200-step rollout at 66x65 (eager generate_many and the one-graph FnoRollout)
  cylinder    Fno2d with 8 case parameters (13 input features)
"""
import numpy as np
import pytest

from oracle import fno_oracle as O
from oracle import synth

pytestmark = pytest.mark.gpu

NORTH_STAR_TOL = 1e-5  # BASELINE.json: outputs within 1e-5 relative nMSE of the reference (fp32)


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _fno(torch, params, C, L, p):
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn
    m = Fno2d(2, 2, p, loss_name_to_fn("nmse"), L, 12, 12, C).cuda()
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()})
    return m


def _cuda(torch, batch):
    return {k: torch.from_numpy(v).cuda() for k, v in batch.items()}


def _check_fingerprints(g, grads, tol_vals, tol_norm=2e-4):
    n = 0
    for key in g.files:
        if key.startswith("gsum::") and key.endswith("::vals"):
            k = key.split("::")[1]
            got = grads[k].reshape(-1)[g[f"gsum::{k}::idx"]]
            ref = g[key]
            if np.abs(ref).max() < 1e-7:  # conv bias in front of a train-mode BatchNorm: the exact gradient is zero
                assert np.abs(got).max() < 1e-6, k
            else:
                assert O.rel_nmse(got, ref) < tol_vals, (k, O.rel_nmse(got, ref))
                nrm = np.sqrt(np.sum(np.abs(grads[k]) ** 2))
                assert abs(nrm - abs(g[f"gsum::{k}::norm"])) <= tol_norm * nrm, k
            n += 1
    assert n > 0


def _named_grads(m):
    return {k: (p.grad.detach().cpu().numpy()) for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("name", ["fno_cfg2_b256", "fno_cyl_p8_64x64"])
def test_fno_whole_model_vs_reference_and_oracle(torch, golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    pseed, bseed, B, C, L, H, W, p = [int(v) for v in g["meta"]]
    params = synth.make_fno_params(pseed, C, L, 12, 12, p)
    batch = synth.make_batch(bseed, B, H, W, p)
    m = _fno(torch, params, C, L, p)
    out = m(**_cuda(torch, batch))
    out["loss"]["nmse"].backward()
    preds = out["preds"].detach().cpu().numpy()
    grads = _named_grads(m)
    # -- the reference's own fp32 results
    assert O.rel_nmse(preds[:2], g["preds_first"]) < 1e-9
    assert O.rel_nmse(preds.reshape(-1)[g["psum::idx"]], g["psum::vals"]) < 1e-9
    norms = np.sqrt((preds.astype(np.float64) ** 2).sum(axis=(1, 2, 3)))
    assert np.max(np.abs(norms - g["preds_sample_norms"]) / g["preds_sample_norms"]) < 1e-5  # every sample of the batch
    for k in ("mse", "rmse", "mae", "nmse"):
        assert abs(out["loss"][k].item() - float(g[f"loss_{k}"])) <= 5e-6 * abs(float(g[f"loss_{k}"]))
    fp_err = None
    try:
        _check_fingerprints(g, grads, tol_vals=1e-6)  # the reference's fp32 gradients carry ~1e-7 of their own round-off
    except AssertionError as e:  # report after the oracle comparison, which says whose round-off it is
        fp_err = e
    # -- the fp64 oracle on the WHOLE batch: every prediction and every gradient entry (~10 s of NumPy on the GPU box's host;
    # CFD_FAST_ORACLE=1 restricts it to the predictions of every 16th sample on slow hosts -- the forward pass is per sample)
    import os
    full = os.environ.get("CFD_FAST_ORACLE") != "1" or B <= 16
    sel = slice(None) if full else slice(0, None, 16)
    p64 = {k: v.astype(np.complex128 if np.iscomplexobj(v) else np.float64) for k, v in params.items()}
    b64 = {k: v.astype(np.float64)[sel] for k, v in batch.items()}
    ref = O.fno_forward(p64, b64["inputs"], b64["case_params"], b64["mask"], b64["label"], L, keep_cache=full)
    err = O.rel_nmse(preds[sel], ref["preds"])
    print(f"{name}: preds nMSE vs the fp64 oracle {err:.2e} ({'whole batch' if full else 'every 16th sample'})")
    assert err < 1e-9 and err < NORTH_STAR_TOL, err
    if full:
        rg = O.fno_backward(p64, ref["cache"], O.loss_grad_wrt_preds(ref["cache"]["preds"], ref["cache"]["label"], "nmse"), L)
        for k, gk in grads.items():
            e = O.rel_nmse(gk, rg[k])
            print(f"{name}: grad {k} nMSE vs the fp64 oracle {e:.2e}")
            assert e < 1e-8 and e < NORTH_STAR_TOL, (k, e)
    if fp_err is not None:
        raise fp_err


@pytest.mark.parametrize("fused_head", [False, True])
def test_engine_step_b256_matches_autograd_path(torch, golden_dir, fused_head):
    """The fused training engine (what bench.py times) at B = 256: same predictions, loss sums and gradients as the
    autograd drop-in that the test above pins to the reference.  With fused_head = False the engine runs the very kernels of
    the autograd path (bitwise equal); its default forms predictions, loss and the head's gradients in ONE pass of the head
    (cfd_fno_forward_train), which changes summation orders only: 1e-12 on the predictions, 1e-10 on every gradient."""
    from cfdbench_amd.engine import FnoTrainEngine
    g = np.load(golden_dir / "fno_cfg2_b256.npz")
    pseed, bseed, B, C, L, H, W, p = [int(v) for v in g["meta"]]
    params = synth.make_fno_params(pseed, C, L, 12, 12, p)
    b = _cuda(torch, synth.make_batch(bseed, B, H, W, p))
    m1 = _fno(torch, params, C, L, p)
    out = m1(**b)
    out["loss"]["nmse"].backward()
    m2 = _fno(torch, params, C, L, p)
    eng = FnoTrainEngine(m2, lr=1e-3, loss_name="nmse", fused_head=fused_head)
    eng.forward_backward(b["inputs"], b["label"], b["case_params"], b["mask"])
    torch.cuda.synchronize()
    sums = eng.sums.tolist()
    assert abs(sums[0] / sums[2] - float(g["loss_nmse"])) <= 5e-6 * float(g["loss_nmse"])
    assert sums[3] == B * 2 * H * W
    if not fused_head:
        assert torch.equal(eng.preds, out["preds"].detach())
    else:
        assert O.rel_nmse(eng.preds.cpu().numpy(), out["preds"].detach().cpu().numpy()) < 1e-12
    for p1, gv in zip(m1.abi_parameters(), eng.flat.grad_views):
        a = torch.view_as_real(p1.grad).reshape(-1) if p1.grad.is_complex() else p1.grad.reshape(-1)
        if not fused_head:
            assert torch.equal(a, gv), "engine and autograd paths run the same kernels in the same order"
        else:
            assert O.rel_nmse(gv.cpu().numpy(), a.cpu().numpy()) < 1e-10


def test_unet_dim12_p8_64x64_vs_reference_golden(torch, golden_dir):
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.unet import UNet
    g = np.load(golden_dir / "unet_dim12_p8_64x64.npz")
    seed, bseed, B, H, W, dim, p = [int(v) for v in g["meta"]]
    m = UNet(2, 2, loss_name_to_fn("nmse"), p, insert_case_params_at="input", bilinear=False, dim=dim).cuda()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert len(shapes) == int(g["n_keys"]) == 136  # the reference's 136-tensor state_dict
    assert sum(p_.numel() for p_ in m.parameters()) == 1095362
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed).items()}
    m.load_state_dict(sd)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, 0, :] = 0
    batch["mask"][:, :, :, 0] = 0
    b = _cuda(torch, batch)
    m.train()
    out = m(inputs=b["inputs"], case_params=b["case_params"], mask=b["mask"], label=b["label"])
    assert O.rel_nmse(out["preds"].detach().cpu().numpy(), g["preds_train"]) < 1e-9
    assert abs(out["loss"]["nmse"].item() - float(g["loss_nmse"])) <= 1e-5 * float(g["loss_nmse"])
    out["loss"]["nmse"].backward()
    # gradient fingerprints are the reference's fp64 backward; the reference's own fp32 backward is ref32_vs_64 (~2e-4) away
    # from them (ReLU kinks), which bounds what an fp32 implementation can be asked to reproduce
    noise = float(g["ref32_vs_64"])
    assert 1e-6 < noise < 1e-2
    _check_fingerprints(g, _named_grads(m), tol_vals=20 * noise, tol_norm=0.05)
    for k, v in m.state_dict().items():
        if "running" in k:
            assert O.rel_nmse(v.cpu().numpy(), g[f"after::{k}"]) < 1e-10, k
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        ev = m(inputs=b["inputs"], case_params=b["case_params"], mask=b["mask"])["preds"]
    assert O.rel_nmse(ev.cpu().numpy(), g["preds_eval"]) < 1e-9


def test_unet_dim12_p8_b128_configs2_batch_vs_reference_golden(torch, golden_dir):
    """configs[2] at its OWN per-GPU batch (B = 128; VERDICT r4 missing #5 / next #9), fingerprint fixture of the reference module:
    predictions, loss, running statistics, eval-mode predictions, and every parameter gradient against BOTH of the reference's
    backward passes -- fp64 (`gsum::`, the exact answer) and its native fp32 (`g32sum::`).  The reference's two sets are
    `ref32_vs_64` apart (ReLU / max-pool kinks under fp32 rounding); an fp32 implementation must sit within a few times that
    distance of BOTH, and a BatchNorm-backward bug of 1e-3 relative size would not (20x the fp64 distance was the only bound in
    round 4)."""
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.unet import UNet
    g = np.load(golden_dir / "unet_dim12_p8_b128.npz")
    seed, bseed, B, H, W, dim, p = [int(v) for v in g["meta"]]
    assert B == 128
    m = UNet(2, 2, loss_name_to_fn("nmse"), p, insert_case_params_at="input", bilinear=False, dim=dim).cuda()
    shapes = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(shapes, seed).items()}
    m.load_state_dict(sd)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, 0, :] = 0
    batch["mask"][:, :, :, 0] = 0
    b = _cuda(torch, batch)
    m.train()
    out = m(inputs=b["inputs"], case_params=b["case_params"], mask=b["mask"], label=b["label"])
    preds = out["preds"].detach().cpu().numpy()
    assert O.rel_nmse(preds.reshape(-1)[g["psum::idx"]], g["psum::vals"]) < 1e-9
    assert np.allclose(np.sqrt((preds.astype(np.float64) ** 2).sum(axis=(1, 2, 3))), g["preds_sample_norms"], rtol=1e-5)
    assert abs(out["loss"]["nmse"].item() - float(g["loss_nmse"])) <= 1e-5 * float(g["loss_nmse"])
    out["loss"]["nmse"].backward()
    noise = float(g["ref32_vs_64"])
    assert 1e-7 < noise < 1e-2
    grads = _named_grads(m)
    _check_fingerprints(g, grads, tol_vals=8 * noise, tol_norm=0.02)  # against the reference's fp64 backward
    worst = 0.0
    for key in g.files:  # against the reference's own fp32 backward
        if key.startswith("g32sum::") and key.endswith("::vals"):
            k = key.split("::")[1]
            ref = g[key]
            if np.abs(ref).max() < 1e-7:
                continue
            worst = max(worst, O.rel_nmse(grads[k].reshape(-1)[g[f"g32sum::{k}::idx"]], ref))
    assert worst < 8 * noise, (worst, noise)
    for k, v in m.state_dict().items():
        if "running" in k:
            assert O.rel_nmse(v.cpu().numpy(), g[f"after::{k}"]) < 1e-10, k
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        ev = m(inputs=b["inputs"], case_params=b["case_params"], mask=b["mask"])["preds"].cpu().numpy()
    assert O.rel_nmse(ev.reshape(-1)[g["esum::idx"]], g["esum::vals"]) < 1e-9


def test_auto_deeponet_b512_66x65_configs3_batch_vs_reference_golden(torch, golden_dir):
    """configs[3] at its OWN per-GPU batch (B = 512), fingerprint fixture of the reference module."""
    from cfdbench_amd.models.auto_deeponet import AutoDeepONet
    from cfdbench_amd.models.loss import loss_name_to_fn
    from oracle import deeponet_oracle as D
    g = np.load(golden_dir / "auto_deeponet_b512_66x65.npz")
    pseed, bseed, B, H, W, width, depth, p = [int(v) for v in g["meta"]]
    assert B == 512
    m = AutoDeepONet(H * W + p, 2, loss_name_to_fn("nmse"), branch_depth=depth, trunk_depth=depth, width=width, act_name="relu").cuda()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in D.make_params(pseed, H * W + p, width, depth, depth).items()})
    b = _cuda(torch, synth.make_smooth_batch(bseed, B, H, W, p))
    out = m(inputs=b["inputs"], case_params=b["case_params"], label=b["label"], mask=b["mask"])
    preds = out["preds"].detach().cpu().numpy()
    assert O.rel_nmse(preds.reshape(-1)[g["psum::idx"]], g["psum::vals"]) < 1e-9
    assert np.allclose(np.sqrt((preds.astype(np.float64) ** 2).reshape(B, -1).sum(axis=1)), g["preds_sample_norms"], rtol=1e-5)
    assert abs(out["loss"]["nmse"].item() - float(g["loss_nmse"])) <= 1e-5 * float(g["loss_nmse"])
    out["loss"]["nmse"].backward()
    _check_fingerprints(g, _named_grads(m), tol_vals=1e-6, tol_norm=1e-3)


def test_resnet_h16_d4_64x64_vs_reference_golden(torch, golden_dir):
    """ResNet at the size init_model builds (hidden 16, depth 4, 7x7 kernels, 64x64: SURVEY a-7, 4.37 GFLOP per frame forward):
    eval-mode predictions, loss, every gradient (fingerprints of the reference's backward) and a rollout."""
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.resnet import ResNet
    g = np.load(golden_dir / "resnet_h16_d4_64x64.npz")
    seed, bseed, B, H, W, hidden, depth, p, steps = [int(v) for v in g["meta"]]
    m = ResNet(2, 2, p, loss_name_to_fn("nmse"), hidden_chan=hidden, num_blocks=depth, kernel_size=7, padding=3).cuda()
    sd = {k[len("sd::"):]: torch.from_numpy(np.ascontiguousarray(g[k])) for k in g.files if k.startswith("sd::")}
    assert list(sd.keys()) == list(m.state_dict().keys())
    m.load_state_dict(sd)
    m.eval()
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, -1, :] = 0
    b = _cuda(torch, batch)
    x = b["inputs"].clone().requires_grad_(True)
    out = m(inputs=x, case_params=b["case_params"], mask=b["mask"], label=b["label"])
    assert O.rel_nmse(out["preds"].detach().cpu().numpy(), g["preds"]) < 1e-9
    assert abs(out["loss"]["nmse"].item() - float(g["loss_nmse"])) <= 1e-5 * float(g["loss_nmse"])
    out["loss"]["nmse"].backward()
    _check_fingerprints(g, _named_grads(m), tol_vals=1e-7, tol_norm=2e-4)
    assert O.rel_nmse(x.grad.cpu().numpy(), g["g_inputs"]) < 1e-7
    with torch.no_grad():
        frames = m.generate_many(b["inputs"][0], b["case_params"][0], steps, b["mask"][0])
    assert len(frames) == steps + 1  # resnet.py:229
    for t in range(steps + 1):
        assert O.rel_nmse(frames[t].cpu().numpy(), g["frames"][t]) < 1e-8


def test_auto_deeponet_66x65_w100_d8_vs_reference_golden(torch, golden_dir):
    from cfdbench_amd.models.auto_deeponet import AutoDeepONet
    from cfdbench_amd.models.loss import loss_name_to_fn
    from oracle import deeponet_oracle as D
    g = np.load(golden_dir / "auto_deeponet_66x65.npz")
    pseed, bseed, B, H, W, width, depth, p = [int(v) for v in g["meta"]]
    m = AutoDeepONet(H * W + p, 2, loss_name_to_fn("nmse"), branch_depth=depth, trunk_depth=depth, width=width,
                     act_name="relu").cuda()
    assert sum(p_.numel() for p_ in m.parameters()) == 571301
    m.load_state_dict({k: torch.from_numpy(v) for k, v in D.make_params(pseed, H * W + p, width, depth, depth).items()})
    b = _cuda(torch, synth.make_smooth_batch(bseed, B, H, W, p))
    out = m(inputs=b["inputs"], case_params=b["case_params"], label=b["label"], mask=b["mask"])
    assert O.rel_nmse(out["preds"].detach().cpu().numpy(), g["preds"]) < 1e-9
    assert abs(out["loss"]["nmse"].item() - float(g["loss_nmse"])) <= 1e-5 * float(g["loss_nmse"])
    out["loss"]["nmse"].backward()
    _check_fingerprints(g, _named_grads(m), tol_vals=1e-6, tol_norm=1e-3)
    m.eval()
    with torch.no_grad():
        frames = m.generate_many(b["inputs"][:2], b["case_params"][:2], b["mask"][:2], 3)
    for t in range(3):
        assert O.rel_nmse(frames[t].cpu().numpy(), g["frames"][t]) < 1e-8


def test_rollout_200_steps_through_the_spectral_branch_vs_reference_golden(torch, golden_dir):
    """The same 200-step horizon with the blocks' identity routed through SpectralConv2d (oracle/synth.py, route = "spectral"): all
    the energy passes DFT -> mode mixing -> inverse DFT in every layer of every step, so the rounding of the transforms' FIXED
    operands (the twiddle tables, two bf16 pieces each on the default route) sits on the identity path and any coherent bias
    it had would accumulate over the 800 transforms.  Checked on the default (split-bf16) route and on
    the exact-fp32 route against the reference's fp32 CPU rollout."""
    from tests import kernel_checks as K
    from tests.backends import TorchBackend
    g = np.load(golden_dir / "rollout200_spectral_c32_66x65.npz")
    pseed, bseed, B, C, L, H, W, p, steps = [int(v) for v in g["meta"]]
    eps, gain, decay = [float(v) for v in g["hyper"]]
    assert str(g["route"]) == "spectral"
    params, batch = synth.make_rollout_case(pseed, bseed, B, C, L, H, W, p, eps, gain, decay, route="spectral")
    m = _fno(torch, params, C, L, p).eval()
    b = _cuda(torch, batch)
    keep = [int(k) for k in g["keep"]]
    out = {}
    for route, knob in (("default (bf16x3 transforms)", -1), ("exact_fp32", 1)):
        with K.tuned(TorchBackend(), exact_fp32=knob):
            with torch.no_grad():
                frames = m.generate_many(b["inputs"], b["case_params"], b["mask"], steps)
        errs = {k: O.rel_nmse(frames[k].cpu().numpy(), g["frames"][i]) for i, k in enumerate(keep)}
        print(f"spectral-branch rollout, {route}: nMSE vs the reference's fp32 frames:", {k: f"{v:.1e}" for k, v in errs.items()})
        out[route] = errs
    for route, errs in out.items():
        for k, e in errs.items():
            assert e < (2e-7 if route == "exact_fp32" else NORTH_STAR_TOL), (route, k, e)
    # the frames really went through the spectral branch: with the spectral weights zeroed the rollout dies away
    dead = {k: (np.zeros_like(v) if "conv0.weights" in k else v) for k, v in params.items()}
    md = _fno(torch, dead, C, L, p).eval()
    with torch.no_grad():
        fd = md.generate_many(b["inputs"], b["case_params"], b["mask"], 3)
    assert float(fd[2].double().pow(2).mean().sqrt()) < 0.05 * float(g["norms"][2])


def test_rollout_200_steps_c32_66x65_vs_reference_golden(torch, golden_dir):
    """BASELINE configs[4] horizon: 200 autoregressive steps on the tube / dam grid at the reference's default width.  The
    fixture's network is a near-identity propagator (oracle/synth.py), so round-off is carried from step to step instead
    of being contracted away; the error against the reference's fp32 CPU rollout must stay far inside the 1e-5 budget at
    EVERY stored step, and the one-graph FnoRollout must reproduce the eager loop bit for bit."""
    from cfdbench_amd.rollout import FnoRollout
    g = np.load(golden_dir / "rollout200_c32_66x65.npz")
    pseed, bseed, B, C, L, H, W, p, steps = [int(v) for v in g["meta"]]
    eps, gain, decay = [float(v) for v in g["hyper"]]
    params, batch = synth.make_rollout_case(pseed, bseed, B, C, L, H, W, p, eps, gain, decay)
    m = _fno(torch, params, C, L, p).eval()
    b = _cuda(torch, batch)
    with torch.no_grad():
        frames = m.generate_many(b["inputs"], b["case_params"], b["mask"], steps)
    assert len(frames) == steps
    keep = [int(k) for k in g["keep"]]
    errs = {k: O.rel_nmse(frames[k].cpu().numpy(), g["frames"][i]) for i, k in enumerate(keep)}
    print("rollout nMSE vs the reference's fp32 frames:", {k: f"{v:.1e}" for k, v in errs.items()})
    for k, e in errs.items():
        assert e < 2e-7 and e < NORTH_STAR_TOL, (k, e)
    norms = torch.stack([f.double().pow(2).mean().sqrt() for f in frames]).cpu().numpy()
    assert np.max(np.abs(norms - g["norms"]) / g["norms"]) < 1e-4  # all 200 steps (an nMSE of 5e-9 allows 7e-5 on a norm)
    graph_frames = FnoRollout(m).generate_many(b["inputs"], b["case_params"], b["mask"], steps)
    for k in (0, 99, 199):
        assert torch.equal(graph_frames[k], frames[k])
    # bf16 activation storage (BASELINE configs[4]).  The reference has no reduced-precision path, so the yardstick is the fp32
    # rollout above; the stated per-step tolerance is  nMSE_k(bf16 vs fp32) <= 4e-5 * k^1.5  (k = step, averaged over the
    # cases; measured 1.3e-5 at k = 1 and 4.1e-2 at k = 200 on this near-identity propagator, growth ~ k^1.44:
    # profiles/r02g_rollout_bf16_study.txt) -- i.e. bf16 storage is OUTSIDE the 1e-5 budget of the fp32 path from the first
    # step on and is offered as an explicit opt-in (FnoRollout(dtype="bf16"), test_multistep --dtype bf16).
    b16 = FnoRollout(m, dtype="bf16").generate_many(b["inputs"], b["case_params"], b["mask"], steps)
    errs16 = {k: O.rel_nmse(b16[k - 1].cpu().numpy(), frames[k - 1].cpu().numpy()) for k in (1, 2, 10, 50, 100, 200)}
    print("bf16-storage rollout nMSE vs the fp32 rollout:", {k: f"{v:.1e}" for k, v in errs16.items()})
    for k, e in errs16.items():
        assert e <= 4e-5 * k ** 1.5, (k, e)
    assert errs16[1] > 1e-7  # the storage format is really in use
    # the same storage rule in the oracle: one rounding per stored activation (first step, where nothing has accumulated)
    p64 = {k: v.astype(np.complex128 if np.iscomplexobj(v) else np.float64) for k, v in params.items()}
    ref16 = O.fno_forward(p64, batch["inputs"].astype(np.float64), batch["case_params"].astype(np.float64),
                          batch["mask"].astype(np.float64), None, L, act_store=O.bf16_round, keep_cache=False)["preds"]
    assert O.rel_nmse(b16[0].cpu().numpy(), ref16) < 1e-7
