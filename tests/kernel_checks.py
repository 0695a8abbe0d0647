"""Backend-agnostic parity checks of every C-ABI entry point against the oracle (oracle/fno_oracle.py).
Used by tests/test_emul_kernels.py (CPU, SIMT emulator) and tests/test_gpu_kernels.py (MI355X)."""
from __future__ import annotations

import contextlib
import ctypes

import numpy as np

from cfdbench_amd._capi import FnoParams, FnoShape
from oracle import fno_oracle as O
from oracle import synth

# north_star tolerance: relative nMSE <= 1e-5 (fp32).  The kernels are exact-fp32 MFMA/FMA chains, so the
# checks hold them to a far tighter bound; TOL is what a fp32 pipeline of this depth can actually deliver.
TOL = 1e-10
NORTH_STAR_TOL = 1e-5

f64 = np.float64
c128 = np.complex128


def nm(a, ref):
    return O.rel_nmse(a, ref)


@contextlib.contextmanager
def tuned(be, **knobs):
    """Dispatch overrides (cfd_tune_set) for the duration of a check; -1 restores the built-in choice."""
    for k, v in knobs.items():
        be.api.call("cfd_tune_set", k.encode(), int(v))
    try:
        yield
    finally:
        for k in knobs:
            be.api.call("cfd_tune_set", k.encode(), -1)


def check_spectral(be, B, Cin, Cout, H, W, m1=12, m2=12, seed=0):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    gy = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    w1 = (rng.random((Cin, Cout, m1, m2)) + 1j * rng.random((Cin, Cout, m1, m2))).astype(np.complex64)
    w2 = (rng.random((Cin, Cout, m1, m2)) + 1j * rng.random((Cin, Cout, m1, m2))).astype(np.complex64)
    plan = api.plan_create(H, W, m1, m2)
    try:
        dx, dgy, dw1, dw2 = be.dev(x), be.dev(gy), be.dev(w1), be.dev(w2)
        xh = be.zeros((B, Cin, 2 * m1, m2), np.complex64)
        z = be.zeros((B, Cout, 2 * m1, m2), np.complex64)
        y = be.zeros((B, Cout, H, W))
        api.call("cfd_spectral_conv2d_fwd", plan, P(dx), P(dw1), P(dw2), P(y), P(xh), P(z), B, Cin, Cout, be.stream)
        be.sync()
        x64, w164, w264 = x.astype(f64), w1.astype(c128), w2.astype(c128)
        res = {}
        res["xh"] = nm(be.host(xh), O.pruned_dft_fwd(x64, m1, m2))
        res["y"] = nm(be.host(y), O.spectral_conv2d_fwd(x64, w164, w264))
        ws = be.bytes(api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, Cin, Cout))
        gx = be.zeros((B, Cin, H, W))
        gw1 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        gw2 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        api.call("cfd_spectral_conv2d_bwd", plan, P(dgy), P(xh), P(dw1), P(dw2), P(gx), P(gw1), P(gw2), P(ws), B, Cin,
                 Cout, be.stream)
        be.sync()
        rgx, rgw1, rgw2 = O.spectral_conv2d_bwd(gy.astype(f64), x64, w164, w264)
        res["gx"] = nm(be.host(gx), rgx)
        res["gw1"] = nm(be.host(gw1), rgw1)
        res["gw2"] = nm(be.host(gw2), rgw2)
        return res
    finally:
        api.plan_destroy(plan)


def check_mix_wgrad(be, B, Cin, Cout, m1=12, m2=12, H=64, W=64, seed=11):
    """cfd_spectral_mix (forward + adjoint) and cfd_spectral_wgrad on their own: reaches every (channels, waves)
    instantiation cheaply (no transforms involved)."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    M2 = (2 * m1, m2)

    def crand(*shape):
        return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)

    xh, gh = crand(B, Cin, *M2), crand(B, Cout, *M2)
    w1, w2 = crand(Cin, Cout, m1, m2), crand(Cin, Cout, m1, m2)
    plan = api.plan_create(H, W, m1, m2)
    try:
        dxh, dgh, dw1, dw2 = be.dev(xh), be.dev(gh), be.dev(w1), be.dev(w2)
        z = be.zeros((B, Cout, *M2), np.complex64)
        api.call("cfd_spectral_mix", plan, P(dxh), P(dw1), P(dw2), P(z), B, Cin, Cout, 0, be.stream)
        gz = be.zeros((B, Cin, *M2), np.complex64)
        api.call("cfd_spectral_mix", plan, P(dgh), P(dw1), P(dw2), P(gz), B, Cin, Cout, 1, be.stream)
        ws = be.bytes(api.size("cfd_spectral_wgrad_workspace_bytes", plan, B, Cin, Cout))
        gw1 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        gw2 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        api.call("cfd_spectral_wgrad", plan, P(dxh), P(dgh), P(gw1), P(gw2), P(ws), B, Cin, Cout, be.stream)
        # the two gradient-mode consumers in one call (one launch where the fused kernel applies)
        fgz = be.zeros((B, Cin, *M2), np.complex64)
        fgw1 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        fgw2 = be.zeros((Cin, Cout, m1, m2), np.complex64)
        api.call("cfd_spectral_mix_adj_wgrad", plan, P(dxh), P(dgh), P(dw1), P(dw2), P(fgz), P(fgw1), P(fgw2), P(ws), B,
                 Cin, Cout, be.stream)
        be.sync()
        X, G = xh.astype(c128), gh.astype(c128)
        Wf = np.concatenate([w1, w2], axis=2).astype(c128)  # (Cin, Cout, 2*m1, m2): rows [0,m1) use w1, the rest w2
        cl = O.hermitian_weights(m2, W) / (H * W)
        rgw = np.einsum("bikl,bokl->iokl", X.conj(), G) * cl[None, None, None, :]
        rgz = np.einsum("bokl,iokl->bikl", G, Wf.conj())
        return {
            "mix": nm(be.host(z), np.einsum("bikl,iokl->bokl", X, Wf)),
            "mix_adj": nm(be.host(gz), rgz),
            "gw1": nm(be.host(gw1), rgw[:, :, :m1]),
            "gw2": nm(be.host(gw2), rgw[:, :, m1:]),
            "fused_mix_adj": nm(be.host(fgz), rgz),
            "fused_gw1": nm(be.host(fgw1), rgw[:, :, :m1]),
            "fused_gw2": nm(be.host(fgw2), rgw[:, :, m1:]),
        }
    finally:
        api.plan_destroy(plan)


def check_block(be, B, Cin, Cout, H, W, m1=12, m2=12, seed=12):
    """cfd_fno_block_fwd / cfd_fno_block_bwd_input (1x1 conv + inverse transform [+ gelu'] in one pass)."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    g = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    z = (rng.standard_normal((B, Cout, 2 * m1, m2)) + 1j * rng.standard_normal((B, Cout, 2 * m1, m2))).astype(np.complex64)
    gz = (rng.standard_normal((B, Cin, 2 * m1, m2)) + 1j * rng.standard_normal((B, Cin, 2 * m1, m2))).astype(np.complex64)
    w0 = rng.standard_normal((Cout, Cin)).astype(np.float32)
    b0 = rng.standard_normal((Cout,)).astype(np.float32)
    plan = api.plan_create(H, W, m1, m2)
    try:
        da, dg, dz, dgz, dw, db = be.dev(a), be.dev(g), be.dev(z), be.dev(gz), be.dev(w0), be.dev(b0)
        res = {}
        a64, w64 = a.astype(f64), w0.astype(f64)
        for act in (0, 1):
            out = be.zeros((B, Cout, H, W))
            api.call("cfd_fno_block_fwd", plan, P(da), P(dz), P(dw), P(db), P(out), B, Cin, Cout, act, be.stream)
            be.sync()
            fa = O.gelu(a64) if act else a64
            ref = np.einsum("oi,bixy->boxy", w64, fa) + b0[None, :, None, None] + O.pruned_idft(z.astype(c128), H, W)
            res[f"fwd_act{act}"] = nm(be.host(out), ref)
        refg = np.einsum("oi,boxy->bixy", w64, g.astype(f64)) + O.pruned_idft(gz.astype(c128), H, W)
        gin = be.zeros((B, Cin, H, W))
        api.call("cfd_fno_block_bwd_input", plan, P(dg), P(dgz), P(dw), None, P(gin), B, Cin, Cout, be.stream)
        be.sync()
        res["bwd"] = nm(be.host(gin), refg)
        gin2 = be.zeros((B, Cin, H, W))
        api.call("cfd_fno_block_bwd_input", plan, P(dg), P(dgz), P(dw), P(da), P(gin2), B, Cin, Cout, be.stream)
        be.sync()
        res["bwd_dgelu"] = nm(be.host(gin2), refg * O.gelu_grad(a64))
        return res
    finally:
        api.plan_destroy(plan)


def check_block_batch_split(be, Bbig, Bsmall, C, H, W, m1=12, m2=12, seed=13):
    """The fused FnoBlock kernel splits batch entries over several workgroups (by row tiles) when there are fewer entries than
    CUs: the first Bsmall entries computed alone (split) must equal, bit for bit, the same entries inside a batch of Bbig (one
    workgroup per entry) -- forward with GELU on load and the gelu' input gradient."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((Bbig, C, H, W)).astype(np.float32)
    g = rng.standard_normal((Bbig, C, H, W)).astype(np.float32)
    z = (rng.standard_normal((Bbig, C, 2 * m1, m2)) + 1j * rng.standard_normal((Bbig, C, 2 * m1, m2))).astype(np.complex64)
    w0 = rng.standard_normal((C, C)).astype(np.float32)
    b0 = rng.standard_normal((C,)).astype(np.float32)
    plan = api.plan_create(H, W, m1, m2)
    try:
        da, dg, dz, dw, db = be.dev(a), be.dev(g), be.dev(z), be.dev(w0), be.dev(b0)
        res = {}
        outs = []
        for B in (Bbig, Bsmall):
            out, gin = be.zeros((B, C, H, W)), be.zeros((B, C, H, W))
            api.call("cfd_fno_block_fwd", plan, P(da), P(dz), P(dw), P(db), P(out), B, C, C, 1, be.stream)
            api.call("cfd_fno_block_bwd_input", plan, P(dg), P(dz), P(dw), P(da), P(gin), B, C, C, be.stream)
            be.sync()
            outs.append((be.host(out), be.host(gin)))
        res["fwd_bitwise"] = float(np.max(np.abs(outs[0][0][:Bsmall] - outs[1][0])))
        res["bwd_bitwise"] = float(np.max(np.abs(outs[0][1][:Bsmall] - outs[1][1])))
        return res
    finally:
        api.plan_destroy(plan)


def check_idft_epilogues(be, nimg, H, W, m1=12, m2=12, seed=1):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    z = (rng.standard_normal((nimg, 1, 2 * m1, m2)) + 1j * rng.standard_normal((nimg, 1, 2 * m1, m2))).astype(np.complex64)
    ad = rng.standard_normal((nimg, 1, H, W)).astype(np.float32)
    ap = rng.standard_normal((nimg, 1, H, W)).astype(np.float32)
    plan = api.plan_create(H, W, m1, m2)
    try:
        ref = O.pruned_idft(z.astype(c128), H, W)
        dz, dap = be.dev(z), be.dev(ap)
        res = {}
        o1 = be.dev(ad)
        api.call("cfd_spectral_idft", plan, P(dz), P(o1), None, P(o1), nimg, 1, be.stream)
        be.sync()
        res["epi1"] = nm(be.host(o1), ref + ad)
        o2 = be.dev(ad)
        api.call("cfd_spectral_idft", plan, P(dz), P(o2), P(dap), P(o2), nimg, 2, be.stream)
        be.sync()
        res["epi2"] = nm(be.host(o2), (ref + ad) * O.gelu_grad(ap.astype(f64)))
        # GELU-on-load variant of the forward transform
        xh = be.zeros((nimg, 1, 2 * m1, m2), np.complex64)
        api.call("cfd_spectral_dft", plan, P(dap), P(xh), nimg, 1, be.stream)
        be.sync()
        res["dft_gelu"] = nm(be.host(xh), O.pruned_dft_fwd(O.gelu(ap.astype(f64)), m1, m2))
        return res
    finally:
        api.plan_destroy(plan)


def check_chanmix(be, B, Ci, Co, HW, act, seed=2):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Ci, HW)).astype(np.float32)
    w = rng.standard_normal((Co, Ci)).astype(np.float32)
    b = rng.standard_normal((Co,)).astype(np.float32)
    g = rng.standard_normal((B, Co, HW)).astype(np.float32)
    f = O.gelu(x.astype(f64)) if act else x.astype(f64)
    res = {}
    out = be.zeros((B, Co, HW))
    dx, dw, db, dg = be.dev(x), be.dev(w), be.dev(b), be.dev(g)  # keep alive across the calls
    api.call("cfd_chanmix", P(dx), P(dw), P(db), P(out), B, Ci, Co, HW, int(act), 0, be.stream)
    be.sync()
    res["fwd"] = nm(be.host(out), np.einsum("oi,bip->bop", w.astype(f64), f) + b[None, :, None])
    gin = be.zeros((B, Ci, HW))
    api.call("cfd_chanmix", P(dg), P(dw), None, P(gin), B, Co, Ci, HW, 0, 1, be.stream)
    be.sync()
    res["bwd_in"] = nm(be.host(gin), np.einsum("oi,bop->bip", w.astype(f64), g.astype(f64)))
    ws = be.bytes(api.size("cfd_chan_wgrad_workspace_bytes", B, Ci, Co, HW))
    gw, gb = be.zeros((Co, Ci)), be.zeros((Co,))
    api.call("cfd_chan_wgrad", P(dg), P(dx), P(gw), P(gb), P(ws), B, Ci, Co, HW, int(act), be.stream)
    be.sync()
    res["gw"] = nm(be.host(gw), np.einsum("bop,bip->oi", g.astype(f64), f))
    res["gb"] = nm(be.host(gb), g.astype(f64).sum(axis=(0, 2)))
    return res


def check_stem(be, B, H, W, P_, C, border, seed=3):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    batch = synth.make_batch(seed, B, H, W, P_, border_mask=border)
    F = 2 + 3 + P_
    w = rng.standard_normal((C, F)).astype(np.float32)
    b = rng.standard_normal((C,)).astype(np.float32)
    g = rng.standard_normal((B, C, H, W)).astype(np.float32)
    plan = api.plan_create(H, W, 12, 12)
    try:
        feats = O.assemble_features(batch["inputs"].astype(f64), batch["case_params"].astype(f64), batch["mask"].astype(f64))
        res = {}
        out = be.zeros((B, C, H, W))
        di, dm, dc = be.dev(batch["inputs"]), be.dev(batch["mask"]), be.dev(batch["case_params"])
        dw, db, dg = be.dev(w), be.dev(b), be.dev(g)
        api.call("cfd_fno_stem_fwd", plan, P(di), P(dm), P(dc), P(dw), P(db), P(out), B, 2, P_, C, be.stream)
        be.sync()
        res["fwd"] = nm(be.host(out), O.conv1x1(feats, w.astype(f64), b.astype(f64)))
        out2 = be.zeros((B, C, H, W))
        api.call("cfd_fno_stem_fwd", plan, P(di), None, P(dc), P(dw), P(db), P(out2), B, 2, P_, C, be.stream)
        be.sync()
        feats1 = O.assemble_features(batch["inputs"].astype(f64), batch["case_params"].astype(f64), np.ones_like(batch["mask"], dtype=f64))
        res["fwd_nomask"] = nm(be.host(out2), O.conv1x1(feats1, w.astype(f64), b.astype(f64)))
        ws = be.bytes(api.size("cfd_fno_stem_bwd_workspace_bytes", plan, B, 2, P_, C))
        gw, gb = be.zeros((C, F)), be.zeros((C,))
        api.call("cfd_fno_stem_bwd", plan, P(dg), P(di), P(dm), P(dc), P(gw), P(gb), P(ws), B, 2, P_, C, be.stream)
        be.sync()
        res["gw"] = nm(be.host(gw), np.einsum("bohw,bihw->oi", g.astype(f64), feats))
        res["gb"] = nm(be.host(gb), g.astype(f64).sum(axis=(0, 2, 3)))
        return res
    finally:
        api.plan_destroy(plan)


def _head_ref(a, mask, label, w1, b1, w2, b2, act):
    h = O.gelu(a) if act else a
    z1 = np.einsum("ji,bip->bjp", w1, h) + b1[None, :, None]
    a1 = O.gelu(z1)
    raw = np.einsum("cj,bjp->bcp", w2, a1) + b2[None, :, None]
    preds = raw * mask
    return h, z1, a1, preds


def check_head(be, B, C, HW, act, which="nmse", with_ext=False, border=True, seed=4):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    Hd, Co = 128, 2
    a = rng.standard_normal((B, C, HW)).astype(np.float32)
    mask = np.ones((B, 1, HW), np.float32)
    if border:
        mask[:, :, ::7] = 0
    label = rng.standard_normal((B, Co, HW)).astype(np.float32)
    w1 = (rng.standard_normal((Hd, C)) / np.sqrt(C)).astype(np.float32)
    b1 = rng.standard_normal((Hd,)).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((Co, Hd)) / np.sqrt(Hd)).astype(np.float32)
    b2 = rng.standard_normal((Co,)).astype(np.float32) * 0.1
    gext = rng.standard_normal((B, Co, HW)).astype(np.float32) if with_ext else None
    A, M, Lb = a.astype(f64), mask.astype(f64), label.astype(f64)
    h, z1, a1, preds_ref = _head_ref(A, M, Lb, w1.astype(f64), b1.astype(f64), w2.astype(f64), b2.astype(f64), act)
    lab_m = Lb * M
    res = {}
    da, dm, dl = be.dev(a), be.dev(mask), be.dev(label)
    dw1, db1, dw2, db2 = be.dev(w1), be.dev(b1), be.dev(w2), be.dev(b2)
    ws = be.bytes(api.size("cfd_fno_head_workspace_bytes", B, C, Hd, Co, HW))
    preds = be.zeros((B, Co, HW))
    sums = be.zeros((4,))
    api.call("cfd_fno_head_fwd", P(da), P(dm), P(dl), P(dw1), P(db1), P(dw2), P(db2), P(preds), P(sums), P(ws), B, C, Hd, Co,
             HW, int(act), be.stream)
    be.sync()
    res["preds"] = nm(be.host(preds), preds_ref)
    d = preds_ref - lab_m
    sref = np.array([np.sum(d * d), np.sum(np.abs(d)), np.sum(lab_m * lab_m), d.size])
    res["sums"] = float(np.max(np.abs(be.host(sums) - sref) / np.abs(sref)))
    scores = be.zeros((4,))
    api.call("cfd_loss_scores", P(sums), P(scores), be.stream)
    be.sync()
    lr = O.mse_loss(preds_ref, lab_m, True)
    sc = be.host(scores)
    res["scores"] = float(max(abs(sc[0] - lr["mse"]) / lr["mse"], abs(sc[1] - lr["rmse"]) / lr["rmse"],
                              abs(sc[2] - lr["mae"]) / lr["mae"], abs(sc[3] - lr["nmse"]) / lr["nmse"]))
    # backward
    coef = be.zeros((2,))
    api.call("cfd_loss_coef", P(sums), P(coef), {"mse": 0, "nmse": 1, "mae": 2}[which], 1.0, be.stream)
    gp = O.loss_grad_wrt_preds(preds_ref, lab_m, which)
    if with_ext:
        gp = gp + gext.astype(f64)
    graw = gp * M
    ga1 = np.einsum("cj,bcp->bjp", w2.astype(f64), graw)
    gz = ga1 * O.gelu_grad(z1)
    gh = np.einsum("ji,bjp->bip", w1.astype(f64), gz)
    ga_ref = gh * O.gelu_grad(A) if act else gh
    ga = be.zeros((B, C, HW))
    gw1, gb1, gw2, gb2 = be.zeros((Hd, C)), be.zeros((Hd,)), be.zeros((Co, Hd)), be.zeros((Co,))
    dgext = be.dev(gext) if with_ext else None
    api.call("cfd_fno_head_bwd", P(da), P(dm), P(dl), P(preds), P(dgext), P(coef), P(dw1),
             P(db1), P(dw2), P(ga), P(gw1), P(gb1), P(gw2), P(gb2), P(ws), B, C, Hd, Co, HW, int(act), be.stream)
    be.sync()
    res["ga"] = nm(be.host(ga), ga_ref)
    res["gw1"] = nm(be.host(gw1), np.einsum("bjp,bip->ji", gz, h))
    res["gb1"] = nm(be.host(gb1), gz.sum(axis=(0, 2)))
    res["gw2"] = nm(be.host(gw2), np.einsum("bcp,bjp->cj", graw, a1))
    res["gb2"] = nm(be.host(gb2), graw.sum(axis=(0, 2)))
    return res


def check_head_train(be, B, C, HW, act, which="nmse", border=True, seed=14):
    """cfd_label_energy_coef + cfd_fno_head_train (both directions of the head in one pass) against the same fp64 references as
    check_head: predictions, all four loss sums, d/da and the four parameter gradients."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    Hd, Co = 128, 2
    a = rng.standard_normal((B, C, HW)).astype(np.float32)
    mask = np.ones((B, 1, HW), np.float32)
    if border:
        mask[:, :, ::7] = 0
    label = rng.standard_normal((B, Co, HW)).astype(np.float32)
    w1 = (rng.standard_normal((Hd, C)) / np.sqrt(C)).astype(np.float32)
    b1 = rng.standard_normal((Hd,)).astype(np.float32) * 0.1
    w2 = (rng.standard_normal((Co, Hd)) / np.sqrt(Hd)).astype(np.float32)
    b2 = rng.standard_normal((Co,)).astype(np.float32) * 0.1
    A, M, Lb = a.astype(f64), mask.astype(f64), label.astype(f64)
    h, z1, a1, preds_ref = _head_ref(A, M, Lb, w1.astype(f64), b1.astype(f64), w2.astype(f64), b2.astype(f64), act)
    lab_m = Lb * M
    da, dm, dl = be.dev(a), be.dev(mask), be.dev(label)
    dw1, db1, dw2, db2 = be.dev(w1), be.dev(b1), be.dev(w2), be.dev(b2)
    ws = be.bytes(max(api.size("cfd_fno_head_workspace_bytes", B, C, Hd, Co, HW), api.size("cfd_label_energy_workspace_bytes")))
    preds, sums, coef = be.zeros((B, Co, HW)), be.zeros((4,)), be.zeros((2,))
    ga = be.zeros((B, C, HW))
    gw1, gb1, gw2, gb2 = be.zeros((Hd, C)), be.zeros((Hd,)), be.zeros((Co, Hd)), be.zeros((Co,))
    wid = {"mse": 0, "nmse": 1, "mae": 2}[which]
    api.call("cfd_label_energy_coef", P(dl), P(dm), P(sums), P(coef), P(ws), B, Co, HW, wid, 1.0, be.stream)
    api.call("cfd_fno_head_train", P(da), P(dm), P(dl), P(coef), P(dw1), P(db1), P(dw2), P(db2), P(preds), P(sums), P(ga), P(gw1),
             P(gb1), P(gw2), P(gb2), P(ws), B, C, Hd, Co, HW, int(act), be.stream)
    be.sync()
    res = {"preds": nm(be.host(preds), preds_ref)}
    d = preds_ref - lab_m
    sref = np.array([np.sum(d * d), np.sum(np.abs(d)), np.sum(lab_m * lab_m), d.size])
    res["sums"] = float(np.max(np.abs(be.host(sums) - sref) / np.abs(sref)))
    graw = O.loss_grad_wrt_preds(preds_ref, lab_m, which) * M
    ga1 = np.einsum("cj,bcp->bjp", w2.astype(f64), graw)
    gz = ga1 * O.gelu_grad(z1)
    gh = np.einsum("ji,bjp->bip", w1.astype(f64), gz)
    res["ga"] = nm(be.host(ga), gh * O.gelu_grad(A) if act else gh)
    res["gw1"] = nm(be.host(gw1), np.einsum("bjp,bip->ji", gz, h))
    res["gb1"] = nm(be.host(gb1), gz.sum(axis=(0, 2)))
    res["gw2"] = nm(be.host(gw2), np.einsum("bcp,bjp->cj", graw, a1))
    res["gb2"] = nm(be.host(gb2), graw.sum(axis=(0, 2)))
    return res


def check_loss_and_adam(be, n=10007, seed=5):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    p = rng.standard_normal(n).astype(np.float32)
    l = rng.standard_normal(n).astype(np.float32)
    ws = be.bytes(api.size("cfd_loss_workspace_bytes", n))
    sums = be.zeros((4,))
    dpp, dll = be.dev(p), be.dev(l)
    api.call("cfd_masked_loss_sums", P(dpp), P(dll), P(sums), P(ws), n, be.stream)
    be.sync()
    d = p.astype(f64) - l.astype(f64)
    ref = np.array([np.sum(d * d), np.sum(np.abs(d)), np.sum(l.astype(f64) ** 2), n])
    res = {"sums": float(np.max(np.abs(be.host(sums) - ref) / ref))}
    # Adam, 3 steps against the oracle (itself pinned against torch.optim.Adam by tests/test_oracle_golden.py)
    prm = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    dp, dm_, dv = be.dev(prm), be.dev(m), be.dev(v)
    p64, m64, v64 = prm.astype(f64), m.astype(f64), v.astype(f64)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * (10.0 ** rng.integers(-6, 1, size=n))
        g = g.astype(np.float32)
        dgr = be.dev(g)
        api.call("cfd_adam_flat", P(dp), P(dgr), P(dm_), P(dv), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, be.stream)
        be.sync()
        O.adam_step(p64, g.astype(f64), m64, v64, step, 1e-3)
    be.sync()
    res["adam_delta"] = nm(be.host(dp).astype(f64) - prm, p64 - prm)
    return res


def make_param_struct(be, params_dev, L):
    s = FnoParams()
    P = be.ptr
    s.fc0_w, s.fc0_b = P(params_dev["fc0.weight"]), P(params_dev["fc0.bias"])
    for l in range(L):
        s.spec_w1[l] = P(params_dev[f"blocks.{l}.conv0.weights1"])
        s.spec_w2[l] = P(params_dev[f"blocks.{l}.conv0.weights2"])
        s.w0_w[l] = P(params_dev[f"blocks.{l}.w0.weight"])
        s.w0_b[l] = P(params_dev[f"blocks.{l}.w0.bias"])
    s.fc1_w, s.fc1_b = P(params_dev["fc1.weight"]), P(params_dev["fc1.bias"])
    s.fc2_w, s.fc2_b = P(params_dev["fc2.weight"]), P(params_dev["fc2.bias"])
    return s


def run_fno(be, params, batch, L, C, H, W, p, with_label=True, which="nmse"):
    """Whole-model forward (+ backward) through cfd_fno_forward / cfd_fno_backward; returns host arrays."""
    api, P = be.api, be.ptr
    B = batch["inputs"].shape[0]
    plan = api.plan_create(H, W, 12, 12)
    try:
        shape = FnoShape(B, H, W, 2, 2, p, C, L, 12, 12, 128)
        pd = {k: be.dev(v) for k, v in params.items()}
        gd = {k: be.zeros(v.shape, np.complex64 if np.iscomplexobj(v) else np.float32) for k, v in params.items()}
        ps, gs = make_param_struct(be, pd, L), make_param_struct(be, gd, L)
        ws = be.bytes(api.size("cfd_fno_workspace_bytes", plan, ctypes.byref(shape), 1))
        di, dc, dm = be.dev(batch["inputs"]), be.dev(batch["case_params"]), be.dev(batch["mask"])
        dl = be.dev(batch["label"]) if with_label else None
        preds = be.zeros((B, 2, H, W))
        sums = be.zeros((4,))
        api.call("cfd_fno_forward", plan, ctypes.byref(shape), ctypes.byref(ps), P(di), P(dc), P(dm), P(dl), P(preds),
                 P(sums), P(ws), 1, be.stream)
        out = {}
        if with_label:
            coef = be.zeros((2,))
            api.call("cfd_loss_coef", P(sums), P(coef), {"mse": 0, "nmse": 1, "mae": 2}[which], 1.0, be.stream)
            api.call("cfd_fno_backward", plan, ctypes.byref(shape), ctypes.byref(ps), ctypes.byref(gs), P(di), P(dc), P(dm),
                     P(dl), P(preds), None, P(coef), P(ws), be.stream)
            scores = be.zeros((4,))
            api.call("cfd_loss_scores", P(sums), P(scores), be.stream)
            be.sync()
            out["grads"] = {k: be.host(v) for k, v in gd.items()}
            out["scores"] = be.host(scores)
        be.sync()
        out["preds"] = be.host(preds)
        # inference-mode workspace (ping-pong activations) must give the same predictions
        ws0 = be.bytes(api.size("cfd_fno_workspace_bytes", plan, ctypes.byref(shape), 0))
        preds0 = be.zeros((B, 2, H, W))
        api.call("cfd_fno_forward", plan, ctypes.byref(shape), ctypes.byref(ps), P(di), P(dc), P(dm), None, P(preds0), None,
                 P(ws0), 0, be.stream)
        be.sync()
        out["preds_infer"] = be.host(preds0)
        return out
    finally:
        api.plan_destroy(plan)


def _flat_struct(be, flat, layout, L):
    """cfd_fno_params whose tensors are slices of one flat float32 buffer (the training engine's layout)."""
    base = be.ptr(flat)
    s = FnoParams()
    at = lambda k: base + 4 * layout[k][0]  # noqa: E731
    s.fc0_w, s.fc0_b = at("fc0.weight"), at("fc0.bias")
    for l in range(L):
        s.spec_w1[l], s.spec_w2[l] = at(f"blocks.{l}.conv0.weights1"), at(f"blocks.{l}.conv0.weights2")
        s.w0_w[l], s.w0_b[l] = at(f"blocks.{l}.w0.weight"), at(f"blocks.{l}.w0.bias")
    s.fc1_w, s.fc1_b, s.fc2_w, s.fc2_b = at("fc1.weight"), at("fc1.bias"), at("fc2.weight"), at("fc2.bias")
    return s


def check_fno_train_step_deferred(be, B, C, L, H, W, p=5, which="nmse", flags=7, steps=2, border=True, pseed=27, bseed=28):
    """Round 6: the fused single-GPU training step with its three tiny launches folded into others (CFD_TRAIN_DEFER_*:
    cfd_fno_forward_train_f / cfd_fno_backward_phase_f / cfd_fno_adam_step) against the same calls with flags = 0 (the label-energy pair,
    the head's own reduction, the fc0 combine, cfd_adam_flat) -- parameters after `steps` Adam steps, predictions, loss sums, and the
    first step's gradient (rescaled by n / sum (label*mask)^2 where the normaliser was deferred) against the fp64 oracle."""
    api, P = be.api, be.ptr
    wid = {"mse": 0, "nmse": 1, "mae": 2}[which]
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=4.0)
    batch = synth.make_batch(bseed, B, H, W, p, border_mask=border)
    names = ["fc0.weight", "fc0.bias"] + [f"blocks.{l}.{t}" for l in range(L) for t in ("conv0.weights1", "conv0.weights2", "w0.weight", "w0.bias")] \
        + ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    layout, off = {}, 0
    for k in names:
        n = params[k].size * (2 if np.iscomplexobj(params[k]) else 1)
        layout[k] = (off, n)
        off += (n + 3) // 4 * 4
    numel = off
    flat0 = np.zeros(numel, np.float32)
    for k in names:
        v = params[k]
        flat0[layout[k][0]:layout[k][0] + layout[k][1]] = (np.stack([v.real, v.imag], -1) if np.iscomplexobj(v) else v).reshape(-1)
    plan = api.plan_create(H, W, 12, 12)
    try:
        shape = FnoShape(B, H, W, 2, 2, p, C, L, 12, 12, 128)
        di, dc, dm, dl = be.dev(batch["inputs"]), be.dev(batch["case_params"]), be.dev(batch["mask"]), be.dev(batch["label"])
        out = {}
        for fl in (0, flags):
            flat, grad = be.dev(flat0), be.zeros((numel,))
            m, v = be.zeros((numel,)), be.zeros((numel,))
            ps, gs = _flat_struct(be, flat, layout, L), _flat_struct(be, grad, layout, L)
            ws = be.bytes(api.size("cfd_fno_workspace_bytes", plan, ctypes.byref(shape), 1))
            preds, sums, coef = be.zeros((B, 2, H, W)), be.zeros((4,)), be.zeros((2,))
            g1 = None
            for step in range(1, steps + 1):
                api.call("cfd_fno_forward_train_f", plan, ctypes.byref(shape), ctypes.byref(ps), ctypes.byref(gs), P(di), P(dc), P(dm), P(dl),
                         P(preds), P(sums), P(coef), P(ws), wid, 1.0, 0, fl, be.stream)
                for phase in range(1, L + 2):
                    api.call("cfd_fno_backward_phase_f", plan, ctypes.byref(shape), ctypes.byref(ps), ctypes.byref(gs), P(di), P(dc), P(dm),
                             P(dl), P(preds), None, P(coef), P(sums), P(ws), phase, wid, 0, fl, be.stream)
                api.call("cfd_fno_adam_step", plan, ctypes.byref(shape), ctypes.byref(ps), ctypes.byref(gs), P(di), P(dc), P(dm), P(sums), P(ws),
                         P(flat), P(grad), P(m), P(v), numel, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, 1.0, wid, 0, fl, be.stream)
                be.sync()
                if step == 1:
                    g1 = be.host(grad).copy()
                    s1 = be.host(sums).copy()
                    if (fl & 1) and which == "nmse":
                        g1 = g1 * (s1[3] / s1[2])
                    out[fl] = dict(g1=g1, sums1=s1, preds1=be.host(preds).copy())
            out[fl]["flat"] = be.host(flat).copy()
        a, b = out[0], out[flags]
        res = {"params": nm(b["flat"], a["flat"]), "preds": nm(b["preds1"], a["preds1"]), "grad_vs_immediate": nm(b["g1"], a["g1"]),
               "sums": float(np.max(np.abs(b["sums1"] - a["sums1"]) / np.abs(a["sums1"])))}
        # the deferred step's first gradient against the oracle
        p64 = {k: v.astype(c128 if np.iscomplexobj(v) else f64) for k, v in params.items()}
        b64 = {k: v.astype(f64) for k, v in batch.items()}
        ref = O.fno_forward(p64, b64["inputs"], b64["case_params"], b64["mask"], b64["label"], L)
        rg = O.fno_backward(p64, ref["cache"], O.loss_grad_wrt_preds(ref["cache"]["preds"], ref["cache"]["label"], which), L)
        for k in ("fc0.weight", "fc0.bias", "fc1.weight", "fc2.bias", f"blocks.{L - 1}.conv0.weights1" if L else "fc1.bias"):
            got = b["g1"][layout[k][0]:layout[k][0] + layout[k][1]]
            want = rg[k]
            want = (np.stack([want.real, want.imag], -1) if np.iscomplexobj(want) else want).reshape(-1)
            res["oracle:" + k] = nm(got, want)
        return res
    finally:
        api.plan_destroy(plan)


def check_stem_dft_fusion(be, B, C, L, p, border, seed=61):
    """Round 6: the lifting layer fused into the first forward transform (stem_dft = 1) against the two launches (stem_dft = 0): predictions
    (training and inference workspaces) and every gradient bit for bit -- a_0 is built by the same fmaf chain and stored for the consumers."""
    params = synth.make_fno_params(seed, C, L, 12, 12, p, spectral_gain=4.0)
    batch = synth.make_batch(seed + 1, B, 64, 64, p, border_mask=border)
    with tuned(be, stem_dft=1):
        a = run_fno(be, params, batch, L, C, 64, 64, p)
    with tuned(be, stem_dft=0):
        b = run_fno(be, params, batch, L, C, 64, 64, p)
    res = {"preds": float(np.max(np.abs(a["preds"] - b["preds"]))), "preds_infer": float(np.max(np.abs(a["preds_infer"] - b["preds_infer"])))}
    for k in a["grads"]:
        res["g:" + k] = float(np.max(np.abs(a["grads"][k] - b["grads"][k])))
    return res


def check_fno_vs_oracle(be, B, C, L, H, W, p=5, border=False, gain=4.0, pseed=7, bseed=8):
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=gain)
    batch = synth.make_batch(bseed, B, H, W, p, border_mask=border)
    out = run_fno(be, params, batch, L, C, H, W, p)
    p64 = {k: v.astype(c128 if np.iscomplexobj(v) else f64) for k, v in params.items()}
    b64 = {k: v.astype(f64) for k, v in batch.items()}
    ref = O.fno_forward(p64, b64["inputs"], b64["case_params"], b64["mask"], b64["label"], L)
    gp = O.loss_grad_wrt_preds(ref["cache"]["preds"], ref["cache"]["label"], "nmse")
    rg = O.fno_backward(p64, ref["cache"], gp, L)
    res = {"preds": nm(out["preds"], ref["preds"]), "preds_infer": nm(out["preds_infer"], ref["preds"])}
    res["nmse_loss"] = abs(out["scores"][3] - ref["loss"]["nmse"]) / ref["loss"]["nmse"]
    for k in params:
        res["g:" + k] = nm(out["grads"][k], rg[k])
    return res


def check_fno_bf16_storage(be, B, C, L, H, W, p=5, border=True, gain=4.0, pseed=17, bseed=18):
    """cfd_fno_forward_ex with bf16 activation storage (BASELINE configs[4]) against the oracle with the SAME storage rule:
    the lifting layer's output and every FnoBlock's pre-activation rounded to bf16 (RNE) where they are stored, everything
    else fp32 / fp64.  An fp32 value that sits within round-off of a bf16 rounding boundary may fall to the other side than
    the fp64 oracle's (one element in ~2e4, one bf16 ulp each), which bounds the agreement near 1e-9; ``vs_f32`` reports how
    far the bf16-storage predictions are from the fp32 path (the price of the storage format, not an error)."""
    api, P = be.api, be.ptr
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=gain)
    batch = synth.make_batch(bseed, B, H, W, p, border_mask=border)
    plan = api.plan_create(H, W, 12, 12)
    try:
        shape = FnoShape(B, H, W, 2, 2, p, C, L, 12, 12, 128)
        pd = {k: be.dev(v) for k, v in params.items()}
        ps = make_param_struct(be, pd, L)
        di, dc, dm, dl = (be.dev(batch[k]) for k in ("inputs", "case_params", "mask", "label"))
        out = {}
        for name, dt in (("bf16", 1), ("f32", 0)):
            nbytes = api.size("cfd_fno_workspace_bytes_ex", plan, ctypes.byref(shape), 0, dt)
            assert nbytes > 0
            ws = be.bytes(nbytes)
            preds, sums = be.zeros((B, 2, H, W)), be.zeros((4,))
            api.call("cfd_fno_forward_ex", plan, ctypes.byref(shape), ctypes.byref(ps), P(di), P(dc), P(dm), P(dl), P(preds),
                     P(sums), P(ws), 0, dt, be.stream)
            be.sync()
            out[name] = (be.host(preds), be.host(sums))
        # bf16-storage TRAINING has its own workspace (round 3): saved activations at 2 bytes + the fp32 scratch tensor
        assert api.size("cfd_fno_workspace_bytes_ex", plan, ctypes.byref(shape), 1, 1) > 0
        p64 = {k: v.astype(c128 if np.iscomplexobj(v) else f64) for k, v in params.items()}
        b64 = {k: v.astype(f64) for k, v in batch.items()}
        ref16 = O.fno_forward(p64, b64["inputs"], b64["case_params"], b64["mask"], b64["label"], L, act_store=O.bf16_round,
                              keep_cache=False)
        ref32 = O.fno_forward(p64, b64["inputs"], b64["case_params"], b64["mask"], b64["label"], L, keep_cache=False)
        return {"bf16_vs_bf16_oracle": nm(out["bf16"][0], ref16["preds"]), "f32_vs_oracle": nm(out["f32"][0], ref32["preds"]),
                "bf16_loss": abs(out["bf16"][1][0] / out["bf16"][1][2] - ref16["loss"]["nmse"]) / ref16["loss"]["nmse"],
                "info:bf16_vs_f32": nm(out["bf16"][0], out["f32"][0]), "info:oracle_bf16_vs_f32": nm(ref16["preds"], ref32["preds"])}
    finally:
        api.plan_destroy(plan)


# ---- dense layers of the DeepONet family (csrc/dense.hip) ---------------------------------------------------------
def check_gemm(be, M, N, K, ta, tb, seed=21):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    Bm = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    dA, dB, C = be.dev(A), be.dev(Bm), be.zeros((M, N))
    ws = be.bytes(api.size("cfd_gemm_workspace_bytes", M, N, K))
    api.call("cfd_gemm", P(dA), P(dB), P(C), P(ws), M, N, K, A.shape[1], Bm.shape[1], N, int(ta), int(tb), be.stream)
    be.sync()
    ref = (A.T if ta else A).astype(f64) @ (Bm.T if tb else Bm).astype(f64)
    return {"c": nm(be.host(C), ref)}


def check_linear(be, M, K, N, act, seed=22):
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"none": 0, "relu": 1, "tanh": 2, "gelu": 3, "swish": 4}[act]
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal((N,)).astype(np.float32) * 0.3
    gy = rng.standard_normal((M, N)).astype(np.float32)
    dx, dw, db, dgy = be.dev(x), be.dev(w), be.dev(b), be.dev(gy)
    y, pre = be.zeros((M, N)), be.zeros((M, N))
    wsf = be.bytes(api.size("cfd_linear_fwd_workspace_bytes", M, K, N))
    api.call("cfd_linear_fwd", P(dx), P(dw), P(db), P(y), P(pre), P(wsf), M, K, N, code, be.stream)
    be.sync()
    z = x.astype(f64) @ w.astype(f64).T + b
    res = {"y": nm(be.host(y), D.act(z, act)), "pre": nm(be.host(pre), z)}
    gx, gw, gb = be.zeros((M, K)), be.zeros((N, K)), be.zeros((N,))
    ws = be.bytes(api.size("cfd_linear_bwd_workspace_bytes", M, K, N))
    api.call("cfd_linear_bwd", P(dgy), P(dx), P(dw), P(y), P(pre), P(gx), P(gw), P(gb), P(ws), M, K, N, code, be.stream)
    be.sync()
    gz = gy.astype(f64) * D.act_grad(z, act)
    res["gx"] = nm(be.host(gx), gz @ w.astype(f64))
    res["gw"] = nm(be.host(gw), gz.T @ x.astype(f64))
    res["gb"] = nm(be.host(gb), gz.sum(axis=0))
    return res


def check_ffn_stack(be, R, dims, act, act_last=False, with_gx=True, seed=29):
    """cfd_ffn_stack_fwd / _bwd (a whole Linear+activation stack per kernel, csrc/ffn.hip) against the fp64 layer-by-layer
    restatement of Ffn.forward (src/models/ffn.py:12-35) and its reverse pass."""
    import ctypes
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"none": 0, "relu": 1, "tanh": 2, "gelu": 3, "swish": 4}[act]
    rng = np.random.default_rng(seed)
    L = len(dims) - 1
    x = rng.standard_normal((R, dims[0])).astype(np.float32)
    ws = [(rng.standard_normal((dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32) for l in range(L)]
    bs = [(0.3 * rng.standard_normal((dims[l + 1],))).astype(np.float32) for l in range(L)]
    gy = rng.standard_normal((R, dims[L])).astype(np.float32)
    dx, dgy = be.dev(x), be.dev(gy)
    dws, dbs = [be.dev(w) for w in ws], [be.dev(b) for b in bs]
    ys = [be.zeros((R, d)) for d in dims[1:]]
    acted = [code != 0 and (l + 1 < L or act_last) for l in range(L)]
    zs = [be.zeros((R, dims[l + 1])) if (code >= 3 and acted[l]) else None for l in range(L)]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[P(t) for t in ts])  # noqa: E731
    cdims = (ctypes.c_int * (L + 1))(*dims)
    api.call("cfd_ffn_stack_fwd", P(dx), arr(dws), arr(dbs), arr(ys), arr(zs), R, cdims, L, code, int(act_last), be.stream)
    be.sync()
    # fp64 restatement
    h, zs64, ys64 = x.astype(f64), [], []
    for l in range(L):
        z = h @ ws[l].astype(f64).T + bs[l]
        h = D.act(z, act) if acted[l] else z
        zs64.append(z)
        ys64.append(h)
    res = {f"y{l}": nm(be.host(ys[l]), ys64[l]) for l in range(L)}
    for l in range(L):
        if zs[l] is not None:
            res[f"z{l}"] = nm(be.host(zs[l]), zs64[l])
    gws, gbs = [be.zeros(w.shape) for w in ws], [be.zeros(b.shape) for b in bs]
    gx = be.zeros((R, dims[0])) if with_gx else None
    wsb = be.bytes(api.size("cfd_ffn_stack_bwd_workspace_bytes", R, cdims, L))
    api.call("cfd_ffn_stack_bwd", P(dx), P(dgy), arr(dws), arr(ys), arr(zs), arr(gws), arr(gbs), P(gx), P(wsb), R, cdims, L, code,
             int(act_last), be.stream)
    be.sync()
    g = gy.astype(f64)
    for l in reversed(range(L)):
        gz = g * D.act_grad(zs64[l], act) if acted[l] else g
        inp = ys64[l - 1] if l > 0 else x.astype(f64)
        res[f"gw{l}"] = nm(be.host(gws[l]), gz.T @ inp)
        res[f"gb{l}"] = nm(be.host(gbs[l]), gz.sum(axis=0))
        g = gz @ ws[l].astype(f64)
    if with_gx:
        res["gx"] = nm(be.host(gx), g)
    return res


def check_rows_concat2(be, rows, ka, lda, kb, ldb, seed=73):
    """cfd_rows_concat2: [a | b] from rows at strides lda / ldb, exactly."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((rows, lda)).astype(np.float32)
    Bm = rng.standard_normal((rows, ldb)).astype(np.float32)
    dA, dB, out = be.dev(A), be.dev(Bm), be.zeros((rows, ka + kb))
    api.call("cfd_rows_concat2", P(dA), lda, ka, P(dB), ldb, kb, P(out), rows, be.stream)
    be.sync()
    return {"differs": float(np.max(np.abs(be.host(out) - np.concatenate([A[:, :ka], Bm[:, :kb]], axis=1))))}


def check_mse_loss_strided_labels(be, rows, cols, ldl, seed=71):
    """cfd_mse_loss_fwd_ld / _bwd_ld (ABI 601): labels as rows at stride ldl against the contiguous entry points on a contiguous copy -- sums,
    scores and the prediction gradient bit for bit (the element -> (row, column) split is the only difference) -- and against fp64."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    preds = rng.standard_normal((rows, cols)).astype(np.float32)
    big = rng.standard_normal((rows, ldl)).astype(np.float32)
    lab = np.ascontiguousarray(big[:, :cols])
    g = [np.array([v], np.float32) for v in (0.3, -0.2, 0.7, 1.1)]
    dp, dbig, dlab = be.dev(preds), be.dev(big), be.dev(lab)
    dg = [be.dev(v) for v in g]
    out = {}
    for name in ("ld", "contig"):
        sums, scores, gp = be.zeros((4,)), be.zeros((4,)), be.zeros((rows, cols))
        ws = be.bytes(api.size("cfd_loss_workspace_bytes", rows * cols))
        if name == "ld":
            api.call("cfd_mse_loss_fwd_ld", P(dp), P(dbig), P(sums), P(scores), P(ws), rows, cols, ldl, be.stream)
            api.call("cfd_mse_loss_bwd_ld", P(dp), P(dbig), P(sums), *[P(v) for v in dg], P(gp), rows, cols, ldl, be.stream)
        else:
            api.call("cfd_mse_loss_fwd", P(dp), P(dlab), P(sums), P(scores), P(ws), rows * cols, be.stream)
            api.call("cfd_mse_loss_bwd", P(dp), P(dlab), P(sums), *[P(v) for v in dg], P(gp), None, rows * cols, be.stream)
        be.sync()
        out[name] = (be.host(sums), be.host(scores), be.host(gp))
    d = preds.astype(f64) - lab.astype(f64)
    mse, den = float((d * d).mean()), float((lab.astype(f64) ** 2).mean())
    ref = np.array([mse, np.sqrt(mse), np.abs(d).mean(), mse / den])
    return {"sums_differ": float(np.max(np.abs(out["ld"][0] - out["contig"][0]))), "scores_differ": float(np.max(np.abs(out["ld"][1] - out["contig"][1]))),
            "gp_differ": float(np.max(np.abs(out["ld"][2] - out["contig"][2]))), "scores_rel": float(np.max(np.abs(out["ld"][1] - ref) / ref))}


def check_linear_rowgemm6(be, M, K, N, act, in_act=None, seed=41, force=True):
    """Round 6: a Linear layer's forward product and input gradient on k_rowgemm6 (three-piece bf16 operands, the weights pre-split
    into MFMA fragments; tall products, M >= 4096 by default -- `force`: gemm_b3 = 2 runs it at any row count) against the fp64 layer,
    and against the fp32-MFMA kernel (gemm_b3 = 0): both are fp32-exact-class, they differ by accumulation order only."""
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"none": 0, "relu": 1, "tanh": 2, "gelu": 3, "swish": 4}
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal((N,)).astype(np.float32) * 0.3
    gy = rng.standard_normal((M, N)).astype(np.float32)
    xin = x
    xpre = None
    if in_act:  # x is the output of a previous layer with activation in_act (the input gradient leaves as that layer's dZ)
        xpre = x
        xin = D.act(x.astype(f64), in_act).astype(np.float32)
    outs = {}
    for knob in ((2 if force else -1), 0):
        with tuned(be, gemm_b3=knob):
            dx, dw, db, dgy = be.dev(xin), be.dev(w), be.dev(b), be.dev(gy)
            dxpre = be.dev(xpre) if xpre is not None else None
            y, pre = be.zeros((M, N)), be.zeros((M, N))
            wsf = be.bytes(api.size("cfd_linear_fwd_workspace_bytes", M, K, N))
            api.call("cfd_linear_fwd", P(dx), P(dw), P(db), P(y), P(pre), P(wsf), M, K, N, code[act], be.stream)
            gx, gw, gb = be.zeros((M, K)), be.zeros((N, K)), be.zeros((N,))
            wsb = be.bytes(api.size("cfd_linear_bwd_workspace_bytes", M, K, N))
            api.call("cfd_linear_bwd_ex", P(dgy), P(dx), P(dw), P(y), P(pre), P(gx), P(gw), P(gb), P(wsb), M, K, N, code[act],
                     code[in_act] if in_act else 0, P(dxpre) if dxpre is not None else None, be.stream)
            be.sync()
            outs[knob] = (be.host(y), be.host(pre), be.host(gx), be.host(gw), be.host(gb))
    z = xin.astype(f64) @ w.astype(f64).T + b
    gz = gy.astype(f64) * D.act_grad(z, act)
    gx_ref = gz @ w.astype(f64)
    if in_act:
        gx_ref = gx_ref * D.act_grad(xpre.astype(f64), in_act)
    a, c = outs[2 if force else -1], outs[0]
    return {"y": nm(a[0], D.act(z, act)), "pre": nm(a[1], z), "gx": nm(a[2], gx_ref), "gw": nm(a[3], gz.T @ xin.astype(f64)),
            "gb": nm(a[4], gz.sum(0)), "y_vs_fp32_kernel": nm(a[0], c[0].astype(f64)), "gx_vs_fp32_kernel": nm(a[2], c[2].astype(f64))}


def check_linear_chain_bwd(be, M, K, N, in_act, seed=37):
    """cfd_linear_bwd_ex with in_act: the input gradient leaves the GEMM as the previous layer's dZ = (gz w) * in_act'(x) -- against
    the fp64 product and, bit for bit, against cfd_linear_bwd followed by cfd_act_bwd (one fp32 multiply after the same GEMM)."""
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"relu": 1, "tanh": 2, "gelu": 3, "swish": 4}[in_act]
    rng = np.random.default_rng(seed)
    zin = rng.standard_normal((M, K)).astype(np.float32)           # the previous layer's pre-activation ...
    x = D.act(zin.astype(f64), in_act).astype(np.float32)          # ... and its output = this layer's input
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    gz = rng.standard_normal((M, N)).astype(np.float32)
    dx, dz, dw, dg = be.dev(x), be.dev(zin), be.dev(w), be.dev(gz)
    gx1, gw1, gb1 = be.zeros((M, K)), be.zeros((N, K)), be.zeros((N,))
    ws = be.bytes(api.size("cfd_linear_bwd_workspace_bytes", M, K, N))
    api.call("cfd_linear_bwd_ex", P(dg), P(dx), P(dw), None, None, P(gx1), P(gw1), P(gb1), P(ws), M, K, N, 0, code,
             P(dz) if code >= 3 else None, be.stream)
    gx0, gw0, gb0, gx2 = be.zeros((M, K)), be.zeros((N, K)), be.zeros((N,)), be.zeros((M, K))
    api.call("cfd_linear_bwd", P(dg), P(dx), P(dw), None, None, P(gx0), P(gw0), P(gb0), P(ws), M, K, N, 0, be.stream)
    api.call("cfd_act_bwd", P(gx0), P(dx), P(dz), P(gx2), M * K, code, be.stream)
    be.sync()
    ref = (gz.astype(f64) @ w.astype(f64)) * D.act_grad(zin.astype(f64), in_act)
    return {"gz_prev": nm(be.host(gx1), ref), "differs_from_two_passes": float(np.count_nonzero(be.host(gx1) != be.host(gx2))),
            "gw_differs": float(np.count_nonzero(be.host(gw1) != be.host(gw0)))}


def check_ffn_stacks(be, specs, seed=31):
    """cfd_ffn_stacks_fwd / _bwd: several stacks in one launch per direction must give BITWISE what the single-stack calls give (same
    kernels, same per-stack work decomposition).  specs: [(R, dims, act name, act_last, with_gx), ...].  Returns the number of
    differing elements per output (all zero) -- the single-stack calls themselves are pinned against the fp64 restatement above."""
    import ctypes
    from cfdbench_amd._capi import FfnStackArgs
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    codes = {"none": 0, "relu": 1, "tanh": 2, "gelu": 3, "swish": 4}
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[P(t) for t in ts])  # noqa: E731
    n = len(specs)
    data, keep = [], []
    for R, dims, act, act_last, with_gx in specs:
        L, code = len(dims) - 1, codes[act]
        d = dict(R=R, dims=dims, L=L, code=code, act_last=act_last, with_gx=with_gx,
                 x=be.dev(rng.standard_normal((R, dims[0])).astype(np.float32)), gy=be.dev(rng.standard_normal((R, dims[L])).astype(np.float32)),
                 w=[be.dev((rng.standard_normal((dims[l + 1], dims[l])) / np.sqrt(dims[l])).astype(np.float32)) for l in range(L)],
                 b=[be.dev((0.3 * rng.standard_normal((dims[l + 1],))).astype(np.float32)) for l in range(L)],
                 cdims=(ctypes.c_int * (L + 1))(*dims))
        acted = [code != 0 and (l + 1 < L or act_last) for l in range(L)]
        for tag in ("one", "many"):  # outputs of the single-stack calls / of the joint call
            d[tag] = dict(y=[be.zeros((R, dd)) for dd in dims[1:]],
                          z=[be.zeros((R, dims[l + 1])) if (code >= 3 and acted[l]) else None for l in range(L)],
                          gw=[be.zeros((dims[l + 1], dims[l])) for l in range(L)], gb=[be.zeros((dims[l + 1],)) for l in range(L)],
                          gx=be.zeros((R, dims[0])) if with_gx else None,
                          ws=be.bytes(api.size("cfd_ffn_stack_bwd_workspace_bytes", R, d["cdims"], L)))
        data.append(d)
    for d in data:
        o = d["one"]
        api.call("cfd_ffn_stack_fwd", P(d["x"]), arr(d["w"]), arr(d["b"]), arr(o["y"]), arr(o["z"]), d["R"], d["cdims"], d["L"], d["code"],
                 int(d["act_last"]), be.stream)
        api.call("cfd_ffn_stack_bwd", P(d["x"]), P(d["gy"]), arr(d["w"]), arr(o["y"]), arr(o["z"]), arr(o["gw"]), arr(o["gb"]), P(o["gx"]),
                 P(o["ws"]), d["R"], d["cdims"], d["L"], d["code"], int(d["act_last"]), be.stream)
    args = (FfnStackArgs * n)()
    for a, d in zip(args, data):
        o = d["many"]
        arrs = [arr(d["w"]), arr(d["b"]), arr(o["y"]), arr(o["z"]), arr(o["gw"]), arr(o["gb"])]
        keep.append(arrs)
        a.x, a.gy, a.gx, a.ws = P(d["x"]), P(d["gy"]), P(o["gx"]), P(o["ws"])
        a.R, a.L, a.act, a.act_last = d["R"], d["L"], d["code"], int(d["act_last"])
        a.w, a.b, a.y, a.z, a.gw, a.gb = (ctypes.addressof(t) for t in arrs)
        a.dims = ctypes.addressof(d["cdims"])
    api.call("cfd_ffn_stacks_fwd", n, args, be.stream)
    api.call("cfd_ffn_stacks_bwd", n, args, be.stream)
    be.sync()
    res = {}
    for i, d in enumerate(data):
        for key in ("y", "z", "gw", "gb"):
            for l in range(d["L"]):
                if d["one"][key][l] is not None:
                    res[f"s{i}.{key}{l}"] = int(np.count_nonzero(be.host(d["one"][key][l]) != be.host(d["many"][key][l])))
        if d["with_gx"]:
            res[f"s{i}.gx"] = int(np.count_nonzero(be.host(d["one"]["gx"]) != be.host(d["many"]["gx"])))
    return res


def check_deeponet_inner(be, B, P_, Kq, HW, with_q, seed=23):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    br = rng.standard_normal((B, P_)).astype(np.float32)
    tr = rng.standard_normal((Kq, P_)).astype(np.float32)
    bias = np.array([0.37], np.float32)
    u = rng.standard_normal((B, HW)).astype(np.float32)
    q = rng.integers(0, HW, size=Kq).astype(np.int32) if with_q else None
    g = rng.standard_normal((B, Kq)).astype(np.float32)
    dbr, dtr, dbi, du, dg = be.dev(br), be.dev(tr), be.dev(bias), be.dev(u), be.dev(g)
    dq = be.dev(q) if with_q else None
    preds = be.zeros((B, Kq))
    api.call("cfd_deeponet_inner_fwd", P(dbr), P(dtr), P(dbi), P(du), P(dq), P(preds), B, P_, Kq, HW, be.stream)
    be.sync()
    resid = u[:, q] if with_q else u[:, :Kq]
    res = {"preds": nm(be.host(preds), br.astype(f64) @ tr.astype(f64).T + 0.37 + resid)}
    p2 = be.zeros((B, Kq))
    api.call("cfd_deeponet_inner_fwd", P(dbr), P(dtr), P(dbi), None, None, P(p2), B, P_, Kq, 0, be.stream)  # deeponet.py:205
    be.sync()
    res["preds_nores"] = nm(be.host(p2), br.astype(f64) @ tr.astype(f64).T + 0.37)
    gbr, gtr, gbi = be.zeros((B, P_)), be.zeros((Kq, P_)), be.zeros((1,))
    ws = be.bytes(api.size("cfd_deeponet_inner_bwd_workspace_bytes", B, P_, Kq))
    api.call("cfd_deeponet_inner_bwd", P(dg), P(dbr), P(dtr), P(gbr), P(gtr), P(gbi), P(ws), B, P_, Kq, be.stream)
    be.sync()
    res["gbranch"] = nm(be.host(gbr), g.astype(f64) @ tr.astype(f64))
    res["gtrunk"] = nm(be.host(gtr), g.astype(f64).T @ br.astype(f64))
    res["gbias"] = float(abs(be.host(gbi)[0] - g.astype(f64).sum()) / abs(g.astype(f64).sum()))
    return res


# ---- convolution stack of the U-Net / ResNet baselines (csrc/conv.hip) ------------------------------------------
def check_conv2d(be, B, Ci, Co, H, W, ks, seed=31):
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks)).astype(np.float32)
    b = rng.standard_normal((Co,)).astype(np.float32) * 0.2
    g = rng.standard_normal((B, Co, H, W)).astype(np.float32)
    dx, dw, db, dg = be.dev(x), be.dev(w), be.dev(b), be.dev(g)
    out = be.zeros((B, Co, H, W))
    nws = api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks)
    fws = be.bytes(nws) if nws else None
    api.call("cfd_conv2d_fwd", P(dx), P(dw), P(db), P(out), P(fws) if nws else None, B, Ci, Co, H, W, ks, be.stream)
    be.sync()
    res = {"out": nm(be.host(out), CO.conv2d(x.astype(f64), w.astype(f64), b.astype(f64)))}
    ws = be.bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks))
    gin, gw, gb = be.zeros((B, Ci, H, W)), be.zeros((Co, Ci, ks, ks)), be.zeros((Co,))
    api.call("cfd_conv2d_bwd", P(dg), P(dx), P(dw), P(gin), P(gw), P(gb), P(ws), B, Ci, Co, H, W, ks, be.stream)
    be.sync()
    rgx, rgw, rgb = CO.conv2d_bwd(g.astype(f64), x.astype(f64), w.astype(f64))
    res["gin"], res["gw"], res["gb"] = nm(be.host(gin), rgx), nm(be.host(gw), rgw), nm(be.host(gb), rgb)
    return res


def check_conv2d_zeropad(be, B, Ci, Co, H, W, ks, seed=41):
    """cfd_conv2d_zeropad_fwd / _bwd (nn.Conv2d's default zero padding on the conv6 kernels) against the fp64 oracle: the oracle's
    replicate-padded convolution of the input zero-padded by k / 2, cropped -- the construction the model used before -- and its
    reverse pass (gradient of the crop = zero-padding the upstream gradient; gradient of the pad = cropping)."""
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    if api.size("cfd_conv2d_zeropad_supported", B, Ci, Co, H, W, ks) != 1:
        return None  # (the caller zero-pads and crops around the replicate kernel: models/auto_deeponet_cnn.py)
    rng = np.random.default_rng(seed)
    p = ks // 2
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks)).astype(np.float32)
    b = rng.standard_normal((Co,)).astype(np.float32) * 0.2
    g = rng.standard_normal((B, Co, H, W)).astype(np.float32)
    dx, dw, db, dg = be.dev(x), be.dev(w), be.dev(b), be.dev(g)
    out = be.zeros((B, Co, H, W))
    fws = be.bytes(api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks))
    api.call("cfd_conv2d_zeropad_fwd", P(dx), P(dw), P(db), P(out), P(fws), B, Ci, Co, H, W, ks, be.stream)
    ws = be.bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks))
    gin, gw, gb = be.zeros((B, Ci, H, W)), be.zeros((Co, Ci, ks, ks)), be.zeros((Co,))
    api.call("cfd_conv2d_zeropad_bwd", P(dg), P(dx), P(dw), P(gin), P(gw), P(gb), P(ws), B, Ci, Co, H, W, ks, be.stream)
    be.sync()
    pad4 = ((0, 0), (0, 0), (p, p), (p, p))
    xp, gp = np.pad(x.astype(f64), pad4), np.pad(g.astype(f64), pad4)
    ref = CO.conv2d(xp, w.astype(f64), b.astype(f64))[:, :, p:-p, p:-p]
    rgx, rgw, rgb = CO.conv2d_bwd(gp, xp, w.astype(f64))
    return {"out": nm(be.host(out), ref), "gin": nm(be.host(gin), rgx[:, :, p:-p, p:-p]), "gw": nm(be.host(gw), rgw), "gb": nm(be.host(gb), rgb)}


def check_conv_prepared(be, layers, seed=37):
    """cfd_conv2d_wprep_batch + cfd_conv2d_fwd_ex / cfd_conv2d_bwd_ex: the fragments of ALL `layers` = [(B, Ci, Co, H, W, ks), ...]
    made by one call (forward and input-gradient forms interleaved), then every layer run with them -- must equal the calls that
    prepare their own weights BIT FOR BIT (same fragments, same kernels).  Returns the number of values that differ."""
    import ctypes
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    data, items = [], []
    for (B, Ci, Co, H, W, ks) in layers:
        x = be.dev(rng.standard_normal((B, Ci, H, W)).astype(np.float32))
        w = be.dev((rng.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks)).astype(np.float32))
        b = be.dev(rng.standard_normal((Co,)).astype(np.float32) * 0.2)
        g = be.dev(rng.standard_normal((B, Co, H, W)).astype(np.float32))
        fr = []
        for tr in (0, 1):
            n = api.size("cfd_conv2d_wfrag_bytes", Ci, Co, ks, tr)
            assert n > 0, (Ci, Co, ks, tr)
            fr.append(be.bytes(n))
            items.append((P(w), P(fr[-1]), Ci, Co, ks, tr))
        data.append((x, w, b, g, fr))
    n = len(items)
    col = lambda j, ty: (ty * n)(*[it[j] for it in items])
    api.call("cfd_conv2d_wprep_batch", n, col(0, ctypes.c_void_p), col(1, ctypes.c_void_p), col(2, ctypes.c_int), col(3, ctypes.c_int),
             col(4, ctypes.c_int), col(5, ctypes.c_int), be.stream)
    be.sync()
    bad = 0
    for (B, Ci, Co, H, W, ks), (x, w, b, g, fr) in zip(layers, data):
        outs = []
        for prepared in (False, True):
            out = be.zeros((B, Co, H, W))
            nws = api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks)
            assert nws > 0
            fws = be.bytes(nws)
            slots = api.size("cfd_conv2d_fwd_stats_slots", B, Ci, Co, H, W, ks)
            stats = be.zeros((Co, max(slots, 1), 4))
            api.call("cfd_conv2d_fwd_ex", P(x), P(w), P(b), P(out), P(fws), P(stats) if slots > 0 else None,
                     P(fr[0]) if prepared else None, B, Ci, Co, H, W, ks, be.stream)
            ws = be.bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks))
            gin, gw, gb = be.zeros((B, Ci, H, W)), be.zeros((Co, Ci, ks, ks)), be.zeros((Co,))
            api.call("cfd_conv2d_bwd_ex", P(g), P(x), P(w), P(gin), P(gw), P(gb), P(ws), P(fr[1]) if prepared else None, B, Ci, Co, H, W,
                     ks, be.stream)
            be.sync()
            outs.append([be.host(t).copy() for t in (out, stats, gin, gw, gb)])
        for a, c in zip(*outs):
            bad += int(np.count_nonzero(a != c))
    return bad


def check_conv_bn_stats(be, B, Ci, Co, H, W, ks, relu=True, seed=36, offset=0.0, spread=1.0):
    """cfd_conv2d_fwd_stats + cfd_batchnorm_fwd_stats (the conv emits the BatchNorm's batch statistics) against the oracle's
    conv -> training-mode BatchNorm; returns None when the layer cannot emit statistics.  `offset` / `spread`: input = offset +
    spread * noise under all-positive weights -- output channels whose mean is far from the bias and |mean| >> std."""
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    slots = api.size("cfd_conv2d_fwd_stats_slots", B, Ci, Co, H, W, ks)
    if slots <= 0:
        return None
    rng = np.random.default_rng(seed)
    x = (offset + spread * rng.standard_normal((B, Ci, H, W))).astype(np.float32)
    w = (rng.standard_normal((Co, Ci, ks, ks)) / np.sqrt(Ci * ks * ks)).astype(np.float32)
    if offset != 0.0:
        w = np.abs(w)
    b = (rng.standard_normal((Co,)) * 0.2 + 3.0 * rng.standard_normal((Co,))).astype(np.float32)  # |mean| >> std in some channels
    gamma = (1 + 0.3 * rng.standard_normal(Co)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(Co)).astype(np.float32)
    rm = (0.5 * rng.standard_normal(Co)).astype(np.float32)
    rv = (1 + rng.random(Co)).astype(np.float32)
    dx, dw, db, dga, dbe, drm, drv = be.dev(x), be.dev(w), be.dev(b), be.dev(gamma), be.dev(beta), be.dev(rm), be.dev(rv)
    out, y, sm, sr = be.zeros((B, Co, H, W)), be.zeros((B, Co, H, W)), be.zeros((Co,)), be.zeros((Co,))
    stats = be.zeros((Co, slots, 4))
    ws = be.bytes(api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks))
    api.call("cfd_conv2d_fwd_stats", P(dx), P(dw), P(db), P(out), P(ws), P(stats), B, Ci, Co, H, W, ks, be.stream)
    api.call("cfd_batchnorm_fwd_stats", P(out), P(dga), P(dbe), P(drm), P(drv), P(y), P(sm), P(sr), P(stats), slots, P(db), B, Co,
             H * W, 1e-5, 0.1, int(relu), be.stream)
    be.sync()
    r0 = CO.conv2d(x.astype(f64), w.astype(f64), b.astype(f64))
    ry, cache, nrm, nrv = CO.batchnorm(r0, gamma.astype(f64), beta.astype(f64), rm.astype(f64), rv.astype(f64), True, relu=relu)
    return {"out": nm(be.host(out), r0), "y": nm(be.host(y), ry), "run_mean": nm(be.host(drm), nrm), "run_var": nm(be.host(drv), nrv)}


def check_batchnorm(be, B, C, H, W, training, relu, seed=32):
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((B, C, H, W)) * 1.7 + 3.0 * rng.standard_normal((1, C, 1, 1))).astype(np.float32)
    gamma = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(C)).astype(np.float32)
    rm = (0.5 * rng.standard_normal(C)).astype(np.float32)
    rv = (1 + rng.random(C)).astype(np.float32)
    gy = rng.standard_normal((B, C, H, W)).astype(np.float32)
    dx, dga, dbe, drm, drv, dgy = be.dev(x), be.dev(gamma), be.dev(beta), be.dev(rm), be.dev(rv), be.dev(gy)
    y, sm, sr = be.zeros((B, C, H, W)), be.zeros((C,)), be.zeros((C,))
    ws = be.bytes(api.size("cfd_batchnorm_workspace_bytes", C))
    api.call("cfd_batchnorm_fwd", P(dx), P(dga), P(dbe), P(drm), P(drv), P(y), P(sm), P(sr), P(ws), B, C, H * W, 1e-5, 0.1,
             int(training), int(relu), be.stream)
    be.sync()
    ry, cache, nrm, nrv = CO.batchnorm(x.astype(f64), gamma.astype(f64), beta.astype(f64), rm.astype(f64), rv.astype(f64),
                                       training, relu=relu)
    res = {"y": nm(be.host(y), ry), "run_mean": nm(be.host(drm), nrm), "run_var": nm(be.host(drv), nrv)}
    gx, gga, gbe = be.zeros((B, C, H, W)), be.zeros((C,)), be.zeros((C,))
    api.call("cfd_batchnorm_bwd", P(dgy), P(dx), P(dga), P(dbe), P(sm), P(sr), P(gx), P(gga), P(gbe), P(ws), B, C, H * W,
             int(training), int(relu), be.stream)
    be.sync()
    rgx, rgg, rgb = CO.batchnorm_bwd(gy.astype(f64), cache)
    res["gx"], res["ggamma"], res["gbeta"] = nm(be.host(gx), rgx), nm(be.host(gga), rgg), nm(be.host(gbe), rgb)
    return res


def check_convt(be, B, Ci, Co, H, W, seed=34, x=None, rng=None):
    """cfd_convt2_fwd / cfd_convt2_bwd (ConvTranspose2d(2, stride 2), unet.py:80) against the oracle"""
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = rng if rng is not None else np.random.default_rng(seed)
    if x is None:
        x = np.maximum(rng.standard_normal((B, Ci, H, W)), 0).astype(np.float32)
    dx = be.dev(x)
    w = (rng.standard_normal((Ci, Co, 2, 2)) / np.sqrt(Ci)).astype(np.float32)
    b = rng.standard_normal((Co,)).astype(np.float32) * 0.1
    g = rng.standard_normal((B, Co, 2 * H, 2 * W)).astype(np.float32)
    dw, db, dg = be.dev(w), be.dev(b), be.dev(g)
    out = be.zeros((B, Co, 2 * H, 2 * W))
    api.call("cfd_convt2_fwd", P(dx), P(dw), P(db), P(out), B, Ci, Co, H, W, be.stream)
    ws = be.bytes(api.size("cfd_convt2_bwd_workspace_bytes", B, Ci, Co, H, W))
    gin, gw, gb = be.zeros((B, Ci, H, W)), be.zeros((Ci, Co, 2, 2)), be.zeros((Co,))
    api.call("cfd_convt2_bwd", P(dg), P(dx), P(dw), P(gin), P(gw), P(gb), P(ws), B, Ci, Co, H, W, be.stream)
    be.sync()
    res = {"convt": nm(be.host(out), CO.convt2(x.astype(f64), w.astype(f64), b.astype(f64)))}
    rgx, rgw, rgb = CO.convt2_bwd(g.astype(f64), x.astype(f64), w.astype(f64))
    res["convt_gin"], res["convt_gw"], res["convt_gb"] = nm(be.host(gin), rgx), nm(be.host(gw), rgw), nm(be.host(gb), rgb)
    return res


def check_pool_convt_resid(be, B, Ci, Co, H, W, seed=33):
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = np.maximum(rng.standard_normal((B, Ci, H, W)), 0).astype(np.float32)  # ReLU-like input: ties at zero
    dx = be.dev(x)
    Ho, Wo = H // 2, W // 2
    y = be.zeros((B, Ci, Ho, Wo))
    api.call("cfd_maxpool2_fwd", P(dx), P(y), B * Ci, H, W, be.stream)
    gy = rng.standard_normal((B, Ci, Ho, Wo)).astype(np.float32)
    dgy, gx = be.dev(gy), be.zeros((B, Ci, H, W))
    api.call("cfd_maxpool2_bwd", P(dx), P(dgy), P(gx), B * Ci, H, W, be.stream)
    be.sync()
    res = {"pool": float(np.abs(be.host(y) - CO.maxpool2(x)).max()),
           "pool_bwd": float(np.abs(be.host(gx) - CO.maxpool2_bwd(x, gy)).max())}
    # the same with the skip connection's gradient summed in: a channel slice (first Ci of Ci + 3 channels) of a larger tensor
    gcat = rng.standard_normal((B, Ci + 3, H, W)).astype(np.float32)
    dgcat, gx2 = be.dev(gcat), be.zeros((B, Ci, H, W))
    api.call("cfd_maxpool2_bwd_add", P(dx), P(dgy), P(dgcat), (Ci + 3) * H * W, P(gx2), B, Ci, H, W, be.stream)
    be.sync()
    res["pool_bwd_add"] = float(np.abs(be.host(gx2) - (be.host(gx) + gcat[:, :Ci])).max())  # one fp32 add: exact
    res.update(check_convt(be, B, Ci, Co, H, W, x=x, rng=rng))
    C2 = min(2, Ci)
    mask = (rng.random((B, H * W)) > 0.2).astype(np.float32)
    xs = rng.standard_normal((B, C2, H * W)).astype(np.float32)
    dxs, dmask, o2 = be.dev(xs), be.dev(mask), be.zeros((B, C2, H * W))
    api.call("cfd_residual_mask", P(dxs), P(dx), P(dmask), P(o2), B, C2, Ci, H * W, be.stream)
    be.sync()
    res["resid"] = nm(be.host(o2), (xs + x.reshape(B, Ci, -1)[:, :C2]) * mask[:, None])
    return res


def check_upsample_bilinear(be, B, C, H, W, seed=35):
    """cfd_upsample2_bilinear_fwd / _bwd against the oracle's restatement of nn.Upsample(2, bilinear, align_corners=True) and its
    adjoint, plus the adjoint identity <up(x), g> == <x, up^T(g)> on the kernels themselves."""
    from oracle import conv_oracle as CO
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    g = rng.standard_normal((B, C, 2 * H, 2 * W)).astype(np.float32)
    dx, dg = be.dev(x), be.dev(g)
    y, gx = be.zeros((B, C, 2 * H, 2 * W)), be.zeros((B, C, H, W))
    api.call("cfd_upsample2_bilinear_fwd", P(dx), P(y), B * C, H, W, be.stream)
    api.call("cfd_upsample2_bilinear_bwd", P(dg), P(gx), B * C, H, W, be.stream)
    be.sync()
    hy, hgx = be.host(y).astype(f64), be.host(gx).astype(f64)
    res = {"up": nm(hy, CO.upsample2_bilinear(x.astype(f64))), "up_bwd": nm(hgx, CO.upsample2_bilinear_bwd(g.astype(f64)))}
    lhs, rhs = float((hy * g).sum()), float((hgx * x).sum())
    res["adjoint"] = (lhs - rhs) ** 2 / max(float((hy ** 2).sum() * (g.astype(f64) ** 2).sum()), 1e-300)
    return res


# ---- NormAct and the non-autoregressive DeepONet pieces (csrc/dense.hip) ---------------------------------------------
def check_normact(be, S, shape, act, seed=41):
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"relu": 1, "tanh": 2, "gelu": 3, "swish": 4}[act]
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((S,) + shape) * 1.3 + 0.4).astype(np.float32)
    g = rng.standard_normal((S,) + shape).astype(np.float32)
    L = int(np.prod(shape))
    dx, dg = be.dev(x), be.dev(g)
    y, stats, gx = be.zeros((S,) + shape), be.zeros((S, 2)), be.zeros((S,) + shape)
    api.call("cfd_normact_fwd", P(dx), P(y), P(stats), S, L, code, be.stream)
    api.call("cfd_normact_bwd", P(dx), P(dg), P(stats), P(gx), S, L, code, be.stream)
    be.sync()
    ry, cache = D.normact(x.astype(f64), act)
    return {"y": nm(be.host(y), ry), "gx": nm(be.host(gx), D.normact_bwd(g.astype(f64), cache))}


def check_bcast_rowdot(be, B, K, P_, seed=42):
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    ft, fxy = rng.standard_normal((B, P_)).astype(np.float32), rng.standard_normal((K, P_)).astype(np.float32)
    g3 = rng.standard_normal((B, K, P_)).astype(np.float32)
    dft, dfxy, dg3 = be.dev(ft), be.dev(fxy), be.dev(g3)
    out, gft, gfxy = be.zeros((B, K, P_)), be.zeros((B, P_)), be.zeros((K, P_))
    api.call("cfd_bcast_add_fwd", P(dft), P(dfxy), P(out), B, K, P_, be.stream)
    api.call("cfd_bcast_add_bwd", P(dg3), P(gft), P(gfxy), B, K, P_, be.stream)
    be.sync()
    res = {"bcast": nm(be.host(out), ft[:, None, :].astype(f64) + fxy[None].astype(f64)),
           "gft": nm(be.host(gft), g3.astype(f64).sum(axis=1)), "gfxy": nm(be.host(gfxy), g3.astype(f64).sum(axis=0))}
    br, tr = rng.standard_normal((B, P_)).astype(np.float32), rng.standard_normal((B, K, P_)).astype(np.float32)
    bias, g2 = np.array([0.21], np.float32), rng.standard_normal((B, K)).astype(np.float32)
    dbr, dtr, dbi, dg2 = be.dev(br), be.dev(tr), be.dev(bias), be.dev(g2)
    preds, gbr, gtr, gbi = be.zeros((B, K)), be.zeros((B, P_)), be.zeros((B, K, P_)), be.zeros((1,))
    ws = be.bytes(api.size("cfd_rowdot_bwd_workspace_bytes"))
    api.call("cfd_rowdot_fwd", P(dbr), P(dtr), P(dbi), P(preds), B, K, P_, be.stream)
    api.call("cfd_rowdot_bwd", P(dg2), P(dbr), P(dtr), P(gbr), P(gtr), P(gbi), P(ws), B, K, P_, be.stream)
    be.sync()
    res["preds"] = nm(be.host(preds), np.einsum("bp,bkp->bk", br.astype(f64), tr.astype(f64)) + 0.21)
    res["gbranch"] = nm(be.host(gbr), np.einsum("bk,bkp->bp", g2.astype(f64), tr.astype(f64)))
    res["gtrunk"] = nm(be.host(gtr), g2.astype(f64)[:, :, None] * br.astype(f64)[:, None, :])
    res["gbias"] = float(abs(be.host(gbi)[0] - g2.astype(f64).sum()) / abs(g2.astype(f64).sum()))
    return res


def check_act(be, n, act, seed=43):
    """stand-alone activation kernels vs the oracle's act / act_grad"""
    from oracle import deeponet_oracle as D
    api, P = be.api, be.ptr
    code = {"relu": 1, "tanh": 2, "gelu": 3, "swish": 4}[act]
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 1.5).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    dx, dg = be.dev(x), be.dev(g)
    y, gx = be.zeros((n,)), be.zeros((n,))
    api.call("cfd_act_fwd", P(dx), P(y), n, code, be.stream)
    api.call("cfd_act_bwd", P(dg), P(y), P(dx), P(gx), n, code, be.stream)
    be.sync()
    return {"y": nm(be.host(y), D.act(x.astype(f64), act)), "gx": nm(be.host(gx), g.astype(f64) * D.act_grad(x.astype(f64), act))}


def check_dropout_gelu(be, n, p, seed=5):
    """cfd_dropout_gelu_fwd / _bwd against cfd_dropout + cfd_gelu_fwd / cfd_gelu_bwd + cfd_dropout: the same values bit for bit, and
    against the exact-erf GELU of numpy on the kept elements.  Returns (values that differ, nMSE of the forward vs numpy)."""
    import math
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = (2.5 * rng.standard_normal(n)).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    dx, dg = be.dev(x), be.dev(g)
    y1, gx1, d, y2, t, gx2 = (be.zeros((n,)) for _ in range(6))
    api.call("cfd_dropout_gelu_fwd", P(dx), P(y1), n, p, 1234, be.stream)
    api.call("cfd_dropout_gelu_bwd", P(dx), P(dg), P(gx1), n, p, 1234, be.stream)
    api.call("cfd_dropout", P(dx), P(d), n, p, 1234, be.stream)
    api.call("cfd_gelu_fwd", P(d), P(y2), n, be.stream)
    api.call("cfd_gelu_bwd", P(d), P(dg), P(t), n, be.stream)
    api.call("cfd_dropout", P(t), P(gx2), n, p, 1234, be.stream)
    be.sync()
    bad = int(np.count_nonzero(be.host(y1) != be.host(y2))) + int(np.count_nonzero(be.host(gx1) != be.host(gx2)))
    dh = be.host(d).astype(f64)
    from scipy.special import erf
    ref = 0.5 * dh * (1.0 + erf(dh / math.sqrt(2.0)))
    return bad, nm(be.host(y1), ref)


def check_dropout_step(be, n, p, base, step, seed=7):
    """cfd_dropout_gelu_fwd_step / _bwd_step (the stream's step counter in device memory, the seed formed by the kernel) against the
    host-seeded calls with seed = splitmix64(base + step) & (2^48 - 1): the same values bit for bit.  Returns the number that differ."""
    api, P = be.api, be.ptr

    def mix64(v):
        v &= 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return v ^ (v >> 31)
    rng = np.random.default_rng(seed)
    x = (2.5 * rng.standard_normal(n)).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    dx, dg = be.dev(x), be.dev(g)
    dstep = be.dev(np.array([step], dtype=np.int64))
    y1, gx1, y2, gx2 = (be.zeros((n,)) for _ in range(4))
    host_seed = mix64(base + step) & 0xFFFFFFFFFFFF
    api.call("cfd_dropout_gelu_fwd_step", P(dx), P(y1), n, p, base, P(dstep), be.stream)
    api.call("cfd_dropout_gelu_bwd_step", P(dx), P(dg), P(gx1), n, p, base, P(dstep), be.stream)
    api.call("cfd_dropout_gelu_fwd", P(dx), P(y2), n, p, host_seed, be.stream)
    api.call("cfd_dropout_gelu_bwd", P(dx), P(dg), P(gx2), n, p, host_seed, be.stream)
    be.sync()
    kept = float(np.count_nonzero(be.host(y1))) / n
    assert abs(kept - (1.0 - p)) < 0.05, kept
    return int(np.count_nonzero(be.host(y1) != be.host(y2))) + int(np.count_nonzero(be.host(gx1) != be.host(gx2)))


def check_loss_scores_bwd(be, seed=6):
    """cfd_loss_scores / cfd_loss_scores_bwd against the formulas of loss.py:27-35 and their derivatives in float64; returns the
    largest relative error over the scores and over d(score)/d(sums) for each score alone and for all four together."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    worst = 0.0
    for _ in range(5):
        n = float(rng.integers(10, 100000))
        sums = np.array([rng.random() * n, rng.random() * n, (0.5 + rng.random()) * n, n], np.float32)
        g = rng.standard_normal(4).astype(np.float32)
        ds, scores = be.dev(sums), be.zeros((4,))
        api.call("cfd_loss_scores", P(ds), P(scores), be.stream)
        s0, s1, s2, nn = (float(v) for v in sums)
        ref = np.array([s0 / nn, np.sqrt(s0 / nn), s1 / nn, s0 / s2])
        be.sync()
        worst = max(worst, float(np.abs(be.host(scores) - ref).max() / np.abs(ref).max()))
        jac = np.array([[1 / nn, 0, 0], [0.5 / np.sqrt(s0 / nn) / nn, 0, 0], [0, 1 / nn, 0], [1 / s2, 0, -s0 / s2 ** 2]])  # d score_i / d sums_j
        for pick in ([0], [1], [2], [3], [0, 1, 2, 3]):
            gd = [be.dev(g[i:i + 1]) if i in pick else None for i in range(4)]
            out = be.zeros((4,))
            api.call("cfd_loss_scores_bwd", P(ds), *[P(t) if t is not None else None for t in gd], P(out), be.stream)
            be.sync()
            want = sum(g[i] * jac[i] for i in pick)
            got = be.host(out)
            assert got[3] == 0.0
            worst = max(worst, float(np.abs(got[:3] - want).max() / max(np.abs(want).max(), 1e-30)))
    return worst


def check_adam_multi(be, sizes=(7, 1025, 300, 1), steps=3, seed=8, shift_odd=False):
    """cfd_adam_multi (many tensors, one launch, step count and rate from device scalars) against cfd_adam_flat tensor by tensor:
    the same update up to the fp32 rounding of the bias corrections.  ``shift_odd``: every second tensor's four buffers start one
    element off the 16-byte grid (the one-element-per-thread path beside the 16-byte-unit one in ONE launch).  Returns the largest
    relative difference of the parameters."""
    import ctypes
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    sh = [1 if (shift_odd and t % 2) else 0 for t in range(len(sizes))]
    pad = lambda a, s_: np.concatenate([np.zeros(s_, np.float32), a])
    p0 = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    pa, ma, va = ([be.dev(pad(t, s_)) for t, s_ in zip(p0, sh)], [be.zeros((n + s_,)) for n, s_ in zip(sizes, sh)],
                  [be.zeros((n + s_,)) for n, s_ in zip(sizes, sh)])
    pb, mb, vb = [be.dev(t) for t in p0], [be.zeros((n,)) for n in sizes], [be.zeros((n,)) for n in sizes]
    lr = be.dev(np.array([2e-3], np.float32))
    col = lambda ts: (ctypes.c_void_p * len(ts))(*[P(t) + 4 * s_ for t, s_ in zip(ts, sh)])
    for k in range(1, steps + 1):
        gh = [rng.standard_normal(n).astype(np.float32) for n in sizes]
        ga, g = [be.dev(pad(t, s_)) for t, s_ in zip(gh, sh)], [be.dev(t) for t in gh]
        stp = be.dev(np.array([float(k)], np.float32))
        api.call("cfd_adam_multi", len(sizes), col(pa), col(ga), col(ma), col(va), (ctypes.c_size_t * len(sizes))(*sizes), P(lr), 0.0, P(stp), 0.0,
                 0.9, 0.999, 1e-8, 0.01, 1.0, be.stream)
        for t in range(len(sizes)):
            api.call("cfd_adam_flat", P(pb[t]), P(g[t]), P(mb[t]), P(vb[t]), sizes[t], 2e-3, 0.9, 0.999, 1e-8, 0.01, k, 1.0, be.stream)
        be.sync()
    return max(float(np.abs(be.host(a)[s_:] - be.host(b)).max() / np.abs(be.host(b)).max()) for a, b, s_ in zip(pa, pb, sh))


def check_adam_flat_unaligned(be, n=1003, seed=10):
    """cfd_adam_flat on buffers that are NOT 16-byte aligned (element offset 1: the one-element-per-thread kernel) against the same
    update on aligned buffers (16-byte units, four elements per thread + a scalar tail).  Returns the number of differing values."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    host = [rng.standard_normal(n).astype(np.float32) for _ in range(2)] + [np.abs(rng.standard_normal(n)).astype(np.float32) * 1e-3 for _ in range(2)]
    al = [be.dev(t) for t in host]                                         # p, g, m, v
    un = [be.dev(np.concatenate([np.zeros(1, np.float32), t])) for t in host]
    api.call("cfd_adam_flat", P(al[0]), P(al[1]), P(al[2]), P(al[3]), n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 2, 0.5, be.stream)
    api.call("cfd_adam_flat", P(un[0]) + 4, P(un[1]) + 4, P(un[2]) + 4, P(un[3]) + 4, n, 1e-3, 0.9, 0.999, 1e-8, 0.01, 2, 0.5, be.stream)
    be.sync()
    bad = 0
    for a, u, h in zip(al, un, host):
        got = be.host(u)
        bad += int(np.sum(got[1:] != be.host(a))) + int(got[0] != 0.0)
    return bad + int(np.array_equal(be.host(al[0]), host[0]))  # (and the update did change the parameters)


def check_scale_copy_multi(be, sizes=(7, 1025, 300, 1, 70001), scale=0.25, seed=9):
    """cfd_scale_copy_multi (the gradient pack of the data-parallel autograd paths): dst[k] = scale * src[k] for many tensors in one
    launch per 80, written into ONE flat buffer at 4-float-aligned offsets; exact (a power-of-two scale), the gaps untouched.
    Returns the number of wrong values."""
    import ctypes
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    src = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) // 4 * 4
    flat = be.dev(np.full((off,), 7.0, np.float32))
    dsrc = [be.dev(t) for t in src]
    api.call("cfd_scale_copy_multi", len(sizes), (ctypes.c_void_p * len(sizes))(*[P(t) for t in dsrc]),
             (ctypes.c_void_p * len(sizes))(*[P(flat) + 4 * o for o in offs]), (ctypes.c_size_t * len(sizes))(*sizes), float(scale), be.stream)
    be.sync()
    got = be.host(flat)
    want = np.full((off,), 7.0, np.float32)
    for t, o in zip(src, offs):
        want[o:o + t.size] = t * np.float32(scale)
    return int((got != want).sum())


def check_convt_strided(be, B, Ci, Co, H, W, C2=3, seed=40):
    """cfd_convt2_fwd_ex / cfd_convt2_bwd_ex with the 2H x 2W tensor as the trailing Co channels of a (C2 + Co)-channel one: the same
    values as the dense calls, bit for bit, and the leading C2 channels untouched.  Returns the number of values that differ."""
    api, P = be.api, be.ptr
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Ci, H, W)).astype(np.float32)
    w = (rng.standard_normal((Ci, Co, 2, 2)) / np.sqrt(Ci)).astype(np.float32)
    b = rng.standard_normal((Co,)).astype(np.float32) * 0.1
    gwide = rng.standard_normal((B, C2 + Co, 2 * H, 2 * W)).astype(np.float32)
    dx, dw, db = be.dev(x), be.dev(w), be.dev(b)
    plane = 4 * H * W
    out = be.zeros((B, Co, 2 * H, 2 * W))
    api.call("cfd_convt2_fwd", P(dx), P(dw), P(db), P(out), B, Ci, Co, H, W, be.stream)
    wide = be.dev(np.full((B, C2 + Co, 2 * H, 2 * W), 7.0, np.float32))
    api.call("cfd_convt2_fwd_ex", P(dx), P(dw), P(db), P(wide) + 4 * C2 * plane, (C2 + Co) * plane, B, Ci, Co, H, W, be.stream)
    ws = be.bytes(api.size("cfd_convt2_bwd_workspace_bytes", B, Ci, Co, H, W))
    dg = be.dev(np.ascontiguousarray(gwide[:, C2:]))
    gin, gw, gb = be.zeros((B, Ci, H, W)), be.zeros((Ci, Co, 2, 2)), be.zeros((Co,))
    api.call("cfd_convt2_bwd", P(dg), P(dx), P(dw), P(gin), P(gw), P(gb), P(ws), B, Ci, Co, H, W, be.stream)
    dgw = be.dev(gwide)
    gin2, gw2, gb2 = be.zeros((B, Ci, H, W)), be.zeros((Ci, Co, 2, 2)), be.zeros((Co,))
    api.call("cfd_convt2_bwd_ex", P(dgw) + 4 * C2 * plane, (C2 + Co) * plane, P(dx), P(dw), P(gin2), P(gw2), P(gb2), P(ws), B, Ci, Co, H, W,
             be.stream)
    be.sync()
    hw = be.host(wide)
    bad = int(np.count_nonzero(hw[:, C2:] != be.host(out))) + int(np.count_nonzero(hw[:, :C2] != 7.0))
    for a, c in ((gin, gin2), (gw, gw2), (gb, gb2)):
        bad += int(np.count_nonzero(be.host(a) != be.host(c)))
    return bad
