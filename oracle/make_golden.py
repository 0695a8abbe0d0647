"""Generate tests/golden/*.npz from the REAL reference modules (imported from /root/reference/src).

Run in the build container only (the GPU box has no /root/reference):
    python oracle/make_golden.py
The reference ships no golden vectors of its own, so these files are what pins
parity: inputs come from oracle/synth.py's NumPy streams (regenerated in the tests from the seeds
stored in each file), outputs/gradients come from the reference's PyTorch-CPU fp32 forward/backward.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, "/root/reference/src")

from models.fno.fno2d import Fno2d, SpectralConv2d_fast  # noqa: E402  (reference)
from models.loss import MseLoss  # noqa: E402  (reference)

from oracle import synth  # noqa: E402
from oracle.make_golden_inputs import spectral_case  # noqa: E402

OUT = REPO / "tests" / "golden"


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def gen_spectral(name, seed, B, Cin, Cout, H, W, m1, m2):
    x, gy, w1, w2 = spectral_case(seed, B, Cin, Cout, H, W, m1, m2)
    mod = SpectralConv2d_fast(Cin, Cout, m1, m2)
    with torch.no_grad():
        mod.weights1.copy_(_t(w1))
        mod.weights2.copy_(_t(w2))
    xt = _t(x).requires_grad_(True)
    y = mod(xt)
    y.backward(_t(gy))
    np.savez_compressed(
        OUT / f"{name}.npz",
        meta=np.array([seed, B, Cin, Cout, H, W, m1, m2]),
        y=y.detach().numpy(), gx=xt.grad.numpy(),
        gw1=mod.weights1.grad.numpy(), gw2=mod.weights2.grad.numpy(),
    )
    print(name, "ok", y.shape)


def build_ref_model(params, C, L, m1, m2, p):
    model = Fno2d(2, 2, p, MseLoss(normalize=True), L, m1, m2, C)
    sd = {k: _t(v) for k, v in params.items()}
    model.load_state_dict(sd)
    return model


def summarize(a: np.ndarray, idx_seed: int, n: int = 32):
    flat = np.ascontiguousarray(a).reshape(-1)
    rng = np.random.default_rng(idx_seed)
    idx = rng.integers(0, flat.size, size=min(n, flat.size))
    return dict(norm=np.sqrt(np.sum(np.abs(flat.astype(np.complex128)) ** 2)), sum=flat.sum(dtype=np.complex128),
                idx=idx, vals=flat[idx])


def gen_fno(name, pseed, bseed, B, C, L, H, W, p=5, border=False, full_grads=True, gain=1.0):
    m1 = m2 = 12
    params = synth.make_fno_params(pseed, C, L, m1, m2, p, spectral_gain=gain)
    batch = synth.make_batch(bseed, B, H, W, p, border_mask=border)
    model = build_ref_model(params, C, L, m1, m2, p)
    tb = {k: _t(v) for k, v in batch.items()}
    tb["inputs"].requires_grad_(True)
    out = model(**tb)
    out["loss"]["nmse"].backward()
    save = dict(
        meta=np.array([pseed, bseed, B, C, L, H, W, p, int(border)]), gain=np.array(gain),
        preds=out["preds"].detach().numpy(),
        **{f"loss_{k}": v.detach().numpy() for k, v in out["loss"].items()},
        g_inputs=tb["inputs"].grad.numpy(),
    )
    for k, prm in model.named_parameters():
        g = prm.grad.numpy()
        if full_grads:
            save[f"grad::{k}"] = g
        else:
            s = summarize(g, 7)
            for kk, vv in s.items():
                save[f"gsum::{k}::{kk}"] = vv
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_rollout(name, pseed, bseed, B, C, L, H, W, steps, p=5, border=False, gain=1.0):
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=gain)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    if border:
        batch["mask"][:, :, 0, :] = 0
        batch["mask"][:, :, -1, :] = 0
        batch["mask"][:, :, :, 0] = 0
    model = build_ref_model(params, C, L, 12, 12, p).eval()
    with torch.no_grad():
        frames = model.generate_many(_t(batch["inputs"]), _t(batch["case_params"]), _t(batch["mask"]), steps)
    frames = np.stack([f.numpy() for f in frames])  # (steps,B,2,H,W)
    np.savez_compressed(
        OUT / f"{name}.npz", meta=np.array([pseed, bseed, B, C, L, H, W, p, steps, int(border)]),
        gain=np.array(gain),
        last=frames[-1], first=frames[0], norms=np.sqrt((frames ** 2).mean(axis=(1, 2, 3, 4))),
    )
    print(name, "ok", frames.shape)


def gen_auto_deeponet(name, pseed, bseed, B, H, W, width, bdepth, tdepth, act_name="relu", p=5, steps=3):
    """AutoDeepONet forward / loss / backward / rollout from the reference module (src/models/auto_deeponet.py)."""
    from models.auto_deeponet import AutoDeepONet  # reference
    from oracle import deeponet_oracle as D
    params = D.make_params(pseed, H * W + p, width, bdepth, tdepth)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    model = AutoDeepONet(H * W + p, 2, MseLoss(normalize=True), branch_depth=bdepth, trunk_depth=tdepth, width=width,
                         act_name=act_name)
    model.load_state_dict({k: _t(v) for k, v in params.items()})
    x = _t(batch["inputs"]).requires_grad_(True)
    out = model(inputs=x, case_params=_t(batch["case_params"]), label=_t(batch["label"]), mask=_t(batch["mask"]))
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([pseed, bseed, B, H, W, width, bdepth, tdepth, p, steps]), act=np.array(act_name),
                preds=out["preds"].detach().numpy(), g_inputs=x.grad.numpy(),
                **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, prm in model.named_parameters():
        save[f"grad::{k}"] = prm.grad.numpy()
    model.eval()
    with torch.no_grad():
        frames = model.generate_many(_t(batch["inputs"]), _t(batch["case_params"]), _t(batch["mask"]), steps)
    save["frames"] = np.stack([f.numpy() for f in frames])
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_auto_variants(name, kind, seed, bseed, B, H, W, width, depth, act_name="relu", p=5, steps=2, nq=37):
    """AutoEDeepONet / AutoFfn from the reference modules (src/models/auto_edeeponet.py, auto_ffn.py): training forward
    on random query points + loss + backward, full-lattice inference forward and a short rollout."""
    torch.manual_seed(seed)
    if kind == "auto_edeeponet":
        from models.auto_edeeponet import AutoEDeepONet  # reference
        model = AutoEDeepONet(H * W, p, 2, MseLoss(normalize=True), branch_depth=depth, trunk_depth=depth, width=width,
                              act_name=act_name)
        with torch.no_grad():
            model.bias.fill_(0.03)
    else:
        from models.auto_ffn import AutoFfn  # reference
        model = AutoFfn(H * W, p, 2, MseLoss(normalize=True), depth=depth, width=width, act_name=act_name)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    g = torch.Generator().manual_seed(bseed + 7)
    q = torch.stack([torch.randint(0, H, (nq,), generator=g), torch.randint(0, W, (nq,), generator=g)], dim=-1)
    x = _t(batch["inputs"]).requires_grad_(True)
    out = model(inputs=x, case_params=_t(batch["case_params"]), label=_t(batch["label"]), mask=_t(batch["mask"]),
                query_idxs=q)
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([seed, bseed, B, H, W, width, depth, p, steps, nq]), act=np.array(act_name),
                kind=np.array(kind), q=q.numpy(), preds=out["preds"].detach().numpy(), g_inputs=x.grad.numpy(),
                **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, v in model.state_dict().items():
        save[f"sd::{k}"] = v.numpy().copy()
    for k, prm in model.named_parameters():
        save[f"grad::{k}"] = prm.grad.numpy()
    model.eval()
    with torch.no_grad():
        full = model(inputs=_t(batch["inputs"]), case_params=_t(batch["case_params"]), mask=_t(batch["mask"]))["preds"]
        save["preds_full"] = full.numpy()
        frames = model.generate_many(_t(batch["inputs"]), _t(batch["case_params"]), _t(batch["mask"]), steps)
    save["frames"] = np.stack([f.numpy() for f in frames])
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_auto_deeponet_cnn(name, seed, bseed, B, trunk_depth=2, p=5, steps=2, nq=33):
    """AutoDeepONetCnn (CNN branch with zero-padded 5x5 convs; needs 64x64 frames) from the reference module
    (src/models/auto_deeponet_cnn.py): training forward on random query points + loss + backward, rollout."""
    from models.auto_deeponet_cnn import AutoDeepONetCnn  # reference
    H = W = 64
    torch.manual_seed(seed)
    model = AutoDeepONetCnn(2, 2, MseLoss(normalize=True), height=H, width=W, num_case_params=p, trunk_depth=trunk_depth)
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, 0, :] = 0
    g = torch.Generator().manual_seed(bseed + 7)
    q = torch.stack([torch.randint(0, H, (nq,), generator=g), torch.randint(0, W, (nq,), generator=g)], dim=-1)
    x = _t(batch["inputs"]).requires_grad_(True)
    out = model(inputs=x, case_params=_t(batch["case_params"]), label=_t(batch["label"]), mask=_t(batch["mask"]),
                query_idxs=q)
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([seed, bseed, B, trunk_depth, p, steps, nq]), q=q.numpy(),
                preds=out["preds"].detach().numpy(), g_inputs=x.grad.numpy(),
                **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, v in model.state_dict().items():
        save[f"sd::{k}"] = v.numpy().copy()
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            save[f"grad::{k}"] = prm.grad.numpy()
    model.eval()
    with torch.no_grad():
        frames = model.generate_many(_t(batch["inputs"][:1]), _t(batch["case_params"][:1]), _t(batch["mask"][:1, 0]), steps)
    save["frames"] = np.stack([f.numpy() for f in frames])
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_unet(name, seed, bseed, B, H, W, dim, p=8, steps=2, insert="input", bilinear=False):
    """UNet (input-insert, ConvTranspose up path) train-mode forward/backward, running-stat update, eval forward and
    rollout from the reference module (src/models/unet.py).  The state_dict itself is stored (torch's init stream)."""
    from models.unet import UNet  # reference
    torch.manual_seed(seed)
    model = UNet(2, 2, MseLoss(normalize=True), p, insert_case_params_at=insert, bilinear=bilinear, dim=dim)
    with torch.no_grad():  # non-trivial BatchNorm affine parameters and running statistics
        g = torch.Generator().manual_seed(seed + 1)
        for k, v in model.state_dict().items():
            if k.endswith(".1.weight"):
                v.copy_(1 + 0.2 * torch.randn(v.shape, generator=g))
            elif k.endswith(".1.bias") or k.endswith("running_mean"):
                v.copy_(0.1 * torch.randn(v.shape, generator=g))
            elif k.endswith("running_var"):
                v.copy_(1 + 0.5 * torch.rand(v.shape, generator=g))
    sd0 = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, 0, :] = 0
    batch["mask"][:, :, :, 0] = 0
    model.train()
    x = _t(batch["inputs"]).requires_grad_(True)
    out = model(inputs=x, case_params=_t(batch["case_params"]), mask=_t(batch["mask"]), label=_t(batch["label"]))
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([seed, bseed, B, H, W, dim, p, steps]), preds_train=out["preds"].detach().numpy(),
                g_inputs=x.grad.numpy(), **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, v in sd0.items():
        save[f"sd::{k}"] = v
    for k, prm in model.named_parameters():
        save[f"grad::{k}"] = prm.grad.numpy()
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            save[f"after::{k}"] = v.numpy().copy()
    model.load_state_dict({k: _t(v) for k, v in sd0.items()})  # the eval outputs use the ORIGINAL running statistics (sd::*)
    model.eval()
    with torch.no_grad():
        save["preds_eval"] = model(inputs=_t(batch["inputs"]), case_params=_t(batch["case_params"]),
                                   mask=_t(batch["mask"]))["preds"].numpy()
        frames = model.generate_many(_t(batch["inputs"][0]), _t(batch["case_params"][0]), _t(batch["mask"][0, 0]), steps)
    save["frames"] = np.stack([f.numpy() for f in frames])
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_resnet(name, seed, bseed, B, H, W, hidden, nblocks, p=5, steps=2):
    """ResNet in EVAL mode (dropout = identity; torch's dropout stream is not reproducible): forward, loss, backward and
    rollout from the reference module (src/models/resnet.py)."""
    from models.resnet import ResNet  # reference
    torch.manual_seed(seed)
    model = ResNet(2, 2, p, MseLoss(normalize=True), hidden_chan=hidden, num_blocks=nblocks, kernel_size=7, padding=3)
    model.eval()
    sd0 = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, -1, :] = 0
    x = _t(batch["inputs"]).requires_grad_(True)
    out = model(inputs=x, case_params=_t(batch["case_params"]), mask=_t(batch["mask"]), label=_t(batch["label"]))
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([seed, bseed, B, H, W, hidden, nblocks, p, steps]), preds=out["preds"].detach().numpy(),
                g_inputs=x.grad.numpy(), **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, v in sd0.items():
        save[f"sd::{k}"] = v
    for k, prm in model.named_parameters():
        if prm.grad is not None:
            save[f"grad::{k}"] = prm.grad.numpy()
    with torch.no_grad():
        frames = model.generate_many(_t(batch["inputs"][0]), _t(batch["case_params"][0]), steps, _t(batch["mask"][0]))
    save["frames"] = np.stack([f.numpy() for f in frames])
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()}, save["frames"].shape)


def gen_nonauto(name, kind, seed, B, K, H, W, width, act_name, act_norm, p=5):
    """Non-autoregressive DeepONet / FfnModel (src/models/deeponet.py, ffn.py:38-181) with given query points:
    forward, loss, backward, generate_one from the reference modules."""
    from models.deeponet import DeepONet  # reference
    from models.ffn import FfnModel  # reference
    torch.manual_seed(seed)
    if kind == "deeponet":
        model = DeepONet(p, 3, MseLoss(normalize=True), branch_depth=3, trunk_depth=3, width=width, act_name=act_name,
                         act_norm=act_norm)
    else:
        model = FfnModel(MseLoss(normalize=True), [p + 3, width, width, 1], act_name=act_name, act_norm=act_norm)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k == "bias":
                v.fill_(0.03)
    rng = np.random.default_rng(seed)
    cp = rng.standard_normal((B, p)).astype(np.float32)
    t = rng.uniform(0, 2, (B, 1)).astype(np.float32)
    label = rng.standard_normal((B, 2, H, W)).astype(np.float32)
    q = np.stack([rng.integers(0, H, K), rng.integers(0, W, K)], axis=-1).astype(np.int64)
    out = model(case_params=_t(cp), t=_t(t), label=_t(label), query_idxs=_t(q))
    out["loss"]["nmse"].backward()
    save = dict(meta=np.array([seed, B, K, H, W, width, p, int(act_norm)]), act=np.array(act_name), kind=np.array(kind),
                cp=cp, t=t, label=label, q=q, preds=out["preds"].detach().numpy(),
                **{f"loss_{k}": np.array(v.item()) for k, v in out["loss"].items()})
    for k, v in model.state_dict().items():
        save[f"sd::{k}"] = v.detach().numpy()
    for k, prm in model.named_parameters():
        save[f"grad::{k}"] = prm.grad.numpy()
    with torch.no_grad():
        save["frame"] = model.generate_one(_t(cp[0]), _t(t[0]), 6, 7).numpy()
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", {k: float(v) for k, v in out["loss"].items()})


def gen_adam(name, pseed, bseed, B, C, L, H, W, nsteps, lr, p=5, gain=1.0):
    """train_auto.py:231-257: model(**batch) -> loss['nmse'].backward() -> Adam.step() -> zero_grad()."""
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=gain)
    model = build_ref_model(params, C, L, 12, 12, p)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    losses = []
    for s in range(nsteps):
        batch = synth.make_batch(bseed + s, B, H, W, p)
        out = model(**{k: _t(v) for k, v in batch.items()})
        opt.zero_grad()
        out["loss"]["nmse"].backward()
        opt.step()
        losses.append(out["loss"]["nmse"].item())
    save = dict(meta=np.array([pseed, bseed, B, C, L, H, W, p, nsteps]), lr=np.array(lr), gain=np.array(gain),
                losses=np.array(losses))
    for k, v in model.state_dict().items():
        save[f"param::{k}"] = v.numpy()
    np.savez_compressed(OUT / f"{name}.npz", **save)
    print(name, "ok", losses)


def gen_cavity_dataset(name, seed=5):
    """Split, preprocessing and item conventions of the reference's cavity loaders (src/dataset/cavity.py) on the
    deterministic synthetic tree of oracle/synth.write_cavity_tree: everything the native loader must reproduce."""
    import hashlib
    import json
    import tempfile
    from dataset.cavity import get_cavity_auto_datasets, get_cavity_datasets  # reference
    digest = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    out = dict(seed=seed, auto={}, nonauto={})
    with tempfile.TemporaryDirectory() as tmp:
        root = synth.write_cavity_tree(tmp, seed)
        for subset, dt, np_, nb in (("prop_bc_geo", 0.1, True, True), ("prop_bc_geo", 0.2, False, True), ("prop_bc", 0.1, True, False)):
            splits = get_cavity_auto_datasets(root / "cavity", subset, norm_props=np_, norm_bc=nb, delta_time=dt)
            rec = []
            for ds in splits:
                i0, l0, cp0 = ds[len(ds) // 2]
                rec.append(dict(cases=[f"{d.parent.name}/{d.name}" for d in ds.case_dirs], n=len(ds),
                                case_ids=[int(v) for v in ds.case_ids], inputs=digest(ds.inputs.numpy()),
                                labels=digest(ds.labels.numpy()), case_params=ds.case_params,
                                mid_item=dict(inp=digest(i0.numpy()), lab=digest(l0.numpy()),
                                              cp={k: float(v) for k, v in cp0.items()}),
                                n_features=[int(f.shape[0]) for f in ds.all_features]))
            out["auto"][f"{subset}|{dt}|{int(np_)}|{int(nb)}"] = rec
        splits = get_cavity_datasets(root / "cavity", "prop_geo", norm_props=True, norm_bc=True)
        rec = []
        for ds in splits:
            cp, t, frame = ds[len(ds) - 2]
            rec.append(dict(cases=[f"{d.parent.name}/{d.name}" for d in ds.case_dirs], n=len(ds),
                            num_frames_before=[int(v) for v in ds.num_frames_before], num_features=int(ds.num_features),
                            case_params=[[float(x) for x in c] for c in ds.case_params],
                            item=dict(cp=[float(x) for x in cp], t=float(t[0]), frame=digest(frame.numpy()))))
        out["nonauto"]["prop_geo"] = rec
    with open(OUT / f"{name}.json", "w", encoding="utf8") as f:
        json.dump(out, f, indent=1)
    print(name, "ok")


def gen_flow_dataset(name, problem, seed=5):
    """As gen_cavity_dataset for the tube / dam loaders of the reference (src/dataset/tube.py, dam.py)."""
    import hashlib
    import importlib
    import json
    import tempfile
    mod = importlib.import_module(f"dataset.{problem}")  # reference
    get_auto, get_nonauto = getattr(mod, f"get_{problem}_auto_datasets"), getattr(mod, f"get_{problem}_datasets")
    digest = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    out = dict(seed=seed, problem=problem, auto={}, nonauto={})
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # the cylinder loader of the reference writes a tensor cache under ./dataset/cache
        root = synth.write_flow_tree(tmp, problem, seed)
        scale = 0.01 if problem == "cylinder" else 1.0  # its autoregressive class assumes 1-ms frames
        for subset, dt, np_, nb in (("prop_bc_geo", 0.1, True, True), ("prop_geo", 0.2, False, True), ("prop_bc", 0.1, True, False)):
            dt = round(dt * scale, 6)
            splits = get_auto(root / problem, subset, norm_props=np_, norm_bc=nb, delta_time=dt)
            rec = []
            for ds in splits:
                i0, l0, cp0 = ds[len(ds) // 2]
                rec.append(dict(cases=[f"{d.parent.name}/{d.name}" for d in ds.case_dirs], n=len(ds),
                                case_ids=[int(v) for v in ds.case_ids], inputs=digest(ds.inputs.numpy()),
                                labels=digest(ds.labels.numpy()), case_params=ds.case_params,
                                shape=list(ds.inputs.shape[1:]),
                                mid_item=dict(inp=digest(i0.numpy()), lab=digest(l0.numpy()),
                                              cp={k: float(v) for k, v in cp0.items()}),
                                n_features=[int(f.shape[0]) for f in ds.all_features]))
            out["auto"][f"{subset}|{dt}|{int(np_)}|{int(nb)}"] = rec
        splits = get_nonauto(root / problem, "prop_geo", norm_props=True, norm_bc=True)
        rec = []
        for ds in splits:
            cp, t, frame = ds[len(ds) - 2]
            rec.append(dict(cases=[f"{d.parent.name}/{d.name}" for d in ds.case_dirs], n=len(ds),
                            num_frames_before=[int(v) for v in ds.num_frames_before], num_features=int(ds.num_features),
                            case_params=[[float(x) for x in c] for c in ds.case_params],
                            has_all_features=hasattr(ds, "all_features") and ds.all_features is not None,
                            item=dict(cp=[float(x) for x in cp], t=float(t[0]), frame=digest(frame.numpy()))))
        out["nonauto"]["prop_geo"] = rec
        os.chdir(cwd)
    with open(OUT / f"{name}.json", "w", encoding="utf8") as f:
        json.dump(out, f, indent=1)
    print(name, "ok")


def gen_mseloss(name, seed):
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((3, 2, 17, 19)).astype(np.float32)
    l = rng.standard_normal((3, 2, 17, 19)).astype(np.float32)
    r = MseLoss(normalize=True)(preds=_t(p), labels=_t(l))
    np.savez_compressed(OUT / f"{name}.npz", meta=np.array([seed]), **{k: v.numpy() for k, v in r.items()})
    print(name, "ok")


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    if len(sys.argv) > 1 and sys.argv[1] == "--only-unet-bilinear":  # added in round 2; the other fixtures are unchanged
        gen_unet("unet_bilinear_dim4_32x48", 64, 74, 2, 32, 48, 4, p=5, bilinear=True)
        return
    gen_spectral("spectral_64x64", 11, 2, 3, 4, 64, 64, 12, 12)
    gen_spectral("spectral_66x65", 12, 2, 4, 3, 66, 65, 12, 12)
    gen_spectral("spectral_c20_64x64", 13, 1, 20, 20, 64, 64, 12, 12)
    gen_fno("fno_small_64x64", 21, 31, 2, 8, 2, 64, 64, gain=8.0)
    gen_fno("fno_small_66x65", 22, 32, 2, 6, 2, 66, 65, border=True, gain=6.0)
    gen_fno("fno_cfg1_b8", 23, 33, 8, 20, 4, 64, 64, full_grads=False)
    gen_rollout("rollout_small_64x64", 24, 34, 2, 8, 2, 64, 64, steps=6, gain=8.0)
    gen_rollout("rollout_small_66x65", 25, 35, 1, 6, 2, 66, 65, steps=4, border=True, gain=6.0)
    gen_auto_deeponet("auto_deeponet_small_16x16", 41, 51, 3, 16, 16, 24, 3, 3, "relu")
    gen_auto_deeponet("auto_deeponet_tanh_18x17", 42, 52, 2, 18, 17, 20, 2, 3, "tanh")
    gen_auto_deeponet("auto_deeponet_gelu_16x16", 43, 53, 2, 16, 16, 16, 2, 2, "gelu")
    gen_unet("unet_dim4_32x32", 61, 71, 3, 32, 32, 4)
    gen_unet("unet_dim3_36x40", 62, 72, 2, 36, 40, 3, p=5)
    gen_unet("unet_hidden_dim2_32x32", 63, 73, 3, 32, 32, 2, p=5, insert="hidden")
    gen_unet("unet_bilinear_dim4_32x48", 64, 74, 2, 32, 48, 4, p=5, bilinear=True)
    gen_resnet("resnet_h4_20x24", 81, 91, 2, 20, 24, 4, 1)
    gen_nonauto("deeponet_normact_relu", "deeponet", 101, 3, 37, 16, 18, 24, "relu", True)
    gen_nonauto("deeponet_plain_tanh", "deeponet", 102, 2, 50, 16, 18, 20, "tanh", False)
    gen_nonauto("ffnmodel_normact_gelu", "ffn", 103, 3, 41, 16, 18, 24, "gelu", True)
    gen_auto_variants("auto_edeeponet_relu_16x18", "auto_edeeponet", 111, 121, 3, 16, 18, 24, 3, "relu")
    gen_auto_variants("auto_edeeponet_gelu_12x12", "auto_edeeponet", 112, 122, 2, 12, 12, 16, 2, "gelu")
    gen_auto_variants("auto_ffn_relu_16x18", "auto_ffn", 113, 123, 3, 16, 18, 24, 3, "relu")
    gen_auto_variants("auto_ffn_tanh_10x12", "auto_ffn", 114, 124, 4, 10, 12, 16, 2, "tanh", nq=30)
    gen_auto_deeponet_cnn("auto_deeponet_cnn_64x64", 115, 125, 2)
    gen_adam("adam_small_64x64", 26, 36, 2, 8, 2, 64, 64, nsteps=3, lr=1e-3, gain=8.0)
    gen_mseloss("mseloss", 41)
    gen_cavity_dataset("cavity_dataset")
    gen_flow_dataset("tube_dataset", "tube")
    gen_flow_dataset("dam_dataset", "dam")
    gen_flow_dataset("cylinder_dataset", "cylinder")


if __name__ == "__main__":
    main()
