"""PyTorch-CPU restatement of the reference's Auto-FNO train step, used ONLY as bench.py's ``cpu_baseline`` leg
(kind "port") and in tests.  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

It issues the same ATen calls as the reference (torch.fft.rfft2 / einsum / irfft2, 1x1 convs as conv2d, exact GELU,
F.mse_loss / F.l1_loss, torch.optim.Adam), so timing it on the GPU box's host cores is timing the reference's CPU path
(/root/reference itself does not exist there).  Each function cites the lines it follows.
"""
from __future__ import annotations

import time
from typing import Dict, List

import torch
import torch.nn.functional as F


def init_params(C: int = 20, L: int = 4, m1: int = 12, m2: int = 12, p: int = 5, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Reference init distributions: Conv2d default (kaiming-uniform) and scale*U[0,1) complex (fno2d.py:31-51,150-176)."""
    g = torch.Generator().manual_seed(seed)
    prm: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci):
        bound = 1.0 / (ci ** 0.5)
        prm[name + ".weight"] = (torch.rand(co, ci, 1, 1, generator=g) * 2 - 1) * bound
        prm[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * bound

    conv("fc0", C, 2 + 3 + p)
    for l in range(L):
        sc = 1.0 / (C * C)
        for w in ("weights1", "weights2"):
            prm[f"blocks.{l}.conv0.{w}"] = torch.complex(sc * torch.rand(C, C, m1, m2, generator=g),
                                                         sc * torch.rand(C, C, m1, m2, generator=g))
        conv(f"blocks.{l}.w0", C, C)
    conv("fc1", 128, C)
    conv("fc2", 2, 128)
    return {k: v.requires_grad_(True) for k, v in prm.items()}


def spectral_conv(x, w1, w2):
    """fno2d.py:59-82."""
    B, _, H, W = x.shape
    m1, m2 = w1.shape[2], w1.shape[3]
    x_ft = torch.fft.rfft2(x)
    out_ft = torch.zeros(B, w1.shape[1], H, W // 2 + 1, dtype=torch.cfloat)
    out_ft[:, :, :m1, :m2] = torch.einsum("bixy,ioxy->boxy", x_ft[:, :, :m1, :m2], w1)
    out_ft[:, :, -m1:, :m2] = torch.einsum("bixy,ioxy->boxy", x_ft[:, :, -m1:, :m2], w2)
    return torch.fft.irfft2(out_ft, s=(H, W))


def forward(prm, inputs, case_params, mask, label, L: int):
    """fno2d.py:178-242 + loss.py:22-37."""
    B, _, H, W = inputs.shape
    gx = torch.linspace(0, 1, H).reshape(1, 1, H, 1).expand(B, 1, H, W)
    gy = torch.linspace(0, 1, W).reshape(1, 1, 1, W).expand(B, 1, H, W)
    props = case_params[:, :, None, None].expand(B, case_params.shape[1], H, W)
    h = torch.cat([inputs, mask, gx, gy, props], dim=1)
    h = F.conv2d(h, prm["fc0.weight"], prm["fc0.bias"])
    for l in range(L):
        h = F.gelu(spectral_conv(h, prm[f"blocks.{l}.conv0.weights1"], prm[f"blocks.{l}.conv0.weights2"])
                   + F.conv2d(h, prm[f"blocks.{l}.w0.weight"], prm[f"blocks.{l}.w0.bias"]))
    h = F.gelu(F.conv2d(h, prm["fc1.weight"], prm["fc1.bias"]))
    preds = F.conv2d(h, prm["fc2.weight"], prm["fc2.bias"]) * mask
    lab = label * mask
    mse = F.mse_loss(preds, lab)
    return preds, dict(mse=mse, mae=F.l1_loss(preds, lab), nmse=mse / torch.square(lab).mean())


def _summary(times: List[float], B: int, steps: int) -> Dict:
    ts = sorted(times)
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    return dict(frames_per_s=B / med, median_s=med, min_s=ts[0], max_s=ts[-1], steps=steps, batch=B, threads=torch.get_num_threads())


def _synthetic(B: int, H: int, W: int, p: int):
    g = torch.Generator().manual_seed(1234)
    inputs = torch.randn(B, 2, H, W, generator=g)
    label = inputs + 0.1 * torch.randn(B, 2, H, W, generator=g)
    return inputs, label, torch.randn(B, p, generator=g), torch.ones(B, 1, H, W)


def time_train_steps(B: int, steps: int, warmup: int = 1, C: int = 20, L: int = 4, H: int = 64, W: int = 64,
                     p: int = 5, threads: int | None = None) -> Dict:
    """train_auto.py:231-257 on synthetic data: forward -> loss['nmse'].backward() -> Adam.step() -> zero_grad().  Returns the
    median (and min / max) of ``steps`` timed steps after ``warmup`` untimed ones (3 + >= 10)."""
    if threads:
        torch.set_num_threads(threads)
    prm = init_params(C, L, 12, 12, p)
    opt = torch.optim.Adam(list(prm.values()), lr=1e-3)
    inputs, label, cp, mask = _synthetic(B, H, W, p)
    times: List[float] = []
    for s in range(warmup + steps):
        t0 = time.perf_counter()
        _, loss = forward(prm, inputs, cp, mask, label, L)
        loss["nmse"].backward()
        opt.step()
        opt.zero_grad()
        if s >= warmup:
            times.append(time.perf_counter() - t0)
    return _summary(times, B, steps)


def cpu_model() -> str:
    """The host CPU's model string (/proc/cpuinfo) -- SURVEY 8(d): state the core count AND the CPU model beside the baseline."""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


# The reference's own modules are timed in a CHILD process: they are untrusted code whose top-level packages (models, utils, args,
# dataset, train) would otherwise shadow anything of the same name for the rest of the calling process.  The child
# gets /root/reference/src on ITS sys.path only.  The Python reference cannot travel to the GPU box in any form, so there
# reference_importable() is False and bench.py's cpu_baseline is kind "port" (this file's ATen call sequence).
_REF_CHILD = r"""
import json, sys, time
sys.path.insert(0, sys.argv[1])
a = json.loads(sys.argv[2])
import torch
from models.fno.fno2d import Fno2d
from models.loss import MseLoss
if a["threads"]:
    torch.set_num_threads(a["threads"])
if a["probe"]:
    print(json.dumps({"ok": True})); sys.exit(0)
B, H, W, p = a["B"], a["H"], a["W"], a["p"]
torch.manual_seed(0)
model = Fno2d(2, 2, p, MseLoss(normalize=True), a["L"], 12, 12, a["C"])
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(1234)
inputs = torch.randn(B, 2, H, W, generator=g)
label = inputs + 0.1 * torch.randn(B, 2, H, W, generator=g)
cp = torch.randn(B, p, generator=g)
mask = torch.ones(B, 1, H, W)
times = []
for s in range(a["warmup"] + a["steps"]):
    t0 = time.perf_counter()
    out = model(inputs=inputs, case_params=cp, mask=mask, label=label)
    out["loss"]["nmse"].backward()
    opt.step()
    opt.zero_grad()
    if s >= a["warmup"]:
        times.append(time.perf_counter() - t0)
print(json.dumps({"times": times, "threads": torch.get_num_threads()}))
"""


def _ref_child(root: str, **a):
    import json
    import os
    import subprocess
    import sys
    if not os.path.isdir(root):
        return None
    r = subprocess.run([sys.executable, "-c", _REF_CHILD, root, json.dumps(a)], capture_output=True, text=True, cwd="/tmp")
    if r.returncode != 0:
        return None
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return None


def reference_importable(root: str = "/root/reference/src") -> bool:
    """True when the reference's own Fno2d imports in a child process (the build container; the GPU box has no /root/reference)."""
    return bool(_ref_child(root, probe=True, threads=0))


def time_reference_steps(B: int, steps: int, warmup: int = 1, C: int = 20, L: int = 4, H: int = 64, W: int = 64,
                         p: int = 5, threads: int | None = None, root: str = "/root/reference/src") -> Dict:
    """The SAME loop on the reference's own module (src/models/fno/fno2d.py:Fno2d, src/train_auto.py:231-257) -- SURVEY 8(d)'s CPU
    baseline "kind: reference"; only where /root/reference/src exists, and in a child process (see above)."""
    res = _ref_child(root, probe=False, B=B, steps=steps, warmup=warmup, C=C, L=L, H=H, W=W, p=p, threads=threads or 0)
    if not res:
        raise RuntimeError(f"the reference modules under {root} could not be timed")
    out = _summary(res["times"], B, steps)
    out["threads"] = res["threads"]
    return out
