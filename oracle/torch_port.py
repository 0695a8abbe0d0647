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


def time_train_steps(B: int, steps: int, warmup: int = 1, C: int = 20, L: int = 4, H: int = 64, W: int = 64,
                     p: int = 5, threads: int | None = None) -> Dict:
    """train_auto.py:231-257 on synthetic data: forward -> loss['nmse'].backward() -> Adam.step() -> zero_grad()."""
    if threads:
        torch.set_num_threads(threads)
    prm = init_params(C, L, 12, 12, p)
    opt = torch.optim.Adam(list(prm.values()), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    inputs = torch.randn(B, 2, H, W, generator=g)
    label = inputs + 0.1 * torch.randn(B, 2, H, W, generator=g)
    cp = torch.randn(B, p, generator=g)
    mask = torch.ones(B, 1, H, W)
    times: List[float] = []
    for s in range(warmup + steps):
        t0 = time.perf_counter()
        _, loss = forward(prm, inputs, cp, mask, label, L)
        loss["nmse"].backward()
        opt.step()
        opt.zero_grad()
        if s >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(frames_per_s=B / med, median_s=med, steps=steps, batch=B, threads=torch.get_num_threads())


def reference_importable(root: str = "/root/reference/src") -> bool:
    """True when the reference's own Fno2d imports (the build container; the GPU box has no /root/reference)."""
    import os
    import sys
    if not os.path.isdir(root):
        return False
    if root not in sys.path:
        sys.path.insert(0, root)
    try:
        from models.fno.fno2d import Fno2d  # noqa: F401  (reference)
        from models.loss import MseLoss  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        return False


def time_reference_steps(B: int, steps: int, warmup: int = 1, C: int = 20, L: int = 4, H: int = 64, W: int = 64,
                         p: int = 5, threads: int | None = None) -> Dict:
    """The SAME loop on the reference's own module (src/models/fno/fno2d.py:Fno2d, src/train_auto.py:231-257) -- SURVEY 8(d)'s CPU
    baseline "kind: reference"; only where /root/reference/src imports."""
    from models.fno.fno2d import Fno2d  # reference
    from models.loss import MseLoss
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = Fno2d(2, 2, p, MseLoss(normalize=True), L, 12, 12, C)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    inputs = torch.randn(B, 2, H, W, generator=g)
    label = inputs + 0.1 * torch.randn(B, 2, H, W, generator=g)
    cp = torch.randn(B, p, generator=g)
    mask = torch.ones(B, 1, H, W)
    times: List[float] = []
    for s in range(warmup + steps):
        t0 = time.perf_counter()
        out = model(inputs=inputs, case_params=cp, mask=mask, label=label)
        out["loss"]["nmse"].backward()
        opt.step()
        opt.zero_grad()
        if s >= warmup:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return dict(frames_per_s=B / med, median_s=med, steps=steps, batch=B, threads=torch.get_num_threads())
