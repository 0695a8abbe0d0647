"""NumPy restatement of the conv-stack ops of the U-Net / ResNet baselines (src/models/unet.py, src/models/resnet.py) and
of UNet.forward.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  dtype-generic (float64 = ground truth).
Pinned by tests/golden/unet_*.npz generated from the reference's own UNet (oracle/make_golden.py)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from . import fno_oracle as O

Array = np.ndarray


def _pad_rep(x: Array, p: int) -> Array:
    return np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)), mode="edge")


def conv2d(x: Array, w: Array, b: Optional[Array]) -> Array:
    """nn.Conv2d(k, padding=k//2, padding_mode='replicate')  (unet.py:20-27, resnet.py:35-41)."""
    k = w.shape[-1]
    p = k // 2
    xp = _pad_rep(x, p)
    H, W = x.shape[-2:]
    out = np.zeros((x.shape[0], w.shape[0], H, W), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            out += np.einsum("oi,bihw->bohw", w[:, :, ky, kx], xp[:, :, ky:ky + H, kx:kx + W])
    if b is not None:
        out += b[None, :, None, None]
    return out


def conv2d_bwd(g: Array, x: Array, w: Array) -> Tuple[Array, Array, Array]:
    """(gx, gw, gb) of the above: gx = P^T V^T g (fold of the pad ring), gw by correlation with the padded input."""
    k = w.shape[-1]
    p = k // 2
    B, Ci, H, W = x.shape
    xp = _pad_rep(x, p)
    gw = np.zeros_like(w)
    gxp = np.zeros_like(xp)
    for ky in range(k):
        for kx in range(k):
            gw[:, :, ky, kx] = np.einsum("bohw,bihw->oi", g, xp[:, :, ky:ky + H, kx:kx + W])
            gxp[:, :, ky:ky + H, kx:kx + W] += np.einsum("oi,bohw->bihw", w[:, :, ky, kx], g)
    gx = gxp[:, :, p:p + H, p:p + W].copy()
    if p:
        gx[:, :, 0, :] += gxp[:, :, :p, p:p + W].sum(axis=2)
        gx[:, :, -1, :] += gxp[:, :, p + H:, p:p + W].sum(axis=2)
        gx[:, :, :, 0] += gxp[:, :, p:p + H, :p].sum(axis=3)
        gx[:, :, :, -1] += gxp[:, :, p:p + H, p + W:].sum(axis=3)
        gx[:, :, 0, 0] += gxp[:, :, :p, :p].sum(axis=(2, 3))
        gx[:, :, 0, -1] += gxp[:, :, :p, p + W:].sum(axis=(2, 3))
        gx[:, :, -1, 0] += gxp[:, :, p + H:, :p].sum(axis=(2, 3))
        gx[:, :, -1, -1] += gxp[:, :, p + H:, p + W:].sum(axis=(2, 3))
    return gx, gw, g.sum(axis=(0, 2, 3))


def batchnorm(x: Array, gamma: Array, beta: Array, rm: Array, rv: Array, training: bool, eps=1e-5, momentum=0.1, relu=False):
    """nn.BatchNorm2d (+ReLU).  Returns (y, cache, new_running_mean, new_running_var)."""
    if training:
        mean = x.mean(axis=(0, 2, 3))
        var = x.var(axis=(0, 2, 3))
        n = x.size / x.shape[1]
        nrm = (1 - momentum) * rm + momentum * mean
        nrv = (1 - momentum) * rv + momentum * var * n / max(n - 1, 1)
    else:
        mean, var, nrm, nrv = rm, rv, rm, rv
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mean[None, :, None, None]) * rstd[None, :, None, None]
    y = xh * gamma[None, :, None, None] + beta[None, :, None, None]
    if relu:
        y = np.maximum(y, 0)
    return y, dict(xh=xh, rstd=rstd, gamma=gamma, y=y, relu=relu, training=training), nrm, nrv


def batchnorm_bwd(gy: Array, cache: dict):
    gz = gy * (cache["y"] > 0) if cache["relu"] else gy
    xh, rstd, gamma = cache["xh"], cache["rstd"], cache["gamma"]
    gbeta = gz.sum(axis=(0, 2, 3))
    ggamma = (gz * xh).sum(axis=(0, 2, 3))
    n = gz.size / gz.shape[1]
    t = gz
    if cache["training"]:
        t = gz - (gbeta[None, :, None, None] + xh * ggamma[None, :, None, None]) / n
    return t * (gamma * rstd)[None, :, None, None], ggamma, gbeta


def maxpool2(x: Array) -> Array:
    B, C, H, W = x.shape
    xx = x[:, :, :H // 2 * 2, :W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2)
    return xx.max(axis=(3, 5))


def maxpool2_bwd(x: Array, gy: Array) -> Array:
    """Gradient to the first maximum in row-major window order (torch's max_pool2d backward)."""
    B, C, H, W = x.shape
    Ho, Wo = H // 2, W // 2
    win = x[:, :, :Ho * 2, :Wo * 2].reshape(B, C, Ho, 2, Wo, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, Ho, Wo, 4)
    arg = win.argmax(axis=-1)  # first maximum
    gx = np.zeros_like(x)
    for k in range(4):
        gx[:, :, (k // 2):Ho * 2:2, (k % 2):Wo * 2:2] = np.where(arg == k, gy, 0)
    return gx


def convt2(x: Array, w: Array, b: Optional[Array]) -> Array:
    """nn.ConvTranspose2d(kernel_size=2, stride=2), w: (Ci, Co, 2, 2)  (unet.py:80)."""
    B, Ci, H, W = x.shape
    out = np.zeros((B, w.shape[1], 2 * H, 2 * W), dtype=x.dtype)
    for ky in range(2):
        for kx in range(2):
            out[:, :, ky::2, kx::2] = np.einsum("bihw,io->bohw", x, w[:, :, ky, kx])
    if b is not None:
        out += b[None, :, None, None]
    return out


def convt2_bwd(g: Array, x: Array, w: Array):
    gx = np.zeros_like(x)
    gw = np.zeros_like(w)
    for ky in range(2):
        for kx in range(2):
            gs = g[:, :, ky::2, kx::2]
            gx += np.einsum("bohw,io->bihw", gs, w[:, :, ky, kx])
            gw[:, :, ky, kx] = np.einsum("bihw,bohw->io", x, gs)
    return gx, gw, g.sum(axis=(0, 2, 3))


def _lerp_tables(n_in: int, dtype):
    """Source rows and weights of nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) along one axis
    (ATen area_pixel_compute_source_index + compute_source_index_and_lambda; src/models/unet.py:74-76)."""
    n_out = 2 * n_in
    scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
    s = scale * np.arange(n_out, dtype=np.float64)
    i0 = np.minimum(np.floor(s).astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = np.clip(s - i0, 0.0, 1.0).astype(dtype)
    return i0, i1, (1 - l1).astype(dtype), l1


def upsample2_bilinear(x: Array) -> Array:
    B, C, H, W = x.shape
    y0, y1, a0, a1 = _lerp_tables(H, x.dtype)
    x0, x1, b0, b1 = _lerp_tables(W, x.dtype)
    rows = x[:, :, y0, :] * a0[None, None, :, None] + x[:, :, y1, :] * a1[None, None, :, None]
    return rows[:, :, :, x0] * b0[None, None, None, :] + rows[:, :, :, x1] * b1[None, None, None, :]


def upsample2_bilinear_bwd(gy: Array) -> Array:
    """Adjoint of upsample2_bilinear: gy (B,C,2H,2W) -> gx (B,C,H,W)."""
    B, C, Ho, Wo = gy.shape
    H, W = Ho // 2, Wo // 2
    y0, y1, a0, a1 = _lerp_tables(H, gy.dtype)
    x0, x1, b0, b1 = _lerp_tables(W, gy.dtype)
    My = np.zeros((Ho, H), dtype=gy.dtype)
    np.add.at(My, (np.arange(Ho), y0), a0)
    np.add.at(My, (np.arange(Ho), y1), a1)
    Mx = np.zeros((Wo, W), dtype=gy.dtype)
    np.add.at(Mx, (np.arange(Wo), x0), b0)
    np.add.at(Mx, (np.arange(Wo), x1), b1)
    return np.einsum("bcyx,yi,xj->bcij", gy, My, Mx)


# ---- UNet.forward (src/models/unet.py:153-223), insert_case_params_at="input" | "hidden", bilinear False | True ----
def _double_conv(P, pre, x, training, stats):
    for j in (1, 2):
        x = conv2d(x, P[f"{pre}.conv{j}.0.weight"], P[f"{pre}.conv{j}.0.bias"])
        x, _, nrm, nrv = batchnorm(x, P[f"{pre}.conv{j}.1.weight"], P[f"{pre}.conv{j}.1.bias"],
                                   P[f"{pre}.conv{j}.1.running_mean"], P[f"{pre}.conv{j}.1.running_var"], training, relu=True)
        stats[f"{pre}.conv{j}.1.running_mean"], stats[f"{pre}.conv{j}.1.running_var"] = nrm, nrv
    return x


def unet_forward(P: Dict[str, Array], inputs: Array, case_params: Array, mask: Array, label: Optional[Array],
                 out_chan: int = 2, training: bool = False):
    B, _, H, W = inputs.shape
    if mask.ndim == 3:
        mask = mask[:, None]
    hidden = "case_params_fc.weight" in P  # insert_case_params_at == "hidden" (unet.py:132-133, 198-204)
    if hidden:
        x = np.concatenate([inputs, mask], axis=1)
    else:
        x = np.concatenate([inputs, mask, np.broadcast_to(case_params[:, :, None, None], (B, case_params.shape[1], H, W))], axis=1)
    stats: Dict[str, Array] = {}
    x1 = _double_conv(P, "in_conv", x, training, stats)
    skips = [x1]
    cur = x1
    for d in (1, 2, 3, 4):
        cur = _double_conv(P, f"down{d}.maxpool_conv.1", maxpool2(cur), training, stats)
        skips.append(cur)
    if hidden:  # x5 + Linear(case_params)[:, :, None, None]; the skip list keeps the un-conditioned x5 out of reach (it is
        # only ever consumed by up1 as its input, unet.py:206)
        conds = case_params @ P["case_params_fc.weight"].T + P["case_params_fc.bias"]
        cur = cur + conds[:, :, None, None]
    bilinear = "up1.up.weight" not in P  # nn.Upsample holds no parameters (unet.py:74-78)
    for u, skip in zip((1, 2, 3, 4), (skips[3], skips[2], skips[1], skips[0])):
        up = upsample2_bilinear(cur) if bilinear else convt2(cur, P[f"up{u}.up.weight"], P[f"up{u}.up.bias"])
        dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
        up = np.pad(up, ((0, 0), (0, 0), (dy // 2, dy - dy // 2), (dx // 2, dx - dx // 2)))
        cur = _double_conv(P, f"up{u}.conv", np.concatenate([skip, up], axis=1), training, stats)
    w = P["out_conv.conv.weight"]
    preds = np.einsum("oi,bihw->bohw", w[:, :, 0, 0], cur) + P["out_conv.conv.bias"][None, :, None, None]
    preds = (preds + inputs[:, :out_chan]) * mask
    out = dict(preds=preds, running=stats)
    if label is not None:
        out["loss"] = O.mse_loss(preds, label * mask, True)
    return out


# ---- ResNet.forward in eval mode (src/models/resnet.py:70-80,145-198): dropout is the identity, bn1/bn2 are never applied ----
def resnet_forward(P: Dict[str, Array], inputs: Array, case_params: Array, mask: Array, label: Optional[Array],
                   out_chan: int = 2):
    B, _, H, W = inputs.shape
    if mask.ndim == 3:
        mask = mask[:, None]
    x = np.concatenate([inputs, mask, np.broadcast_to(case_params[:, :, None, None], (B, case_params.shape[1], H, W))], axis=1)
    i = 0
    while f"blocks.{i}.conv1.weight" in P:
        pre = f"blocks.{i}"
        res = conv2d(x, P[f"{pre}.res_conv.weight"], P[f"{pre}.res_conv.bias"]) if f"{pre}.res_conv.weight" in P else x
        h = O.gelu(conv2d(x, P[f"{pre}.conv1.weight"], P[f"{pre}.conv1.bias"]))
        x = conv2d(h, P[f"{pre}.conv2.weight"], P[f"{pre}.conv2.bias"]) + res
        i += 1
    preds = (x + inputs[:, :out_chan]) * mask
    out = dict(preds=preds)
    if label is not None:
        out["loss"] = O.mse_loss(preds, label * mask, True)
    return out
