"""NumPy restatement of the reference's Auto-FNO hot path (forward, backward, loss,
Adam, rollout).  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Every function cites the reference lines (relative to /root/reference) it restates.
All functions are dtype-generic: pass float64 arrays for a ground-truth run, float32
arrays to mimic the reference's fp32 CPU arithmetic.

Two independent statements of SpectralConv2d are kept on purpose:
  * ``spectral_conv2d_fwd``      -- literal rfft2 -> einsum -> zero-filled spectrum -> irfft2
                                    (src/models/fno/fno2d.py:59-82), via numpy's pocketfft.
  * ``spectral_conv2d_fwd_dft``  -- the pruned-DFT closed form that the HIP
                                    kernels implement; ``spectral_conv2d_bwd`` is its adjoint.
tests/test_oracle_golden.py checks both against outputs of the imported reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

Array = np.ndarray


# --------------------------------------------------------------------------------------
# SpectralConv2d_fast  (src/models/fno/fno2d.py:17-82)
# --------------------------------------------------------------------------------------
def _cdtype(x: Array):
    return np.complex128 if x.dtype == np.float64 else np.complex64


def compl_mul2d(inp: Array, weights: Array) -> Array:
    """(b,i,x,y),(i,o,x,y)->(b,o,x,y).  src/models/fno/fno2d.py:54-57."""
    return np.einsum("bixy,ioxy->boxy", inp, weights)


def spectral_conv2d_fwd(x: Array, w1: Array, w2: Array) -> Array:
    """Literal restatement of SpectralConv2d_fast.forward, src/models/fno/fno2d.py:59-82."""
    B, Cin, H, W = x.shape
    Cout, m1, m2 = w1.shape[1], w1.shape[2], w1.shape[3]
    x_ft = np.fft.rfft2(x)  # fno2d.py:62
    out_ft = np.zeros((B, Cout, H, W // 2 + 1), dtype=_cdtype(x))  # :65-72
    out_ft[:, :, :m1, :m2] = compl_mul2d(x_ft[:, :, :m1, :m2], w1)  # :73-75
    out_ft[:, :, H - m1:, :m2] = compl_mul2d(x_ft[:, :, H - m1:, :m2], w2)  # :76-78
    y = np.fft.irfft2(out_ft, s=(H, W))  # :81
    return y.astype(x.dtype, copy=False)


def kept_rows(H: int, m1: int) -> Array:
    """The 2*m1 kx rows ever consumed: [0,m1) and [H-m1,H) (fno2d.py:73-78)."""
    return np.concatenate([np.arange(m1), np.arange(H - m1, H)])


def pruned_dft_fwd(x: Array, m1: int, m2: int) -> Array:
    """X^[b,c,k,l] = sum_{x,y} x[x,y] e^{-2 pi i (k x/H + l y/W)}, k in kept_rows, l < m2.

    Equals rfft2(x)[..., kept_rows, :m2] (fno2d.py:62,74,77)."""
    H, W = x.shape[-2:]
    rd = np.float64 if x.dtype == np.float64 else np.float32
    k = kept_rows(H, m1)[:, None].astype(np.float64)
    Fh = np.exp(-2j * np.pi * k * np.arange(H)[None, :] / H)  # (2m1,H)
    Fw = np.exp(-2j * np.pi * np.arange(W)[:, None] * np.arange(m2)[None, :] / W)  # (W,m2)
    Fh = Fh.astype(_cdtype(x))
    Fw = Fw.astype(_cdtype(x))
    del rd
    return np.einsum("kx,bcxy,yl->bckl", Fh, x, Fw, optimize=True)


def hermitian_weights(m2: int, W: int, dtype=np.float64) -> Array:
    """c_l of the C2R transform: 1 for l=0 (and the Nyquist column), 2 otherwise."""
    c = np.full(m2, 2.0, dtype=dtype)
    c[0] = 1.0
    if W % 2 == 0 and m2 > W // 2:
        c[W // 2] = 1.0
    return c


def pruned_idft(Z: Array, H: int, W: int, scale: bool = True) -> Array:
    """y[x,y] = s * Re sum_{k,l} c_l Z[k,l] e^{+2 pi i (k x/H + l y/W)}  with s = 1/(HW) if
    ``scale`` (irfft2 of the zero-filled spectrum, fno2d.py:65-81) else s = 1 and c_l = 1
    (the adjoint needed by the backward pass)."""
    m1 = Z.shape[-2] // 2
    m2 = Z.shape[-1]
    real = np.float64 if Z.dtype == np.complex128 else np.float32
    k = kept_rows(H, m1)[None, :].astype(np.float64)
    Gh = np.exp(2j * np.pi * np.arange(H)[:, None] * k / H).astype(Z.dtype)  # (H,2m1)
    Gw = np.exp(2j * np.pi * np.arange(m2)[:, None] * np.arange(W)[None, :] / W)  # (m2,W)
    if scale:
        Gw = Gw * hermitian_weights(m2, W)[:, None] / (H * W)
    Gw = Gw.astype(Z.dtype)
    y = np.einsum("xk,bckl,ly->bcxy", Gh, Z, Gw, optimize=True)
    return np.ascontiguousarray(y.real).astype(real, copy=False)


def mode_mix(Xh: Array, w1: Array, w2: Array) -> Array:
    """Z[:, :, :m1] = X^[:, :, :m1] * w1 ; Z[:, :, m1:] = X^[:, :, m1:] * w2 (fno2d.py:73-78)."""
    m1 = w1.shape[2]
    return np.concatenate(
        [compl_mul2d(Xh[:, :, :m1], w1), compl_mul2d(Xh[:, :, m1:], w2)], axis=2
    )


def spectral_conv2d_fwd_dft(x: Array, w1: Array, w2: Array) -> Array:
    """Closed form of fno2d.py:59-82 on the 2*m1 x m2 kept modes only."""
    H, W = x.shape[-2:]
    m1, m2 = w1.shape[2], w1.shape[3]
    return pruned_idft(mode_mix(pruned_dft_fwd(x, m1, m2), w1, w2), H, W)


def spectral_conv2d_bwd(gy: Array, x: Array, w1: Array, w2: Array) -> Tuple[Array, Array, Array]:
    """Adjoint of fno2d.py:59-82 (what autograd produces for x, weights1, weights2).

    G^ = pruned_dft(gy); gZ = (c_l/HW) G^; gW[i,o] = sum_b conj(X^[b,i]) gZ[b,o] (torch's complex
    gradient convention dL/dRe + i dL/dIm); gX^[b,i] = sum_o conj(W[i,o]) gZ[b,o];
    gx = Re sum gX^ e^{+...} (no c_l, no 1/HW)."""
    H, W = x.shape[-2:]
    m1, m2 = w1.shape[2], w1.shape[3]
    Xh = pruned_dft_fwd(x, m1, m2)
    Gh = pruned_dft_fwd(gy, m1, m2)
    c = hermitian_weights(m2, W).astype(x.dtype)
    gZ = Gh * (c / (H * W))
    gw1 = np.einsum("bixy,boxy->ioxy", np.conj(Xh[:, :, :m1]), gZ[:, :, :m1])
    gw2 = np.einsum("bixy,boxy->ioxy", np.conj(Xh[:, :, m1:]), gZ[:, :, m1:])
    gXh = np.concatenate(
        [
            np.einsum("ioxy,boxy->bixy", np.conj(w1), gZ[:, :, :m1]),
            np.einsum("ioxy,boxy->bixy", np.conj(w2), gZ[:, :, m1:]),
        ],
        axis=2,
    )
    gx = pruned_idft(gXh, H, W, scale=False)
    return gx, gw1, gw2


# --------------------------------------------------------------------------------------
# Pointwise pieces
# --------------------------------------------------------------------------------------
_SQRT1_2 = 1.0 / math.sqrt(2.0)
_INV_SQRT_2PI = 1.0 / math.sqrt(2.0 * math.pi)


def gelu(x: Array) -> Array:
    """nn.GELU() (exact erf form), src/models/fno/fno2d.py:147."""
    return (0.5 * x * (1.0 + _erf(x * _SQRT1_2))).astype(x.dtype, copy=False)


def gelu_grad(x: Array) -> Array:
    """d gelu / dx = Phi(x) + x phi(x)."""
    return (0.5 * (1.0 + _erf(x * _SQRT1_2)) + x * np.exp(-0.5 * x * x) * _INV_SQRT_2PI).astype(
        x.dtype, copy=False
    )


def conv1x1(x: Array, w: Array, b: Array) -> Array:
    """nn.Conv2d(cin, cout, 1): w (cout,cin,1,1) or (cout,cin).  fno2d.py:104,150,175,176."""
    w2 = w.reshape(w.shape[0], w.shape[1])
    return np.einsum("oi,bihw->bohw", w2, x, optimize=True) + b[None, :, None, None]


def get_coords(H: int, W: int, dtype) -> Tuple[Array, Array]:
    """grid_x varies along rows (linspace over H), grid_y along cols.  fno2d.py:244-255
    (the reference casts np.linspace's float64 to float32)."""
    gx = np.linspace(0, 1, H).astype(np.float32).astype(dtype)
    gy = np.linspace(0, 1, W).astype(np.float32).astype(dtype)
    return gx, gy


def assemble_features(inputs: Array, case_params: Array, mask: Array) -> Array:
    """[u, v, mask, grid_x, grid_y, props...]  fno2d.py:197-214."""
    B, _, H, W = inputs.shape
    gx, gy = get_coords(H, W, inputs.dtype)
    grid_x = np.broadcast_to(gx[None, None, :, None], (B, 1, H, W))
    grid_y = np.broadcast_to(gy[None, None, None, :], (B, 1, H, W))
    props = np.broadcast_to(case_params[:, :, None, None], (B, case_params.shape[1], H, W))
    return np.concatenate([inputs, mask, grid_x, grid_y, props], axis=1)


# --------------------------------------------------------------------------------------
# MseLoss  (src/models/loss.py:22-37)
# --------------------------------------------------------------------------------------
def mse_loss(preds: Array, labels: Array, normalize: bool = True) -> Dict[str, float]:
    d = preds - labels
    mse = np.mean(d * d)  # loss.py:27
    mae = np.mean(np.abs(d))  # loss.py:28
    out = dict(mse=mse, rmse=np.sqrt(mse), mae=mae)  # loss.py:29-33
    if normalize:
        out["nmse"] = mse / np.mean(labels * labels)  # loss.py:35
    return out


# --------------------------------------------------------------------------------------
# Fno2d forward / backward  (src/models/fno/fno2d.py:178-242)
# --------------------------------------------------------------------------------------
def bf16_round(x: Array) -> Array:
    """Round to the nearest bfloat16 (ties to even) and return in x's dtype: what storing an activation as bf16 does."""
    f = np.ascontiguousarray(x, dtype=np.float32)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(x.dtype)


def fno_forward(
    params: Dict[str, Array],
    inputs: Array,
    case_params: Array,
    mask: Optional[Array] = None,
    label: Optional[Array] = None,
    num_layers: int = 4,
    normalize: bool = True,
    use_fft: bool = True,
    keep_cache: bool = True,
    act_store=None,
) -> Dict:
    """``params`` uses the reference's state_dict key names (SURVEY.md 8b Checkpoint ABI).
    ``act_store`` (e.g. ``bf16_round``) is applied to every activation that the bf16-storage inference path keeps in
    memory between kernels: the lifting layer's output and each FnoBlock's pre-activation (BASELINE configs[4]; the
    reference itself has no reduced-precision path)."""
    if act_store is None:
        act_store = lambda a: a  # noqa: E731
    B, _, H, W = inputs.shape
    dt = inputs.dtype
    if mask is None:
        mask = np.ones((B, 1, H, W), dtype=dt)  # fno2d.py:189-191
    elif mask.ndim == 3:
        mask = mask[:, None]  # :193-194
    feats = assemble_features(inputs, case_params, mask)  # :195-214
    spec = spectral_conv2d_fwd if use_fft else spectral_conv2d_fwd_dft
    acts: List[Array] = []  # block inputs h_l
    pres: List[Array] = []  # pre-activations of block l
    h = act_store(conv1x1(feats, params["fc0.weight"], params["fc0.bias"]).astype(dt))  # :217
    for l in range(num_layers):  # :223, FnoBlock.forward :106-112
        acts.append(h)
        pre = spec(h, params[f"blocks.{l}.conv0.weights1"], params[f"blocks.{l}.conv0.weights2"]) + conv1x1(
            h, params[f"blocks.{l}.w0.weight"], params[f"blocks.{l}.w0.bias"]
        )
        pre = act_store(pre.astype(dt))
        pres.append(pre)
        h = gelu(pre)
    z1 = conv1x1(h, params["fc1.weight"], params["fc1.bias"]).astype(dt)  # :228
    a1 = gelu(z1)  # :229
    raw = conv1x1(a1, params["fc2.weight"], params["fc2.bias"]).astype(dt)  # :230
    preds = raw * mask  # :233
    out: Dict = dict(preds=preds)
    if label is not None:
        lab = label * mask  # :236
        out["loss"] = mse_loss(preds, lab, normalize)  # :237
    if keep_cache:
        out["cache"] = dict(feats=feats, acts=acts, pres=pres, hL=h, z1=z1, a1=a1, mask=mask,
                            label=None if label is None else label * mask, preds=preds)
    return out


def loss_grad_wrt_preds(preds: Array, lab: Array, loss_name: str = "nmse") -> Array:
    """d loss / d preds for loss in {mse, nmse, mae} (loss.py:27-35)."""
    n = preds.size
    d = preds - lab
    if loss_name == "mse":
        return 2.0 * d / n
    if loss_name == "nmse":
        return 2.0 * d / np.sum(lab * lab)
    if loss_name == "mae":
        return np.sign(d) / n
    raise ValueError(loss_name)


def fno_backward(
    params: Dict[str, Array], cache: Dict, gpreds: Array, num_layers: int = 4
) -> Dict[str, Array]:
    """Manual reverse pass of ``fno_forward`` (what ``loss.backward()`` at train_auto.py:255 computes).
    Returns gradients keyed like ``params`` (complex weights: torch convention)."""
    g: Dict[str, Array] = {}
    mask = cache["mask"]
    graw = gpreds * mask
    a1, z1, hL = cache["a1"], cache["z1"], cache["hL"]
    w2 = params["fc2.weight"].reshape(params["fc2.weight"].shape[0], -1)
    g["fc2.weight"] = np.einsum("bohw,bihw->oi", graw, a1, optimize=True).reshape(params["fc2.weight"].shape)
    g["fc2.bias"] = graw.sum(axis=(0, 2, 3))
    ga1 = np.einsum("oi,bohw->bihw", w2, graw, optimize=True)
    gz1 = ga1 * gelu_grad(z1)
    w1 = params["fc1.weight"].reshape(params["fc1.weight"].shape[0], -1)
    g["fc1.weight"] = np.einsum("bohw,bihw->oi", gz1, hL, optimize=True).reshape(params["fc1.weight"].shape)
    g["fc1.bias"] = gz1.sum(axis=(0, 2, 3))
    gh = np.einsum("oi,bohw->bihw", w1, gz1, optimize=True)
    for l in reversed(range(num_layers)):
        gpre = gh * gelu_grad(cache["pres"][l])
        h_in = cache["acts"][l]
        kw = f"blocks.{l}.w0.weight"
        w0 = params[kw].reshape(params[kw].shape[0], -1)
        g[kw] = np.einsum("bohw,bihw->oi", gpre, h_in, optimize=True).reshape(params[kw].shape)
        g[f"blocks.{l}.w0.bias"] = gpre.sum(axis=(0, 2, 3))
        gx_s, gw1, gw2 = spectral_conv2d_bwd(
            gpre, h_in, params[f"blocks.{l}.conv0.weights1"], params[f"blocks.{l}.conv0.weights2"]
        )
        g[f"blocks.{l}.conv0.weights1"] = gw1
        g[f"blocks.{l}.conv0.weights2"] = gw2
        gh = gx_s + np.einsum("oi,bohw->bihw", w0, gpre, optimize=True)
    feats = cache["feats"]
    g["fc0.weight"] = np.einsum("bohw,bihw->oi", gh, feats, optimize=True).reshape(params["fc0.weight"].shape)
    g["fc0.bias"] = gh.sum(axis=(0, 2, 3))
    # gradient w.r.t. the u,v input channels (needed when rollouts are differentiated; also a parity probe)
    w_fc0 = params["fc0.weight"].reshape(params["fc0.weight"].shape[0], -1)
    g["__inputs__"] = np.einsum("oi,bohw->bihw", w_fc0[:, :2], gh, optimize=True)
    return g


# --------------------------------------------------------------------------------------
# torch.optim.Adam  (train_auto.py:213,256; complex params via view_as_real)
# --------------------------------------------------------------------------------------
def adam_step(
    p: Array, grad: Array, m: Array, v: Array, step: int, lr: float,
    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0,
) -> None:
    """In-place single-tensor Adam on REAL views (complex tensors: pass ``x.view(float)``)."""
    if weight_decay:
        grad = grad + weight_decay * p
    m *= beta1
    m += (1 - beta1) * grad
    v *= beta2
    v += (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    p -= (lr / bc1) * m / denom


# --------------------------------------------------------------------------------------
# generate / generate_many  (src/models/fno/fno2d.py:257-295)
# --------------------------------------------------------------------------------------
def generate_many(
    params: Dict[str, Array], inputs: Array, case_params: Array, mask: Array, steps: int,
    num_layers: int = 4, use_fft: bool = True, act_store=None,
) -> List[Array]:
    assert inputs.ndim == case_params.ndim + 2  # fno2d.py:280
    if inputs.ndim == 3:  # :281-285
        inputs, case_params, mask = inputs[None], case_params[None], mask[None]
    cur = inputs
    out = []
    for _ in range(steps):  # :290-294
        cur = fno_forward(params, cur, case_params, mask, None, num_layers, use_fft=use_fft,
                          keep_cache=False, act_store=act_store)["preds"]
        out.append(cur)
    return out


def rel_nmse(a: Array, ref: Array) -> float:
    """Parity metric of north_star: mean((a-ref)^2)/mean(ref^2)."""
    a = np.asarray(a)
    ref = np.asarray(ref)
    num = np.mean(np.abs(a - ref) ** 2)
    den = np.mean(np.abs(ref) ** 2)
    return float(num / den) if den > 0 else float(num)
