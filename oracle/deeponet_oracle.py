"""NumPy restatement of the Auto-DeepONet hot path (Ffn stacks, branch x trunk inner product + bias + residual, loss)
and its backward pass.  TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  dtype-generic (float64 = ground truth).
Pinned by tests/golden/auto_deeponet_*.npz, generated from the reference's own modules (oracle/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

from . import fno_oracle as O

Array = np.ndarray


def act(z: Array, name: Optional[str]) -> Array:
    """get_act_fn, src/models/act_fn.py:5-18."""
    if name in (None, "none"):
        return z
    if name == "relu":
        return np.maximum(z, 0)
    if name == "tanh":
        return np.tanh(z)
    if name == "gelu":
        return O.gelu(z)
    if name == "swish":
        return z / (1 + np.exp(-z))
    raise ValueError(name)


def act_grad(z: Array, name: Optional[str]) -> Array:
    if name in (None, "none"):
        return np.ones_like(z)
    if name == "relu":
        return (z > 0).astype(z.dtype)
    if name == "tanh":
        return 1 - np.tanh(z) ** 2
    if name == "gelu":
        return O.gelu_grad(z)
    if name == "swish":
        s = 1 / (1 + np.exp(-z))
        return s * (1 + z * (1 - s))
    raise ValueError(name)


def ffn_layers(params: Dict[str, Array], prefix: str) -> List[Tuple[Array, Array]]:
    """(weight, bias) of ``{prefix}.layers.{0,2,4,...}`` in order (Ffn, src/models/ffn.py:21-31)."""
    out, i = [], 0
    while f"{prefix}.layers.{i}.weight" in params:
        out.append((params[f"{prefix}.layers.{i}.weight"], params[f"{prefix}.layers.{i}.bias"]))
        i += 2
    return out


def ffn_forward(layers, x: Array, act_name: str, act_on_output: bool = False):
    """Linear -> act -> ... -> Linear [-> act]  (ffn.py:21-35).  Returns (y, cache of (input, pre-activation))."""
    cache = []
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        z = x @ w.T + b
        has_act = i < n - 1 or act_on_output
        cache.append((x, z, has_act))
        x = act(z, act_name) if has_act else z
    return x, cache


def ffn_backward(layers, cache, gy: Array, act_name: str):
    grads = []
    g = gy
    for (w, b), (x, z, has_act) in zip(reversed(layers), reversed(cache)):
        gz = g * act_grad(z, act_name) if has_act else g
        x2, gz2 = x.reshape(-1, x.shape[-1]), gz.reshape(-1, gz.shape[-1])
        grads.append((gz2.T @ x2, gz2.sum(axis=0)))
        g = gz @ w
    return g, list(reversed(grads))


def auto_deeponet_forward(params: Dict[str, Array], inputs: Array, case_params: Array, label: Optional[Array],
                          act_name: str = "relu", query_idxs: Optional[Array] = None):
    """AutoDeepONet.forward, src/models/auto_deeponet.py:76-147."""
    B, _, H, W = inputs.shape
    u = inputs[:, 0]
    flat = np.concatenate([u.reshape(B, -1), case_params], axis=1)          # :109-116
    br_layers, tr_layers = ffn_layers(params, "branch_net"), ffn_layers(params, "trunk_net")
    xb, cb = ffn_forward(br_layers, flat, act_name)
    if query_idxs is None:
        query_idxs = np.array([(i, j) for i in range(H) for j in range(W)], dtype=np.int64)  # :119-124
    xt_in = ((query_idxs.astype(inputs.dtype) - 50) / 100)                  # :127
    xt, ct = ffn_forward(tr_layers, xt_in, act_name)
    preds = xb @ xt.T + params["bias"][0]                                    # :129-131
    preds = preds + u[:, query_idxs[:, 0], query_idxs[:, 1]]                # :134-135
    out = dict(preds=preds, cache=dict(cb=cb, ct=ct, xb=xb, xt=xt, q=query_idxs, shape=(B, H, W)))
    if label is not None:
        labels = label[:, 0][:, query_idxs[:, 0], query_idxs[:, 1]]
        out["loss"] = O.mse_loss(preds, labels, True)
        out["cache"]["labels"] = labels
    else:
        out["preds"] = preds.reshape(B, 1, H, W)
    return out


def auto_deeponet_backward(params: Dict[str, Array], cache: dict, gpreds: Array, act_name: str = "relu") -> Dict[str, Array]:
    """Gradients of all parameters (and of ``inputs``) given d loss / d preds (b, k)."""
    br_layers, tr_layers = ffn_layers(params, "branch_net"), ffn_layers(params, "trunk_net")
    gxb = gpreds @ cache["xt"]
    gxt = gpreds.T @ cache["xb"]
    g_flat, gb_br = ffn_backward(br_layers, cache["cb"], gxb, act_name)
    _, gb_tr = ffn_backward(tr_layers, cache["ct"], gxt, act_name)
    grads = {"bias": np.array([gpreds.sum()])}
    for i, (gw, gb) in enumerate(gb_br):
        grads[f"branch_net.layers.{2 * i}.weight"], grads[f"branch_net.layers.{2 * i}.bias"] = gw, gb
    for i, (gw, gb) in enumerate(gb_tr):
        grads[f"trunk_net.layers.{2 * i}.weight"], grads[f"trunk_net.layers.{2 * i}.bias"] = gw, gb
    B, H, W = cache["shape"]
    gu = g_flat[:, :H * W].reshape(B, H, W).copy()
    q = cache["q"]
    np.add.at(gu, (slice(None), q[:, 0], q[:, 1]), gpreds)
    grads["__u__"] = gu
    return grads


def make_params(seed: int, branch_dim: int, width: int, branch_depth: int, trunk_depth: int, trunk_dim: int = 2,
                dtype=np.float32) -> Dict[str, Array]:
    """nn.Linear default init distributions (U(+-1/sqrt(fan_in))) from a NumPy stream; bias parameter non-zero so its
    path is exercised.  Key order = the reference's state_dict order (bias first)."""
    rng = np.random.default_rng(seed)
    p: Dict[str, Array] = {"bias": np.array([0.05], dtype=dtype)}
    for prefix, dims in (("branch_net", [branch_dim] + [width] * branch_depth), ("trunk_net", [trunk_dim] + [width] * trunk_depth)):
        for i in range(len(dims) - 1):
            bound = 1.0 / np.sqrt(dims[i])
            p[f"{prefix}.layers.{2 * i}.weight"] = rng.uniform(-bound, bound, (dims[i + 1], dims[i])).astype(dtype)
            p[f"{prefix}.layers.{2 * i}.bias"] = rng.uniform(-bound, bound, (dims[i + 1],)).astype(dtype)
    return p


# ---- NormAct (src/models/act_fn.py:21-47) and the non-autoregressive DeepONet / FfnModel forward ----
def normact(x: Array, name: str):
    dims = tuple(range(1, x.ndim))
    mean = x.mean(axis=dims, keepdims=True)
    std = x.std(axis=dims, keepdims=True, ddof=1)
    u = (x - mean) / std
    return act(u, name) * std + mean, dict(u=u, std=std, name=name)


def normact_bwd(g: Array, cache: dict) -> Array:
    u, std, name = cache["u"], cache["std"], cache["name"]
    dims = tuple(range(1, u.ndim))
    L = u[0].size
    gu = g * std * act_grad(u, name)
    Gmu = g.sum(axis=dims, keepdims=True) - gu.sum(axis=dims, keepdims=True) / std
    Gsd = (g * act(u, name)).sum(axis=dims, keepdims=True) - (gu * u).sum(axis=dims, keepdims=True) / std
    return gu / std + Gmu / L + Gsd * u / (L - 1)


def ffn_forward_norm(layers, x: Array, act_name: str, act_norm: bool, act_on_output: bool = False) -> Array:
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        x = x @ w.T + b
        if i < n - 1 or act_on_output:
            x = normact(x, act_name)[0] if act_norm else act(x, act_name)
    return x


def deeponet_forward(params: Dict[str, Array], case_params: Array, t: Array, query_idxs: Array, act_name: str = "relu",
                     act_norm: bool = True) -> Array:
    """DeepONet.forward, src/models/deeponet.py:153-223 (given query_idxs): preds (b,k)."""
    xt = t @ params["fc_trunk_t.weight"].T + params["fc_trunk_t.bias"]                                   # (b,p)
    xy = query_idxs.astype(t.dtype) @ params["fc_trunk_xy.weight"].T + params["fc_trunk_xy.bias"]        # (k,p)
    x_trunk = xt[:, None, :] + xy[None, :, :]
    xb = ffn_forward_norm(ffn_layers(params, "branch_net"), case_params, act_name, act_norm)
    xtr = ffn_forward_norm(ffn_layers(params, "trunk_net"), x_trunk, act_name, act_norm)
    return np.einsum("bp,bkp->bk", xb, xtr) + params["bias"][0]


def ffnmodel_forward(params: Dict[str, Array], case_params: Array, t: Array, query_idxs: Array, act_name: str = "relu",
                     act_norm: bool = True) -> Array:
    """FfnModel.forward, src/models/ffn.py:73-146 (given query_idxs): preds (b,k)."""
    B, K = case_params.shape[0], query_idxs.shape[0]
    coords = np.broadcast_to(query_idxs.astype(t.dtype)[None], (B, K, 2))
    tt = np.broadcast_to(t[:, None, :], (B, K, 1))
    cp = np.broadcast_to(case_params[:, None, :], (B, K, case_params.shape[1]))
    inp = np.concatenate([cp, coords, tt], axis=-1).reshape(B * K, -1)
    return ffn_forward_norm(ffn_layers(params, "ffn"), inp, act_name, act_norm).reshape(B, K)


# ---- AutoEDeepONet / AutoFfn (forward only; their gradients are pinned by the golden files of the reference) --------
def auto_edeeponet_forward(params: Dict[str, Array], inputs: Array, case_params: Array, act_name: str = "relu",
                           query_idxs: Optional[Array] = None) -> Array:
    """AutoEDeepONet.forward without a label, src/models/auto_edeeponet.py:66-131: (b, k) predictions."""
    B, _, H, W = inputs.shape
    u = inputs[:, 0]
    b1, _ = ffn_forward(ffn_layers(params, "branch1"), u.reshape(B, -1), act_name)
    b2, _ = ffn_forward(ffn_layers(params, "branch2"), case_params, act_name)
    if query_idxs is None:
        query_idxs = np.array([(i, j) for i in range(H) for j in range(W)], dtype=np.int64)
    xt, _ = ffn_forward(ffn_layers(params, "trunk_net"), (query_idxs.astype(inputs.dtype) - 50) / 100, act_name)
    preds = (b1 * b2) @ xt.T + params["bias"][0]                         # :93,107-109
    return preds + u[:, query_idxs[:, 0], query_idxs[:, 1]]             # :111-112


def auto_ffn_forward(params: Dict[str, Array], inputs: Array, case_params: Array, act_name: str = "relu",
                     query_idxs: Optional[Array] = None) -> Array:
    """AutoFfn.forward, src/models/auto_ffn.py:55-125, with the reference's own sample pairing: ``repeat`` tiles the b
    frames k times and the k queries b times, so sample r pairs frame r % b with query r % k, and ``view(b, -1)`` then
    reads the b*k rows frame-major."""
    B, _, H, W = inputs.shape
    u = inputs[:, 0]
    flat = np.concatenate([u.reshape(B, -1), case_params], axis=1)
    if query_idxs is None:
        query_idxs = np.array([(i, j) for i in range(H) for j in range(W)], dtype=np.int64)
    K = query_idxs.shape[0]
    rows = np.concatenate([np.tile(flat, (K, 1)), np.tile(query_idxs.astype(inputs.dtype), (B, 1))], axis=1)  # :100-106
    y, _ = ffn_forward(ffn_layers(params, "ffn"), rows, act_name)
    preds = y.reshape(B, -1)                                              # :109
    return preds + u[:, query_idxs[:, 0], query_idxs[:, 1]]             # :112-113
