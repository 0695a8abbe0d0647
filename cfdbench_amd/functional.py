"""torch.autograd.Function wrappers over the C ABI (include/cfdbench_amd.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current stream and autograd's graph.
All arithmetic on the tensors happens in the HIP kernels; nothing in this module has a CPU or ATen fallback.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _lib
from ._capi import FnoParams, FnoShape


def _require_cuda(*ts: Optional[Tensor]):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cfdbench_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _bytes(n: int, device) -> Tensor:
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def _creal(t: Tensor) -> Tensor:
    """complex64 parameter -> its contiguous interleaved storage (no copy when already contiguous)."""
    return torch.view_as_real(t.contiguous())


# ----------------------------------------------------------------------------------------------------
# SpectralConv2d_fast  (src/models/fno/fno2d.py:59-82)
# ----------------------------------------------------------------------------------------------------
class SpectralConv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, w2: Tensor):
        _require_cuda(x, w1, w2)
        api = _lib.api()
        x = _f32c(x)
        B, Cin, H, W = x.shape
        Cin_w, Cout, m1, m2 = w1.shape
        if Cin_w != Cin:
            raise RuntimeError(f"SpectralConv2d: input has {Cin} channels, weights expect {Cin_w}")
        plan = _lib.plan(H, W, m1, m2, x.device.index)
        w1r, w2r = _creal(w1.detach()), _creal(w2.detach())
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        xh = torch.empty((B, Cin, 2 * m1, m2, 2), dtype=torch.float32, device=x.device)
        z = torch.empty((B, Cout, 2 * m1, m2, 2), dtype=torch.float32, device=x.device)
        api.call("cfd_spectral_conv2d_fwd", plan, _ptr(x), _ptr(w1r), _ptr(w2r), _ptr(y), _ptr(xh), _ptr(z), B, Cin, Cout,
                 _stream())
        ctx.save_for_backward(xh, w1r, w2r)
        ctx.dims = (B, Cin, Cout, H, W, m1, m2, plan)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        xh, w1r, w2r = ctx.saved_tensors
        B, Cin, Cout, H, W, m1, m2, plan = ctx.dims
        gy = _f32c(gy)
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=gy.device) if need_x else None
        gw1 = torch.empty((Cin, Cout, m1, m2, 2), dtype=torch.float32, device=gy.device) if need_w else None
        gw2 = torch.empty((Cin, Cout, m1, m2, 2), dtype=torch.float32, device=gy.device) if need_w else None
        ws = _bytes(api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, Cin, Cout), gy.device)
        api.call("cfd_spectral_conv2d_bwd", plan, _ptr(gy), _ptr(xh), _ptr(w1r), _ptr(w2r), _ptr(gx), _ptr(gw1), _ptr(gw2),
                 _ptr(ws), B, Cin, Cout, _stream())
        return (gx, torch.view_as_complex(gw1) if need_w else None, torch.view_as_complex(gw2) if need_w else None)


def spectral_conv2d(x: Tensor, w1: Tensor, w2: Tensor) -> Tensor:
    return SpectralConv2dFn.apply(x, w1, w2)


# ----------------------------------------------------------------------------------------------------
# FnoBlock  (src/models/fno/fno2d.py:85-112) as a stand-alone module
# ----------------------------------------------------------------------------------------------------
class FnoBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, w2: Tensor, w0: Tensor, b0: Tensor, gelu: bool):
        _require_cuda(x, w1, w2, w0, b0)
        api = _lib.api()
        x = _f32c(x)
        B, Cin, H, W = x.shape
        _, Cout, m1, m2 = w1.shape
        plan = _lib.plan(H, W, m1, m2, x.device.index)
        w1r, w2r = _creal(w1.detach()), _creal(w2.detach())
        w0f, b0f = _f32c(w0.detach()), _f32c(b0.detach())
        dev = x.device
        xh = torch.empty((B, Cin, 2 * m1, m2, 2), dtype=torch.float32, device=dev)
        z = torch.empty((B, Cout, 2 * m1, m2, 2), dtype=torch.float32, device=dev)
        pre = torch.empty((B, Cout, H, W), dtype=torch.float32, device=dev)
        st = _stream()
        api.call("cfd_spectral_dft", plan, _ptr(x), _ptr(xh), B * Cin, 0, st)
        api.call("cfd_spectral_mix", plan, _ptr(xh), _ptr(w1r), _ptr(w2r), _ptr(z), B, Cin, Cout, 0, st)
        api.call("cfd_fno_block_fwd", plan, _ptr(x), _ptr(z), _ptr(w0f), _ptr(b0f), _ptr(pre), B, Cin, Cout, 0, st)
        out = pre
        if gelu:
            out = torch.empty_like(pre)
            api.call("cfd_gelu_fwd", _ptr(pre), _ptr(out), pre.numel(), st)
        ctx.save_for_backward(x, xh, pre, w1r, w2r, w0f)
        ctx.meta = (B, Cin, Cout, H, W, m1, m2, plan, gelu)
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        api = _lib.api()
        x, xh, pre, w1r, w2r, w0f = ctx.saved_tensors
        B, Cin, Cout, H, W, m1, m2, plan, gelu = ctx.meta
        dev, st = x.device, _stream()
        g = _f32c(gout)
        if gelu:
            gp = torch.empty_like(g)
            api.call("cfd_gelu_bwd", _ptr(pre), _ptr(g), _ptr(gp), g.numel(), st)
            g = gp
        gh = torch.empty((B, Cout, 2 * m1, m2, 2), dtype=torch.float32, device=dev)
        api.call("cfd_spectral_dft", plan, _ptr(g), _ptr(gh), B * Cout, 0, st)
        gw1 = torch.empty_like(w1r)
        gw2 = torch.empty_like(w2r)
        ws = _bytes(max(api.size("cfd_spectral_wgrad_workspace_bytes", plan, B, Cin, Cout),
                        api.size("cfd_chan_wgrad_workspace_bytes", B, Cin, Cout, H * W)), dev)
        api.call("cfd_spectral_wgrad", plan, _ptr(xh), _ptr(gh), _ptr(gw1), _ptr(gw2), _ptr(ws), B, Cin, Cout, st)
        gw0 = torch.empty_like(w0f)
        gb0 = torch.empty(Cout, dtype=torch.float32, device=dev)
        api.call("cfd_chan_wgrad", _ptr(g), _ptr(x), _ptr(gw0), _ptr(gb0), _ptr(ws), B, Cin, Cout, H * W, 0, st)
        gz = torch.empty((B, Cin, 2 * m1, m2, 2), dtype=torch.float32, device=dev)
        api.call("cfd_spectral_mix", plan, _ptr(gh), _ptr(w1r), _ptr(w2r), _ptr(gz), B, Cin, Cout, 1, st)
        gx = torch.empty_like(x)
        api.call("cfd_fno_block_bwd_input", plan, _ptr(g), _ptr(gz), _ptr(w0f), None, _ptr(gx), B, Cin, Cout, st)
        return gx, torch.view_as_complex(gw1), torch.view_as_complex(gw2), gw0, gb0, None


def fno_block(x: Tensor, w1: Tensor, w2: Tensor, w0: Tensor, b0: Tensor, gelu: bool = True) -> Tensor:
    return FnoBlockFn.apply(x, w1, w2, w0, b0, gelu)


# ----------------------------------------------------------------------------------------------------
# MseLoss sums  (src/models/loss.py:22-37)
# ----------------------------------------------------------------------------------------------------
class LossSumsFn(torch.autograd.Function):
    """sums = [sum (p-l)^2, sum |p-l|, sum l^2, n] as one device tensor."""

    @staticmethod
    def forward(ctx, preds: Tensor, labels: Tensor):
        _require_cuda(preds, labels)
        api = _lib.api()
        if preds.shape != labels.shape:
            raise RuntimeError(f"MseLoss: preds {tuple(preds.shape)} vs labels {tuple(labels.shape)}")
        p, l = _f32c(preds), _f32c(labels)
        n = p.numel()
        sums = torch.empty(4, dtype=torch.float32, device=p.device)
        ws = _bytes(api.size("cfd_loss_workspace_bytes", n), p.device)
        api.call("cfd_masked_loss_sums", _ptr(p), _ptr(l), _ptr(sums), _ptr(ws), n, _stream())
        ctx.save_for_backward(p, l)
        return sums

    @staticmethod
    def backward(ctx, gs: Tensor):
        api = _lib.api()
        p, l = ctx.saved_tensors
        gs = _f32c(gs)
        gp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        gl = torch.empty_like(l) if ctx.needs_input_grad[1] else None
        api.call("cfd_loss_sums_bwd", _ptr(p), _ptr(l), _ptr(gs), _ptr(gp), _ptr(gl), p.numel(), _stream())
        return gp, gl


class LossScoresFn(torch.autograd.Function):
    """(mse, rmse, mae, nmse) from the sums tensor (loss.py:27-35) as one launch per direction.  The first version wrote the
    formulas with torch scalar ops: ~25 one-element kernels per training step (select, div, sqrt and -- in the backward pass -- a
    zero fill plus a copy for every ``sums[i]``), 60 us of a 600-us Auto-DeepONet step.  Same fp32 operations in the same order."""

    @staticmethod
    def forward(ctx, sums: Tensor):
        _require_cuda(sums)
        sums = _f32c(sums)
        scores = torch.empty(4, dtype=torch.float32, device=sums.device)
        _lib.api().call("cfd_loss_scores", _ptr(sums), _ptr(scores), _stream())
        ctx.save_for_backward(sums)
        ctx.set_materialize_grads(False)
        return scores[0], scores[1], scores[2], scores[3]

    @staticmethod
    def backward(ctx, g_mse, g_rmse, g_mae, g_nmse):
        (sums,) = ctx.saved_tensors
        gs = [None if g is None else _f32c(g) for g in (g_mse, g_rmse, g_mae, g_nmse)]
        gsums = torch.empty(4, dtype=torch.float32, device=sums.device)
        _lib.api().call("cfd_loss_scores_bwd", _ptr(sums), *[_ptr(g) for g in gs], _ptr(gsums), _stream())
        return gsums


def rows_concat2(a: Tensor, b: Tensor) -> Tensor:
    """[a | b] along dim 1 for 2-D float32 CUDA tensors whose rows are contiguous (any row stride: a channel slice viewed as (B, H W)) in one
    launch (cfd_rows_concat2); no autograd (the Auto-DeepONet family's branch input is assembled from tensors without gradients)."""
    _require_cuda(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0] or a.stride(1) != 1 or b.stride(1) != 1 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("rows_concat2: (rows, ka) and (rows, kb) float32 tensors with unit inner stride")
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
    _lib.api().call("cfd_rows_concat2", _ptr(a), a.stride(0), a.shape[1], _ptr(b), b.stride(0), b.shape[1], _ptr(out), a.shape[0], _stream())
    return out


def _row_strided(t: Tensor):
    """(rows, cols, row stride) if `t` is a float32 NON-contiguous tensor whose rows t[i] are contiguous blocks at a uniform stride (a channel
    slice x[:, 0] of a contiguous (B, C, H, W) tensor, or its (B, H W) view) -- else None."""
    if t.dtype != torch.float32 or t.dim() < 2 or t.is_contiguous() or t.shape[0] < 1 or t.numel() == 0:
        return None
    cols = t.numel() // t.shape[0]
    if not t[0].is_contiguous() or t.stride(0) < cols:
        return None
    return int(t.shape[0]), int(cols), int(t.stride(0))


class MseLossFn(torch.autograd.Function):
    """(mse, rmse, mae, nmse) of MseLoss.forward (loss.py:22-37) as ONE autograd node: LossSumsFn + LossScoresFn took five launches per
    training step (partial sums, final sum, scores | score gradients, element gradients), this one takes three (cfd_mse_loss_fwd /
    cfd_mse_loss_bwd: the same fp32 operations in the same order).  Every autograd model's loss goes through it."""

    @staticmethod
    def forward(ctx, preds: Tensor, labels: Tensor):
        _require_cuda(preds, labels)
        api = _lib.api()
        if preds.shape != labels.shape:
            raise RuntimeError(f"MseLoss: preds {tuple(preds.shape)} vs labels {tuple(labels.shape)}")
        p = _f32c(preds)
        n = p.numel()
        sums = torch.empty(4, dtype=torch.float32, device=p.device)
        scores = torch.empty(4, dtype=torch.float32, device=p.device)
        ws = _bytes(api.size("cfd_loss_workspace_bytes", n), p.device)
        # labels as strided rows (a channel slice label[:, 0] viewed as (B, H W): cfd_mse_loss_fwd_ld) -- no contiguous copy
        ld = _row_strided(labels)
        if ld is not None and not labels.requires_grad and n < 2 ** 31:
            rows, cols, ldl = ld
            l = labels.detach()
            api.call("cfd_mse_loss_fwd_ld", _ptr(p), _ptr(l), _ptr(sums), _ptr(scores), _ptr(ws), rows, cols, ldl, _stream())
            ctx.ld = ld
        else:
            l = _f32c(labels)
            api.call("cfd_mse_loss_fwd", _ptr(p), _ptr(l), _ptr(sums), _ptr(scores), _ptr(ws), n, _stream())
            ctx.ld = None
        ctx.save_for_backward(p, l, sums)
        ctx.set_materialize_grads(False)
        return scores[0], scores[1], scores[2], scores[3]

    @staticmethod
    def backward(ctx, g_mse, g_rmse, g_mae, g_nmse):
        p, l, sums = ctx.saved_tensors
        gs = [None if g is None else _f32c(g) for g in (g_mse, g_rmse, g_mae, g_nmse)]
        gp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        if ctx.ld is not None:
            rows, cols, ldl = ctx.ld
            _lib.api().call("cfd_mse_loss_bwd_ld", _ptr(p), _ptr(l), _ptr(sums), *[_ptr(g) for g in gs], _ptr(gp), rows, cols, ldl, _stream())
            return gp, None
        gl = torch.empty_like(l) if ctx.needs_input_grad[1] else None
        _lib.api().call("cfd_mse_loss_bwd", _ptr(p), _ptr(l), _ptr(sums), *[_ptr(g) for g in gs], _ptr(gp), _ptr(gl), p.numel(), _stream())
        return gp, gl


def mse_loss_scores(preds: Tensor, labels: Tensor, normalize: bool) -> dict:
    mse, rmse, mae, nmse = MseLossFn.apply(preds, labels)
    out = dict(mse=mse, rmse=rmse, mae=mae)
    if normalize:
        out["nmse"] = nmse
    return out


def scores_from_sums(sums: Tensor, normalize: bool) -> dict:
    """loss.py:27-35 on the 4-float sums tensor (differentiable w.r.t. sums)."""
    mse, rmse, mae, nmse = LossScoresFn.apply(sums)
    out = dict(mse=mse, rmse=rmse, mae=mae)
    if normalize:
        out["nmse"] = nmse
    return out


# ----------------------------------------------------------------------------------------------------
# whole Fno2d forward/backward  (src/models/fno/fno2d.py:178-242)
# ----------------------------------------------------------------------------------------------------
FNO_PARAM_ORDER_DOC = "fc0.weight, fc0.bias, [weights1, weights2, w0.weight, w0.bias] * L, fc1.weight, fc1.bias, fc2.weight, fc2.bias"


def _param_struct(ptrs: Sequence[Optional[int]], L: int) -> FnoParams:
    s = FnoParams()
    s.fc0_w, s.fc0_b = ptrs[0], ptrs[1]
    for l in range(L):
        s.spec_w1[l], s.spec_w2[l], s.w0_w[l], s.w0_b[l] = ptrs[2 + 4 * l: 6 + 4 * l]
    s.fc1_w, s.fc1_b, s.fc2_w, s.fc2_b = ptrs[2 + 4 * L: 6 + 4 * L]
    return s


class FnoForwardFn(torch.autograd.Function):
    """(preds, sums) = Fno2d(inputs, case_params, mask, label); params in FNO_PARAM_ORDER_DOC order."""

    @staticmethod
    def forward(ctx, cfg: dict, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor], label: Optional[Tensor],
                *params: Tensor):
        _require_cuda(inputs, case_params, mask, label, *params)
        api = _lib.api()
        L, C, m1, m2, head = cfg["num_layers"], cfg["hidden"], cfg["modes1"], cfg["modes2"], cfg["head"]
        inputs, case_params, mask, label = _f32c(inputs), _f32c(case_params), _f32c(mask), _f32c(label)
        B, in_chan, H, W = inputs.shape
        out_chan = cfg["out_chan"]
        P = case_params.shape[1]
        if len(params) != 6 + 4 * L:
            raise RuntimeError(f"FnoForwardFn: expected {6 + 4 * L} parameter tensors, got {len(params)}")
        if params[0].shape[1] != in_chan + 3 + P:
            raise RuntimeError(f"Fno2d: fc0 expects {params[0].shape[1]} features but inputs provide {in_chan}+3+{P}")
        plan = _lib.plan(H, W, m1, m2, inputs.device.index)
        shape = FnoShape(B, H, W, in_chan, out_chan, P, C, L, m1, m2, head)
        flat = [(_creal(p.detach()) if p.is_complex() else _f32c(p.detach())) for p in params]
        pstruct = _param_struct([t.data_ptr() for t in flat], L)
        # needs_input_grad reflects requires_grad whatever the grad mode is (and grad mode is always off inside
        # Function.forward), so the caller passes torch.is_grad_enabled(): under no_grad / inference_mode the forward-only
        # workspace is used and nothing is saved.
        training = bool(cfg.get("grad_enabled", True)) and any(ctx.needs_input_grad[5:])
        ws = _bytes(api.size("cfd_fno_workspace_bytes", plan, ctypes.byref(shape), int(training)), inputs.device)
        preds = torch.empty((B, out_chan, H, W), dtype=torch.float32, device=inputs.device)
        sums = torch.empty(4, dtype=torch.float32, device=inputs.device) if label is not None else None
        api.call("cfd_fno_forward", plan, ctypes.byref(shape), ctypes.byref(pstruct), _ptr(inputs), _ptr(case_params),
                 _ptr(mask), _ptr(label), _ptr(preds), _ptr(sums), _ptr(ws), int(training), _stream())
        if training:
            ctx.cfg, ctx.shape, ctx.plan, ctx.ws = cfg, shape, plan, ws
            ctx.has_mask, ctx.has_label = mask is not None, label is not None
            saved = [inputs, case_params, preds] + ([mask] if mask is not None else []) + ([label] if label is not None else [])
            ctx.save_for_backward(*saved, *flat)
            ctx.n_saved_head = len(saved)
        ctx.set_materialize_grads(False)
        return preds, sums

    @staticmethod
    def backward(ctx, gpreds: Optional[Tensor], gsums: Optional[Tensor]):
        api = _lib.api()
        saved = ctx.saved_tensors
        head, flat = saved[:ctx.n_saved_head], saved[ctx.n_saved_head:]
        inputs, case_params, preds = head[0], head[1], head[2]
        k = 3
        mask = head[k] if ctx.has_mask else None
        k += int(ctx.has_mask)
        label = head[k] if ctx.has_label else None
        L = ctx.cfg["num_layers"]
        if ctx.ws is None:
            raise RuntimeError("Fno2d: backward called twice on the same graph -- the activations live in a workspace that "
                               "the first backward pass consumed (retain_graph is not supported; run forward again)")
        if gpreds is None and gsums is None:
            return (None,) * (5 + len(flat))
        use_label = label is not None and gsums is not None
        coef = _f32c(gsums)[:2].contiguous() if use_label else None
        gext = _f32c(gpreds) if gpreds is not None else None
        if not use_label and gext is None:
            return (None,) * (5 + len(flat))
        grads = [torch.empty_like(t) for t in flat]
        pstruct = _param_struct([t.data_ptr() for t in flat], L)
        gstruct = _param_struct([t.data_ptr() for t in grads], L)
        api.call("cfd_fno_backward", ctx.plan, ctypes.byref(ctx.shape), ctypes.byref(pstruct), ctypes.byref(gstruct),
                 _ptr(inputs), _ptr(case_params), _ptr(mask), _ptr(label if use_label else None), _ptr(preds), _ptr(gext),
                 _ptr(coef), _ptr(ctx.ws), _stream())
        ctx.ws = None
        out = [torch.view_as_complex(g) if (g.dim() == 5 and g.shape[-1] == 2) else g for g in grads]
        # reshape 1x1 conv weights back to (out, in, 1, 1): empty_like(flat) already carries the parameter's shape
        return (None, None, None, None, None, *out)


# ----------------------------------------------------------------------------------------------------
# Dense layers of the DeepONet family  (src/models/ffn.py:12-35, src/models/auto_deeponet.py:127-135)
# ----------------------------------------------------------------------------------------------------
ACT_CODES = {None: 0, "none": 0, "relu": 1, "tanh": 2, "gelu": 3, "swish": 4}


class LinearActFn(torch.autograd.Function):
    """y = act(x w^T + b) as one MFMA GEMM with fused epilogue (nn.Linear followed by get_act_fn(name))."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], act: int):
        _require_cuda(x, w, b)
        api = _lib.api()
        lead = x.shape[:-1]
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w, b = _f32c(w.detach()), _f32c(b.detach()) if b is not None else None
        M, K = x2.shape
        N = w.shape[0]
        if w.shape[1] != K:
            raise RuntimeError(f"Linear: input has {K} features, weight expects {w.shape[1]}")
        y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        pre = torch.empty_like(y) if act >= 3 else None
        nws = api.size("cfd_linear_fwd_workspace_bytes", M, K, N)
        ws = _bytes(nws, x2.device) if nws else None
        api.call("cfd_linear_fwd", _ptr(x2), _ptr(w), _ptr(b), _ptr(y), _ptr(pre), _ptr(ws), M, K, N, act, _stream())
        ctx.save_for_backward(x2, w, y if act in (1, 2) else None, pre)
        ctx.meta = (M, K, N, act, lead, b is not None)
        return y.reshape(*lead, N)

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        x2, w, y, pre = ctx.saved_tensors
        M, K, N, act, lead, has_b = ctx.meta
        gy2 = _f32c(gy).reshape(M, N)
        dev = gy2.device
        gx = torch.empty((M, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        gw = torch.empty((N, K), dtype=torch.float32, device=dev)
        gb = torch.empty((N,), dtype=torch.float32, device=dev) if has_b else None
        ws = _bytes(api.size("cfd_linear_bwd_workspace_bytes", M, K, N), dev)
        api.call("cfd_linear_bwd", _ptr(gy2), _ptr(x2), _ptr(w), _ptr(y), _ptr(pre), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(ws),
                 M, K, N, act, _stream())
        return (gx.reshape(*lead, K) if gx is not None else None), gw, gb, None


class LinearChainFn(torch.autograd.Function):
    """A run of Linear(+activation) layers too wide for the stack kernel (FfnStackFn: widths <= 128) as ONE autograd node: the same
    GEMM launches per layer as LinearActFn, but in the backward pass a layer's input gradient leaves its GEMM already multiplied by the
    previous layer's activation derivative (cfd_linear_bwd_ex), so the separate activation-gradient pass over every (rows, width)
    gradient disappears (8 x 51 us per Auto-FFN train step).  apply(x, acts, *[w_0, b_0, w_1, b_1, ...]), acts = activation code per layer."""

    @staticmethod
    def forward(ctx, x: Tensor, acts, *wb: Tensor):
        _require_cuda(x, *wb)
        api = _lib.api()
        L = len(wb) // 2
        ws_ = [_f32c(t.detach()) for t in wb[0::2]]
        bs_ = [None if t is None else _f32c(t.detach()) for t in wb[1::2]]
        lead = x.shape[:-1]
        h = _f32c(x).reshape(-1, x.shape[-1])
        x2, M = h, h.shape[0]
        ys, pres = [], []
        for l in range(L):
            N, K = ws_[l].shape
            if h.shape[1] != K:
                raise RuntimeError(f"Linear: input has {h.shape[1]} features, weight expects {K}")
            y = torch.empty((M, N), dtype=torch.float32, device=h.device)
            pre = torch.empty_like(y) if acts[l] >= 3 else None
            nws = api.size("cfd_linear_fwd_workspace_bytes", M, K, N)
            ws = _bytes(nws, h.device) if nws else None
            api.call("cfd_linear_fwd", _ptr(h), _ptr(ws_[l]), _ptr(bs_[l]), _ptr(y), _ptr(pre), _ptr(ws), M, K, N, acts[l], _stream())
            ys.append(y)
            pres.append(pre)
            h = y
        ctx.save_for_backward(x2, *ws_, *ys, *[p for p in pres if p is not None])
        ctx.meta = (L, tuple(acts), lead, [b is not None for b in bs_], [p is not None for p in pres])
        return h.reshape(*lead, h.shape[1])

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        L, acts, lead, has_b, has_p = ctx.meta
        saved = ctx.saved_tensors
        x2, ws_, ys = saved[0], list(saved[1:1 + L]), list(saved[1 + L:1 + 2 * L])
        pi = iter(saved[1 + 2 * L:])
        pres = [next(pi) if hp else None for hp in has_p]
        M, dev = x2.shape[0], x2.device
        g = _f32c(gy).reshape(M, ws_[-1].shape[0])
        grads = [None] * (2 * L)
        for l in reversed(range(L)):
            N, K = ws_[l].shape
            xin = ys[l - 1] if l > 0 else x2
            need_gx = l > 0 or ctx.needs_input_grad[0]
            gx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_gx else None
            gw = torch.empty((N, K), dtype=torch.float32, device=dev)
            gb = torch.empty((N,), dtype=torch.float32, device=dev) if has_b[l] else None
            ws = _bytes(api.size("cfd_linear_bwd_workspace_bytes", M, K, N), dev)
            act = acts[l] if l == L - 1 else 0  # (below the last layer g already is dZ_l: the layer above folded act_l' into its GEMM)
            in_act = acts[l - 1] if l > 0 else 0
            api.call("cfd_linear_bwd_ex", _ptr(g), _ptr(xin), _ptr(ws_[l]), _ptr(ys[l]), _ptr(pres[l]), _ptr(gx), _ptr(gw), _ptr(gb),
                     _ptr(ws), M, K, N, act, in_act, _ptr(pres[l - 1]) if l > 0 else None, _stream())
            grads[2 * l], grads[2 * l + 1] = gw, gb
            g = gx
        return (g.reshape(*lead, x2.shape[1]) if g is not None else None), None, *grads


def linear_chain(x: Tensor, weights, biases, acts) -> Tensor:
    wb = []
    for w, b in zip(weights, biases):
        wb += [w, b]
    return LinearChainFn.apply(x, tuple(ACT_CODES[a] for a in acts), *wb)


class ActFn(torch.autograd.Function):
    """y = act(x) elementwise (get_act_fn(name), act_fn.py:8-18) for tensors that are not a GEMM output."""

    @staticmethod
    def forward(ctx, x: Tensor, act: int):
        _require_cuda(x)
        api = _lib.api()
        x = _f32c(x)
        y = torch.empty_like(x)
        api.call("cfd_act_fwd", _ptr(x), _ptr(y), x.numel(), act, _stream())
        ctx.save_for_backward(y if act in (1, 2) else x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        (s,) = ctx.saved_tensors
        gy = _f32c(gy)
        gx = torch.empty_like(gy)
        yy, xx = (s, None) if ctx.act in (1, 2) else (None, s)
        api.call("cfd_act_bwd", _ptr(gy), _ptr(yy), _ptr(xx), _ptr(gx), gy.numel(), ctx.act, _stream())
        return gx, None


def act(x: Tensor, name: Optional[str]) -> Tensor:
    return x if name is None else ActFn.apply(x, ACT_CODES[name])


def linear_act(x: Tensor, w: Tensor, b: Optional[Tensor], act: Optional[str]) -> Tensor:
    return LinearActFn.apply(x, w, b, ACT_CODES[act])


FFN_STACK_MAX_WIDTH, FFN_STACK_MAX_LAYERS = 128, 16


def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class FfnStackFn(torch.autograd.Function):
    """A run of Linear(+activation) layers whose widths are all <= 128 as ONE kernel per direction (cfd_ffn_stack_fwd / _bwd,
    csrc/ffn.hip): the rows' activations stay in LDS from layer to layer; only what backward needs is stored."""

    @staticmethod
    def forward(ctx, x: Tensor, act: int, act_last: bool, *wb: Tensor):
        import ctypes
        _require_cuda(x, *wb)
        api = _lib.api()
        L = len(wb) // 2
        ws_ = [_f32c(t.detach()) for t in wb[0::2]]
        bs_ = [None if t is None else _f32c(t.detach()) for t in wb[1::2]]
        lead = x.shape[:-1]
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        R = x2.shape[0]
        dims = [x2.shape[1]] + [w.shape[0] for w in ws_]
        for l, w in enumerate(ws_):
            if w.shape[1] != dims[l]:
                raise RuntimeError(f"Ffn: layer {l} expects {w.shape[1]} features, got {dims[l]}")
        ys = [torch.empty((R, d), dtype=torch.float32, device=x2.device) for d in dims[1:]]
        zs = [torch.empty_like(y) if (act >= 3 and (l + 1 < L or act_last)) else None for l, y in enumerate(ys)]
        cdims = (ctypes.c_int * (L + 1))(*dims)
        api.call("cfd_ffn_stack_fwd", _ptr(x2), _ptr_array(ws_), _ptr_array(bs_), _ptr_array(ys), _ptr_array(zs), R, cdims, L, act,
                 int(act_last), _stream())
        ctx.save_for_backward(x2, *ws_, *ys, *[z for z in zs if z is not None])
        ctx.meta = (L, dims, act, bool(act_last), lead, [b is not None for b in bs_], [z is not None for z in zs])
        return ys[-1].reshape(*lead, dims[-1])

    @staticmethod
    def backward(ctx, gy: Tensor):
        import ctypes
        api = _lib.api()
        L, dims, act, act_last, lead, has_b, has_z = ctx.meta
        saved = ctx.saved_tensors
        x2, ws_, ys = saved[0], list(saved[1:1 + L]), list(saved[1 + L:1 + 2 * L])
        zi = iter(saved[1 + 2 * L:])
        zs = [next(zi) if hz else None for hz in has_z]
        R = x2.shape[0]
        dev = x2.device
        gy2 = _f32c(gy).reshape(R, dims[-1])
        gws = [torch.empty_like(w) for w in ws_]
        gbs = [torch.empty((w.shape[0],), dtype=torch.float32, device=dev) if hb else None for w, hb in zip(ws_, has_b)]
        gx = torch.empty((R, dims[0]), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        cdims = (ctypes.c_int * (L + 1))(*dims)
        ws = _bytes(api.size("cfd_ffn_stack_bwd_workspace_bytes", R, cdims, L), dev)
        api.call("cfd_ffn_stack_bwd", _ptr(x2), _ptr(gy2), _ptr_array(ws_), _ptr_array(ys), _ptr_array(zs), _ptr_array(gws),
                 _ptr_array(gbs), _ptr(gx), _ptr(ws), R, cdims, L, act, int(act_last), _stream())
        grads = []
        for gw, gb in zip(gws, gbs):
            grads += [gw, gb]
        return (gx.reshape(*lead, dims[0]) if gx is not None else None), None, None, *grads


class FfnStacksFn(torch.autograd.Function):
    """Up to three independent Linear(+activation) stacks -- the branch and trunk nets of a DeepONet variant -- as ONE kernel launch
    per direction (cfd_ffn_stacks_fwd / _bwd): a stack of 512 rows is 32 workgroups, one of 4290 rows 269, and each launch sits at
    its latency floor.  apply(meta, *tensors): meta = ((act code, act_last, number of layers), ...) per stack, tensors = per stack
    [x, w_0, b_0, w_1, b_1, ...]; returns one output per stack."""

    @staticmethod
    def _args(api, n):
        from ._capi import FfnStackArgs
        return (FfnStackArgs * n)()

    @staticmethod
    def forward(ctx, meta, *tensors):
        import ctypes
        api = _lib.api()
        n = len(meta)
        args = FfnStacksFn._args(api, n)
        keep, per, outs, pos = [], [], [], 0
        for i, (act, act_last, L) in enumerate(meta):
            x = tensors[pos]
            wb = tensors[pos + 1:pos + 1 + 2 * L]
            pos += 1 + 2 * L
            _require_cuda(x, *wb)
            ws_ = [_f32c(t.detach()) for t in wb[0::2]]
            bs_ = [None if t is None else _f32c(t.detach()) for t in wb[1::2]]
            lead = x.shape[:-1]
            x2 = _f32c(x).reshape(-1, x.shape[-1])
            R = x2.shape[0]
            dims = [x2.shape[1]] + [w.shape[0] for w in ws_]
            for l, w in enumerate(ws_):
                if w.shape[1] != dims[l]:
                    raise RuntimeError(f"Ffn: layer {l} expects {w.shape[1]} features, got {dims[l]}")
            ys = [torch.empty((R, d), dtype=torch.float32, device=x2.device) for d in dims[1:]]
            zs = [torch.empty_like(y) if (act >= 3 and (l + 1 < L or act_last)) else None for l, y in enumerate(ys)]
            arrs = (_ptr_array(ws_), _ptr_array(bs_), _ptr_array(ys), _ptr_array(zs), (ctypes.c_int * (L + 1))(*dims))
            keep.append(arrs)
            a = args[i]
            a.x, a.R, a.L, a.act, a.act_last = _ptr(x2), R, L, act, int(act_last)
            a.w, a.b, a.y, a.z, a.dims = (ctypes.addressof(t) for t in arrs)
            per.append((x2, ws_, ys, zs, dims, lead, [b is not None for b in bs_]))
            outs.append(ys[-1].reshape(*lead, dims[-1]))
        api.call("cfd_ffn_stacks_fwd", n, args, _stream())
        saved, layout = [], []
        for x2, ws_, ys, zs, dims, lead, has_b in per:
            layout.append((len(saved), dims, lead, has_b, [z is not None for z in zs]))
            saved += [x2, *ws_, *ys, *[z for z in zs if z is not None]]
        ctx.save_for_backward(*saved)
        ctx.meta, ctx.layout = meta, layout
        # which x need a gradient: tensors[...] positions of the inputs
        ctx.x_pos = []
        pos = 0
        for (_, _, L) in meta:
            ctx.x_pos.append(pos + 1)  # (+1: `meta` is argument 0 of apply)
            pos += 1 + 2 * L
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        import ctypes
        api = _lib.api()
        n = len(ctx.meta)
        args = FfnStacksFn._args(api, n)
        saved = ctx.saved_tensors
        keep, grads = [], [None]
        for i, ((act, act_last, L), (s0, dims, lead, has_b, has_z)) in enumerate(zip(ctx.meta, ctx.layout)):
            x2 = saved[s0]
            ws_ = list(saved[s0 + 1:s0 + 1 + L])
            ys = list(saved[s0 + 1 + L:s0 + 1 + 2 * L])
            zi = iter(saved[s0 + 1 + 2 * L:s0 + 1 + 2 * L + sum(has_z)])
            zs = [next(zi) if hz else None for hz in has_z]
            R, dev = x2.shape[0], x2.device
            gy = gys[i]
            gy2 = _f32c(gy).reshape(R, dims[-1]) if gy is not None else torch.zeros((R, dims[-1]), dtype=torch.float32, device=dev)
            gws = [torch.empty_like(w) for w in ws_]
            gbs = [torch.empty((w.shape[0],), dtype=torch.float32, device=dev) if hb else None for w, hb in zip(ws_, has_b)]
            gx = torch.empty((R, dims[0]), dtype=torch.float32, device=dev) if ctx.needs_input_grad[ctx.x_pos[i]] else None
            cdims = (ctypes.c_int * (L + 1))(*dims)
            ws = _bytes(api.size("cfd_ffn_stack_bwd_workspace_bytes", R, cdims, L), dev)
            arrs = (_ptr_array(ws_), _ptr_array(ys), _ptr_array(zs), _ptr_array(gws), _ptr_array(gbs), cdims)
            keep.append((arrs, gy2, ws))
            a = args[i]
            a.x, a.gy, a.gx, a.ws, a.R, a.L, a.act, a.act_last = _ptr(x2), _ptr(gy2), _ptr(gx), _ptr(ws), R, L, act, int(act_last)
            a.w, a.y, a.z, a.gw, a.gb, a.dims = (ctypes.addressof(t) for t in arrs)
            grads.append(gx.reshape(*lead, dims[0]) if gx is not None else None)
            for gw, gb in zip(gws, gbs):
                grads += [gw, gb]
        api.call("cfd_ffn_stacks_bwd", n, args, _stream())
        return tuple(grads)


FFN_STACKS_MAX = 3


def ffn_stacks(runs):
    """runs: [(x, weights, biases, act name, act_last), ...] (at most FFN_STACKS_MAX) -> the stacks' outputs, one launch per direction."""
    meta, tensors = [], []
    for x, weights, biases, act, act_last in runs:
        meta.append((ACT_CODES[act], bool(act_last), len(weights)))
        tensors.append(x)
        for w, b in zip(weights, biases):
            tensors += [w, b]
    return list(FfnStacksFn.apply(tuple(meta), *tensors))


def ffn_stack(x: Tensor, weights, biases, act: Optional[str], act_last: bool) -> Tensor:
    wb = []
    for w, b in zip(weights, biases):
        wb += [w, b]
    return FfnStackFn.apply(x, ACT_CODES[act], bool(act_last), *wb)


class DeepONetInnerFn(torch.autograd.Function):
    """preds[b,k] = <branch[b], trunk[k]> + bias + u[b, query k]  (auto_deeponet.py:129-135) as one GEMM."""

    @staticmethod
    def forward(ctx, branch: Tensor, trunk: Tensor, bias: Tensor, u: Optional[Tensor], qidx: Optional[Tensor]):
        _require_cuda(branch, trunk, bias, u, qidx)
        api = _lib.api()
        branch, trunk, bias = _f32c(branch), _f32c(trunk), _f32c(bias.detach())
        B, P = branch.shape
        Kq = trunk.shape[0]
        if u is not None and u.dim() == 2 and u.dtype == torch.float32 and u.stride(1) == 1 and u.stride(0) >= u.shape[1]:
            u2 = u  # rows with a stride (the leading columns of the branch input matrix): read in place
        else:
            u2 = _f32c(u).reshape(B, -1) if u is not None else None
        HW = u2.shape[1] if u2 is not None else 0
        qi = qidx.to(torch.int32).contiguous() if qidx is not None else None
        preds = torch.empty((B, Kq), dtype=torch.float32, device=branch.device)
        api.call("cfd_deeponet_inner_fwd_ex", _ptr(branch), _ptr(trunk), _ptr(bias), _ptr(u2), u2.stride(0) if u2 is not None else 0,
                 _ptr(qi), _ptr(preds), B, P, Kq, HW, _stream())
        ctx.save_for_backward(branch, trunk)
        ctx.has_u = u is not None
        ctx.u_shape = None if u is None else tuple(u.shape)
        ctx.qidx = qi
        return preds

    @staticmethod
    def backward(ctx, g: Tensor):
        api = _lib.api()
        branch, trunk = ctx.saved_tensors
        B, P = branch.shape
        Kq = trunk.shape[0]
        g = _f32c(g)
        dev = g.device
        gbr = torch.empty_like(branch) if ctx.needs_input_grad[0] else None
        gtr = torch.empty_like(trunk) if ctx.needs_input_grad[1] else None
        gbias = torch.empty(1, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        ws = _bytes(api.size("cfd_deeponet_inner_bwd_workspace_bytes", B, P, Kq), dev)
        api.call("cfd_deeponet_inner_bwd", _ptr(g), _ptr(branch), _ptr(trunk), _ptr(gbr), _ptr(gtr), _ptr(gbias), _ptr(ws),
                 B, P, Kq, _stream())
        gu = None
        if ctx.has_u and ctx.needs_input_grad[3]:  # the residual's gradient is g scattered to the query points
            if ctx.qidx is None and Kq == int(torch.tensor(ctx.u_shape[1:]).prod()):
                gu = g.reshape(ctx.u_shape)
            else:
                gu2 = torch.zeros((B, int(torch.tensor(ctx.u_shape[1:]).prod())), dtype=torch.float32, device=dev)
                idx = ctx.qidx.long() if ctx.qidx is not None else torch.arange(Kq, device=dev)
                gu2.index_add_(1, idx, g)
                gu = gu2.reshape(ctx.u_shape)
        return gbr, gtr, gbias, gu, None


# ----------------------------------------------------------------------------------------------------
# Convolution stack of the U-Net / ResNet baselines  (src/models/unet.py, src/models/resnet.py)
# ----------------------------------------------------------------------------------------------------
class PreparedConvWeights:
    """The MFMA fragments of a model's k = 3 / 7 convolution weights, made by ONE launch per forward pass
    (cfd_conv2d_wprep_batch) instead of one small launch in front of every convolution and every input gradient
    (35 per U-Net step, 26 per ResNet step).  ``refresh`` is called at the top of the model's forward; each layer then hands
    ``frags(conv)`` to Conv2dReplicateFn / ConvBnReluFn.  The fragments are a pure function of the weights, so they are
    trusted only while the weight's autograd version is the one they were made from (an optimizer step or a
    ``load_state_dict`` in between sends the layer back to preparing its own) -- and a forward always starts with a refresh.
    CFDBENCH_CONV_PREP=0 turns the whole thing off (timing comparisons)."""

    def __init__(self, convs):
        import os
        self.convs = [c for c in convs if c.kernel_size in ((3, 3), (7, 7))]
        self.enabled = os.environ.get("CFDBENCH_CONV_PREP", "1") != "0"
        self._key = {}
        self._tables = {}
        for c in self.convs:
            c._cfd_wfrag = None

    def __getstate__(self):  # copy.deepcopy / pickle of the owning model: the pointer tables are per-process caches, rebuilt on demand
        d = self.__dict__.copy()
        d["_key"], d["_tables"] = {}, {}
        return d

    def _build(self, api, transposed: bool):
        import ctypes
        items = []
        for c in self.convs:
            w = c.weight
            Co, Ci, ks, _ = w.shape
            for tr in ((0, 1) if transposed else (0,)):
                n = api.size("cfd_conv2d_wfrag_bytes", Ci, Co, ks, tr)
                if n == 0:
                    continue
                buf = getattr(c, "_cfd_wfrag_buf", {})
                if tr not in buf or buf[tr].device != w.device:
                    buf[tr] = torch.empty(n, dtype=torch.uint8, device=w.device)
                c._cfd_wfrag_buf = buf
                items.append((w.data_ptr(), buf[tr].data_ptr(), Ci, Co, ks, tr))
        n = len(items)
        col = lambda j, ty: (ty * n)(*[it[j] for it in items])
        self._tables[transposed] = (n, col(0, ctypes.c_void_p), col(1, ctypes.c_void_p), col(2, ctypes.c_int), col(3, ctypes.c_int),
                                    col(4, ctypes.c_int), col(5, ctypes.c_int))

    def refresh(self, transposed: bool) -> None:
        """Remake every layer's fragments from the current weights (``transposed``: also the input-gradient form)."""
        ws = [c.weight for c in self.convs]
        if not self.enabled or not all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in ws):
            for c in self.convs:
                c._cfd_wfrag = None
            return
        api = _lib.api()
        transposed = bool(transposed)
        key = tuple(w.data_ptr() for w in ws)
        if self._key.get(transposed) != key:
            self._build(api, transposed)
            self._key[transposed] = key
        n, wp, fp, ci, co, ks, tr = self._tables[transposed]
        if n == 0:
            return
        api.call("cfd_conv2d_wprep_batch", n, wp, fp, ci, co, ks, tr, _stream())
        for c in self.convs:
            buf = c._cfd_wfrag_buf
            c._cfd_wfrag = (buf.get(0), buf.get(1) if transposed else None, c.weight._version)


def conv_frags(conv):
    """(forward fragments, input-gradient fragments) of a layer registered with PreparedConvWeights, or (None, None) when
    there are none or the weights changed since they were made."""
    st = getattr(conv, "_cfd_wfrag", None)
    if st is None or st[2] != conv.weight._version:
        return None, None
    return st[0], st[1]


class Conv2dReplicateFn(torch.autograd.Function):
    """nn.Conv2d(k, padding=k//2, padding_mode='replicate') as an implicit GEMM on the matrix pipe."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], wfrag: Optional[Tensor] = None, wfrag_t: Optional[Tensor] = None):
        _require_cuda(x, w, b)
        api = _lib.api()
        x, w = _f32c(x), _f32c(w.detach())
        b = _f32c(b.detach()) if b is not None else None
        B, Ci, H, W = x.shape
        Co, Ci_w, ks, ks2 = w.shape
        if Ci_w != Ci or ks != ks2:
            raise RuntimeError(f"conv2d: input has {Ci} channels, weight is {tuple(w.shape)}")
        out = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        nws = api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks)  # split-K partials of narrow, deep layers
        ws = _bytes(nws, x.device) if nws else None
        api.call("cfd_conv2d_fwd_ex", _ptr(x), _ptr(w), _ptr(b), _ptr(out), _ptr(ws), None, _ptr(wfrag), B, Ci, Co, H, W, ks, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.wfrag_t = wfrag_t  # (PreparedConvWeights: made from these very weights at the top of this forward pass)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        api = _lib.api()
        x, w = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co, _, ks, _ = w.shape
        g = _f32c(g)
        gin = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(Co, dtype=torch.float32, device=g.device) if ctx.has_b else None
        ws = _bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks), g.device)
        api.call("cfd_conv2d_bwd_ex", _ptr(g), _ptr(x), _ptr(w), _ptr(gin), _ptr(gw), _ptr(gb), _ptr(ws), _ptr(ctx.wfrag_t), B, Ci, Co,
                 H, W, ks, _stream())
        return gin, gw, gb, None, None


class BatchNormFn(torch.autograd.Function):
    """[ReLU](BatchNorm2d(x)); running statistics are updated in place in training mode."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, run_mean: Optional[Tensor], run_var: Optional[Tensor],
                training: bool, relu: bool, eps: float, momentum: float):
        _require_cuda(x, gamma, beta, run_mean, run_var)
        api = _lib.api()
        x, gamma, beta = _f32c(x), _f32c(gamma.detach()), _f32c(beta.detach())
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        sm = torch.empty(C, dtype=torch.float32, device=x.device)
        sr = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bytes(api.size("cfd_batchnorm_workspace_bytes", C), x.device)
        api.call("cfd_batchnorm_fwd", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(run_mean), _ptr(run_var), _ptr(y), _ptr(sm),
                 _ptr(sr), _ptr(ws), B, C, H * W, float(eps), float(momentum), int(training), int(relu), _stream())
        ctx.save_for_backward(x, gamma, beta, sm, sr)
        ctx.meta = (B, C, H * W, bool(training), bool(relu))
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        x, gamma, beta, sm, sr = ctx.saved_tensors
        B, C, HW, training, relu = ctx.meta
        gy = _f32c(gy)
        gx = torch.empty_like(x)
        gg = torch.empty(C, dtype=torch.float32, device=x.device)
        gb = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = _bytes(api.size("cfd_batchnorm_workspace_bytes", C), x.device)
        api.call("cfd_batchnorm_bwd", _ptr(gy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(sm), _ptr(sr), _ptr(gx), _ptr(gg),
                 _ptr(gb), _ptr(ws), B, C, HW, int(training), int(relu), _stream())
        return gx, gg, gb, None, None, None, None, None, None


class Conv2dZeroPadFn(torch.autograd.Function):
    """nn.Conv2d(k, padding=k // 2) with its default ZERO padding (the CNN branch of auto_deeponet_cnn.py:17-33) on the three-piece bf16
    kernels (cfd_conv2d_zeropad_fwd / _bwd).  ``supported`` tells whether the kernels take the shape; otherwise the caller zero-pads,
    runs the replicate-padding kernel on the larger grid and crops."""

    @staticmethod
    def supported(x: Tensor, w: Tensor) -> bool:
        if not (x.is_cuda and x.dim() == 4 and w.dim() == 4 and w.shape[2] == w.shape[3] and w.shape[1] == x.shape[1]):
            return False
        B, Ci, H, W = x.shape
        return _lib.api().size("cfd_conv2d_zeropad_supported", B, Ci, w.shape[0], H, W, w.shape[2]) == 1

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor]):
        _require_cuda(x, w, b)
        api = _lib.api()
        x, w = _f32c(x), _f32c(w.detach())
        b = _f32c(b.detach()) if b is not None else None
        B, Ci, H, W = x.shape
        Co, _, ks, _ = w.shape
        y = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        ws = _bytes(api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks), x.device)
        api.call("cfd_conv2d_zeropad_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), _ptr(ws), B, Ci, Co, H, W, ks, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        api = _lib.api()
        x, w = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co, _, ks, _ = w.shape
        gy = _f32c(gy)
        gin = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(Co, dtype=torch.float32, device=x.device) if ctx.has_b else None
        ws = _bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks), x.device)
        api.call("cfd_conv2d_zeropad_bwd", _ptr(gy), _ptr(x), _ptr(w), _ptr(gin), _ptr(gw), _ptr(gb), _ptr(ws), B, Ci, Co, H, W, ks, _stream())
        return gin, gw, gb


class ConvBnReluFn(torch.autograd.Function):
    """[ReLU](BatchNorm2d(Conv2d(x))) in TRAINING mode (unet.py:20-30) as one autograd node: where the conv kernel can emit the
    per-channel partial sums of its output (cfd_conv2d_fwd_stats), the BatchNorm takes its batch statistics from them and runs as
    ONE launch instead of a statistics pass plus a normalising pass; otherwise the two stand-alone calls.  The backward pass is
    BatchNormFn's followed by Conv2dReplicateFn's."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], gamma: Tensor, beta: Tensor, run_mean: Optional[Tensor],
                run_var: Optional[Tensor], relu: bool, eps: float, momentum: float, wfrag: Optional[Tensor] = None,
                wfrag_t: Optional[Tensor] = None):
        _require_cuda(x, w, b, gamma, beta, run_mean, run_var)
        api = _lib.api()
        x, w = _f32c(x), _f32c(w.detach())
        b = _f32c(b.detach()) if b is not None else None
        gamma, beta = _f32c(gamma.detach()), _f32c(beta.detach())
        B, Ci, H, W = x.shape
        Co, Ci_w, ks, ks2 = w.shape
        if Ci_w != Ci or ks != ks2:
            raise RuntimeError(f"conv2d: input has {Ci} channels, weight is {tuple(w.shape)}")
        y0 = torch.empty((B, Co, H, W), dtype=torch.float32, device=x.device)
        z = torch.empty_like(y0)
        sm = torch.empty(Co, dtype=torch.float32, device=x.device)
        sr = torch.empty(Co, dtype=torch.float32, device=x.device)
        nws = api.size("cfd_conv2d_fwd_workspace_bytes", B, Ci, Co, H, W, ks)
        ws = _bytes(nws, x.device) if nws else None
        slots = api.size("cfd_conv2d_fwd_stats_slots", B, Ci, Co, H, W, ks) if ws is not None else 0
        if slots > 0:
            stats = torch.empty((Co, slots, 4), dtype=torch.float32, device=x.device)
            api.call("cfd_conv2d_fwd_ex", _ptr(x), _ptr(w), _ptr(b), _ptr(y0), _ptr(ws), _ptr(stats), _ptr(wfrag), B, Ci, Co, H, W, ks,
                     _stream())
            api.call("cfd_batchnorm_fwd_stats", _ptr(y0), _ptr(gamma), _ptr(beta), _ptr(run_mean), _ptr(run_var), _ptr(z), _ptr(sm),
                     _ptr(sr), _ptr(stats), slots, _ptr(b), B, Co, H * W, float(eps), float(momentum), int(relu), _stream())
        else:
            api.call("cfd_conv2d_fwd_ex", _ptr(x), _ptr(w), _ptr(b), _ptr(y0), _ptr(ws), None, _ptr(wfrag), B, Ci, Co, H, W, ks, _stream())
            bws = _bytes(api.size("cfd_batchnorm_workspace_bytes", Co), x.device)
            api.call("cfd_batchnorm_fwd", _ptr(y0), _ptr(gamma), _ptr(beta), _ptr(run_mean), _ptr(run_var), _ptr(z), _ptr(sm),
                     _ptr(sr), _ptr(bws), B, Co, H * W, float(eps), float(momentum), 1, int(relu), _stream())
        ctx.save_for_backward(x, w, y0, gamma, beta, sm, sr)
        ctx.meta = (b is not None, bool(relu))
        ctx.wfrag_t = wfrag_t
        return z

    @staticmethod
    def backward(ctx, gz: Tensor):
        api = _lib.api()
        x, w, y0, gamma, beta, sm, sr = ctx.saved_tensors
        has_b, relu = ctx.meta
        B, Ci, H, W = x.shape
        Co, _, ks, _ = w.shape
        gz = _f32c(gz)
        gy0 = torch.empty_like(y0)
        gg = torch.empty(Co, dtype=torch.float32, device=x.device)
        gbeta = torch.empty(Co, dtype=torch.float32, device=x.device)
        bws = _bytes(api.size("cfd_batchnorm_workspace_bytes", Co), x.device)
        api.call("cfd_batchnorm_bwd", _ptr(gz), _ptr(y0), _ptr(gamma), _ptr(beta), _ptr(sm), _ptr(sr), _ptr(gy0), _ptr(gg),
                 _ptr(gbeta), _ptr(bws), B, Co, H * W, 1, int(relu), _stream())
        gin = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(Co, dtype=torch.float32, device=x.device) if has_b else None
        ws = _bytes(api.size("cfd_conv2d_bwd_workspace_bytes", B, Ci, Co, H, W, ks), x.device)
        api.call("cfd_conv2d_bwd_ex", _ptr(gy0), _ptr(x), _ptr(w), _ptr(gin), _ptr(gw), _ptr(gb), _ptr(ws), _ptr(ctx.wfrag_t), B, Ci,
                 Co, H, W, ks, _stream())
        return gin, gw, gb, gg, gbeta, None, None, None, None, None, None, None


class MaxPool2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        _require_cuda(x)
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _lib.api().call("cfd_maxpool2_fwd", _ptr(x), _ptr(y), B * C, H, W, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        _lib.api().call("cfd_maxpool2_bwd", _ptr(x), _ptr(_f32c(gy)), _ptr(gx), B * C, H, W, _stream())
        return gx


class MaxPoolSkipFn(torch.autograd.Function):
    """``(MaxPool2d(2)(x), x)`` for an encoder output that also feeds a skip connection (unet.py:179-196: x1 .. x4 go to the next
    Down block AND to an Up block's concatenation).  The second output is x itself; taking both gradients here lets the backward
    pass add the skip connection's share inside the max-pool gradient kernel -- read in place from the gradient of the decoder's
    concatenation (a channel slice with a batch stride) -- instead of a contiguous copy of that slice + autograd's add kernel."""

    @staticmethod
    def forward(ctx, x: Tensor):
        _require_cuda(x)
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _lib.api().call("cfd_maxpool2_fwd", _ptr(x), _ptr(y), B * C, H, W, _stream())
        ctx.save_for_backward(x)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy: Optional[Tensor], gskip: Optional[Tensor]):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        if gy is None:
            return gskip
        gx = torch.empty_like(x)
        api = _lib.api()
        if gskip is None:
            api.call("cfd_maxpool2_bwd", _ptr(x), _ptr(_f32c(gy)), _ptr(gx), B * C, H, W, _stream())
            return gx
        if not (gskip.dtype == torch.float32 and gskip.stride()[1:] == (H * W, W, 1) and gskip.stride(0) >= C * H * W):
            gskip = _f32c(gskip)
        api.call("cfd_maxpool2_bwd_add", _ptr(x), _ptr(_f32c(gy)), _ptr(gskip), gskip.stride(0), _ptr(gx), B, C, H, W, _stream())
        return gx


class UpsampleBilinear2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)  (src/models/unet.py:74-76)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        _require_cuda(x)
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        _lib.api().call("cfd_upsample2_bilinear_fwd", _ptr(x), _ptr(y), B * C, H, W, _stream())
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        B, C, H, W = ctx.shape
        gx = torch.empty((B, C, H, W), dtype=torch.float32, device=gy.device)
        _lib.api().call("cfd_upsample2_bilinear_bwd", _ptr(_f32c(gy)), _ptr(gx), B * C, H, W, _stream())
        return gx


class ConvTranspose2x2Fn(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel_size=2, stride=2)."""

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor]):
        _require_cuda(x, w, b)
        x, w = _f32c(x), _f32c(w.detach())
        b = _f32c(b.detach()) if b is not None else None
        B, Ci, H, W = x.shape
        Co = w.shape[1]
        out = torch.empty((B, Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        _lib.api().call("cfd_convt2_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(out), B, Ci, Co, H, W, _stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        api = _lib.api()
        x, w = ctx.saved_tensors
        B, Ci, H, W = x.shape
        Co = w.shape[1]
        g = _f32c(g)
        gin = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(Co, dtype=torch.float32, device=g.device) if ctx.has_b else None
        ws = _bytes(api.size("cfd_convt2_bwd_workspace_bytes", B, Ci, Co, H, W), g.device)
        api.call("cfd_convt2_bwd", _ptr(g), _ptr(x), _ptr(w), _ptr(gin), _ptr(gw), _ptr(gb), _ptr(ws), B, Ci, Co, H, W, _stream())
        return gin, gw, gb


class ConvTransposeCatFn(torch.autograd.Function):
    """torch.cat([skip, ConvTranspose2d(2, 2)(x)], dim=1) (unet.py:80-88) with the transposed convolution writing straight into the
    second half of the concatenation and its gradient read from there (cfd_convt2_fwd_ex / cfd_convt2_bwd_ex): a pass over the
    up-sampled tensor less in each direction.  ``supported`` tells whether the strided kernels take the shape; otherwise the caller
    uses ConvTranspose2x2Fn + torch.cat."""

    @staticmethod
    def supported(x: Tensor, w: Tensor, skip: Tensor) -> bool:
        import os
        B, Ci, H, W = x.shape
        return (x.is_cuda and skip.is_cuda and skip.dtype == torch.float32 and x.dtype == torch.float32 and w.dtype == torch.float32
                and tuple(skip.shape[2:]) == (2 * H, 2 * W) and skip.shape[0] == B and W % 4 == 0 and (H * W) % 8 == 0
                and w.data_ptr() % 16 == 0 and w.is_contiguous() and os.environ.get("CFD_CONVT_MFMA", "1") != "0")

    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, b: Optional[Tensor], skip: Tensor):
        _require_cuda(x, w, b, skip)
        x, w = _f32c(x), _f32c(w.detach())
        b = _f32c(b.detach()) if b is not None else None
        B, Ci, H, W = x.shape
        Co, C2 = w.shape[1], skip.shape[1]
        out = torch.empty((B, C2 + Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        out[:, :C2].copy_(skip)
        plane = 4 * H * W
        _lib.api().call("cfd_convt2_fwd_ex", _ptr(x), _ptr(w), _ptr(b), out.data_ptr() + 4 * C2 * plane, (C2 + Co) * plane, B, Ci, Co, H, W,
                        _stream())
        ctx.save_for_backward(x, w)
        ctx.meta = (b is not None, C2)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        api = _lib.api()
        x, w = ctx.saved_tensors
        has_b, C2 = ctx.meta
        B, Ci, H, W = x.shape
        Co = w.shape[1]
        g = _f32c(g)
        plane = 4 * H * W
        gin = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w)
        gb = torch.empty(Co, dtype=torch.float32, device=g.device) if has_b else None
        ws = _bytes(api.size("cfd_convt2_bwd_workspace_bytes", B, Ci, Co, H, W), g.device)
        api.call("cfd_convt2_bwd_ex", g.data_ptr() + 4 * C2 * plane, (C2 + Co) * plane, _ptr(x), _ptr(w), _ptr(gin), _ptr(gw), _ptr(gb),
                 _ptr(ws), B, Ci, Co, H, W, _stream())
        # the skip connection's share stays a view of g: MaxPoolSkipFn reads it in place (any other consumer takes the strided tensor)
        gskip = g[:, :C2] if ctx.needs_input_grad[3] else None
        return gin, gw, gb, gskip


class ResidualMaskFn(torch.autograd.Function):
    """(x + resid[:, :C]) * mask  (unet.py:206-208).  resid and mask are data (no gradient is taken for them)."""

    @staticmethod
    def forward(ctx, x: Tensor, resid: Optional[Tensor], mask: Optional[Tensor]):
        _require_cuda(x, resid, mask)
        x, resid, mask = _f32c(x), _f32c(resid), _f32c(mask)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        _lib.api().call("cfd_residual_mask", _ptr(x), _ptr(resid), _ptr(mask), _ptr(out), B, C,
                        resid.shape[1] if resid is not None else C, H * W, _stream())
        ctx.save_for_backward(mask)
        ctx.has_mask = mask is not None
        ctx.resid_shape = None if resid is None else tuple(resid.shape)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        g = _f32c(g)
        gx = g
        if ctx.has_mask:
            (mask,) = ctx.saved_tensors
            B, C, H, W = g.shape
            gx = torch.empty_like(g)
            _lib.api().call("cfd_residual_mask", _ptr(g), None, _ptr(mask), _ptr(gx), B, C, C, H * W, _stream())
        gres = None
        if ctx.resid_shape is not None and ctx.needs_input_grad[1]:  # the residual frame's own gradient (copy of gx)
            gres = torch.zeros(ctx.resid_shape, dtype=torch.float32, device=g.device)
            gres[:, :g.shape[1]] = gx
        return gx, gres, None


class ChannelBiasAddFn(torch.autograd.Function):
    """x + s[:, :, None, None]  (UNet with ``insert_case_params_at="hidden"``, unet.py:198-204): the per-(sample, channel)
    conditioning vector added to the bottleneck.  Forward = the residual-add kernel on the broadcast (a copy, pure data
    movement); backward: gx = g, gs[b, c] = sum over pixels of g -- the row-dot kernel against a row of ones."""

    @staticmethod
    def forward(ctx, x: Tensor, s: Tensor):
        _require_cuda(x, s)
        x, s = _f32c(x), _f32c(s)
        B, C, H, W = x.shape
        se = s.reshape(B, C, 1, 1).expand(B, C, H, W).contiguous()
        out = torch.empty_like(x)
        _lib.api().call("cfd_residual_mask", _ptr(x), _ptr(se), None, _ptr(out), B, C, C, H * W, _stream())
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        g = _f32c(g)
        B, C, H, W = g.shape
        ones = torch.ones((B * C, 1, H * W), dtype=torch.float32, device=g.device)
        zero = torch.zeros(1, dtype=torch.float32, device=g.device)
        gs = torch.empty((B * C, 1), dtype=torch.float32, device=g.device)
        _lib.api().call("cfd_rowdot_fwd", _ptr(g), _ptr(ones), _ptr(zero), _ptr(gs), B * C, 1, H * W, _stream())
        return g, gs.reshape(B, C)


class GeluFn(torch.autograd.Function):
    """nn.GELU() (exact erf) as a stand-alone pass (resnet.py:46,77)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        _require_cuda(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        _lib.api().call("cfd_gelu_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        (x,) = ctx.saved_tensors
        gx = torch.empty_like(x)
        _lib.api().call("cfd_gelu_bwd", _ptr(x), _ptr(_f32c(gy)), _ptr(gx), x.numel(), _stream())
        return gx


class DropoutFn(torch.autograd.Function):
    """nn.Dropout in training mode: keep mask from a hash of (seed, index); backward regenerates the same mask."""

    @staticmethod
    def forward(ctx, x: Tensor, p: float, seed: int):
        _require_cuda(x)
        x = _f32c(x)
        y = torch.empty_like(x)
        _lib.api().call("cfd_dropout", _ptr(x), _ptr(y), x.numel(), float(p), int(seed), _stream())
        ctx.meta = (float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        p, seed = ctx.meta
        gy = _f32c(gy)
        gx = torch.empty_like(gy)
        _lib.api().call("cfd_dropout", _ptr(gy), _ptr(gx), gy.numel(), p, seed, _stream())
        return gx, None, None


class DropoutGeluFn(torch.autograd.Function):
    """gelu(dropout(x)) as one pass per direction (ResidualBlock: conv1 -> nn.Dropout -> nn.GELU, resnet.py:70-77); p = 0 (eval
    mode) is the plain GELU.  Value for value GeluFn(DropoutFn(x)); falls back to the two passes for tensors the fused kernel does
    not take (element count not a multiple of four)."""

    @staticmethod
    def _mix64(v: int) -> int:  # splitmix64 finaliser (csrc/pointwise.hip: cfd_mix64)
        v &= 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return v ^ (v >> 31)

    @staticmethod
    def forward(ctx, x: Tensor, p: float, seed: int, step: Optional[Tensor] = None):
        """``step`` (a 0-d int64 CUDA tensor): the stream's step counter on the device; the mask's seed is then
        mix64(seed + step) & (2^48 - 1), formed by the kernel -- a captured train step draws a new mask on every replay.
        The caller hands over a tensor that is NOT modified between this forward and its backward (ResNet.forward passes a per-call
        snapshot of its counter); it is saved through save_for_backward, so an in-place change raises instead of silently
        regenerating a different mask."""
        _require_cuda(x)
        x = _f32c(x)
        ctx.fused = x.numel() % 4 == 0 and x.data_ptr() % 16 == 0
        ctx.step = step if (step is not None and p > 0) else None
        if ctx.step is not None and not ctx.fused:  # (the two-pass fallback takes a host seed: one synchronisation, not capturable)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DropoutGeluFn: the two-pass fallback (element count not a multiple of 4, or an unaligned tensor) reads "
                                   "the step counter on the host and cannot run inside a stream capture; use shapes the fused kernel takes")
            seed = DropoutGeluFn._mix64(int(seed) + int(step.item())) & 0xFFFFFFFFFFFF
            ctx.step = None
        ctx.meta = (float(p), int(seed))
        y = torch.empty_like(x)
        api = _lib.api()
        if ctx.fused and ctx.step is not None:
            api.call("cfd_dropout_gelu_fwd_step", _ptr(x), _ptr(y), x.numel(), float(p), int(seed), _ptr(ctx.step), _stream())
            ctx.save_for_backward(x, ctx.step)  # the counter too: autograd's version check guards it
            ctx.step = True
        elif ctx.fused:
            api.call("cfd_dropout_gelu_fwd", _ptr(x), _ptr(y), x.numel(), float(p), int(seed), _stream())
            ctx.save_for_backward(x)
        else:
            d = x
            if p > 0:
                d = torch.empty_like(x)
                api.call("cfd_dropout", _ptr(x), _ptr(d), x.numel(), float(p), int(seed), _stream())
            api.call("cfd_gelu_fwd", _ptr(d), _ptr(y), x.numel(), _stream())
            ctx.save_for_backward(d)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x = ctx.saved_tensors[0]
        p, seed = ctx.meta
        gy = _f32c(gy)
        gx = torch.empty_like(x)
        api = _lib.api()
        if ctx.step is not None:
            step = ctx.saved_tensors[1]
            if gy.data_ptr() % 16 != 0:
                gy = gy.clone()  # (a fresh allocation is 16-byte aligned)
            api.call("cfd_dropout_gelu_bwd_step", _ptr(x), _ptr(gy), _ptr(gx), x.numel(), p, seed, _ptr(step), _stream())
            return gx, None, None, None
        if ctx.fused and gy.data_ptr() % 16 == 0:
            api.call("cfd_dropout_gelu_bwd", _ptr(x), _ptr(gy), _ptr(gx), x.numel(), p, seed, _stream())
            return gx, None, None, None
        if ctx.fused:  # (the saved tensor is x, not dropout(x))
            d = x
            if p > 0:
                d = torch.empty_like(x)
                api.call("cfd_dropout", _ptr(x), _ptr(d), x.numel(), p, seed, _stream())
            x = d
        api.call("cfd_gelu_bwd", _ptr(x), _ptr(gy), _ptr(gx), x.numel(), _stream())
        if p > 0:
            g2 = torch.empty_like(gx)
            api.call("cfd_dropout", _ptr(gx), _ptr(g2), gx.numel(), p, seed, _stream())
            gx = g2
        return gx, None, None, None


class AddFn(torch.autograd.Function):
    """x + y for equal shapes (the skip connection of a ResidualBlock, resnet.py:79)."""

    @staticmethod
    def forward(ctx, x: Tensor, y: Tensor):
        _require_cuda(x, y)
        x, y = _f32c(x), _f32c(y)
        B, C = x.shape[0], x.shape[1]
        out = torch.empty_like(x)
        _lib.api().call("cfd_residual_mask", _ptr(x), _ptr(y), None, _ptr(out), B, C, C, x[0, 0].numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        return g, g


# ----------------------------------------------------------------------------------------------------
# NormAct and the non-autoregressive DeepONet pieces  (src/models/act_fn.py:21-47, src/models/deeponet.py:184-205)
# ----------------------------------------------------------------------------------------------------
class NormActFn(torch.autograd.Function):
    """Per-sample (all dims but the first) normalise -> activation -> de-normalise."""

    @staticmethod
    def forward(ctx, x: Tensor, act: int):
        _require_cuda(x)
        x = _f32c(x)
        S = x.shape[0]
        L = x[0].numel()
        y = torch.empty_like(x)
        stats = torch.empty((S, 2), dtype=torch.float32, device=x.device)
        _lib.api().call("cfd_normact_fwd", _ptr(x), _ptr(y), _ptr(stats), S, L, act, _stream())
        ctx.save_for_backward(x, stats)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, stats = ctx.saved_tensors
        gx = torch.empty_like(x)
        _lib.api().call("cfd_normact_bwd", _ptr(x), _ptr(_f32c(gy)), _ptr(stats), _ptr(gx), x.shape[0], x[0].numel(), ctx.act,
                        _stream())
        return gx, None


class BcastAddFn(torch.autograd.Function):
    """(b,p) + (k,p) -> (b,k,p)  (deeponet.py:190-192)."""

    @staticmethod
    def forward(ctx, ft: Tensor, fxy: Tensor):
        _require_cuda(ft, fxy)
        ft, fxy = _f32c(ft), _f32c(fxy)
        B, P = ft.shape
        K = fxy.shape[0]
        out = torch.empty((B, K, P), dtype=torch.float32, device=ft.device)
        _lib.api().call("cfd_bcast_add_fwd", _ptr(ft), _ptr(fxy), _ptr(out), B, K, P, _stream())
        ctx.dims = (B, K, P)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        B, K, P = ctx.dims
        g = _f32c(g)
        gft = torch.empty((B, P), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        gfxy = torch.empty((K, P), dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        _lib.api().call("cfd_bcast_add_bwd", _ptr(g), _ptr(gft), _ptr(gfxy), B, K, P, _stream())
        return gft, gfxy


class RowDotFn(torch.autograd.Function):
    """preds[b,k] = <branch[b], trunk[b,k]> + bias  (deeponet.py:204-205)."""

    @staticmethod
    def forward(ctx, branch: Tensor, trunk: Tensor, bias: Tensor):
        _require_cuda(branch, trunk, bias)
        branch, trunk, bias = _f32c(branch), _f32c(trunk), _f32c(bias.detach())
        B, K, P = trunk.shape
        preds = torch.empty((B, K), dtype=torch.float32, device=trunk.device)
        _lib.api().call("cfd_rowdot_fwd", _ptr(branch), _ptr(trunk), _ptr(bias), _ptr(preds), B, K, P, _stream())
        ctx.save_for_backward(branch, trunk)
        return preds

    @staticmethod
    def backward(ctx, g: Tensor):
        api = _lib.api()
        branch, trunk = ctx.saved_tensors
        B, K, P = trunk.shape
        g = _f32c(g)
        gbr = torch.empty_like(branch) if ctx.needs_input_grad[0] else None
        gtr = torch.empty_like(trunk) if ctx.needs_input_grad[1] else None
        gb = torch.empty(1, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        ws = _bytes(api.size("cfd_rowdot_bwd_workspace_bytes"), g.device)
        api.call("cfd_rowdot_bwd", _ptr(g), _ptr(branch), _ptr(trunk), _ptr(gbr), _ptr(gtr), _ptr(gb), _ptr(ws), B, K, P, _stream())
        return gbr, gtr, gb
