"""Fused training engine for Auto-FNO: the reference's step
    model(**batch) -> loss["nmse"].backward() -> Adam.step() -> zero_grad()      (src/train_auto.py:231-257)
as three C-ABI calls on one flat parameter / gradient / moment buffer, plus (world_size > 1) one RCCL all-reduce of
the flat gradient between backward and Adam.  One process per GPU; nothing here syncs with the host.

The engine re-points the model's parameters at slices of the flat buffer, so ``model.state_dict()`` (reference
key names, complex64 spectral weights) always reflects the trained weights and checkpoints stay interchangeable.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import _lib
from ._capi import FnoShape
from .functional import _param_struct

_LOSS_IDS = {"mse": 0, "nmse": 1, "mae": 2}
_ACT_DTYPES = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


def flatten_layout(params: Sequence[Tensor], align: int = 4) -> Tuple[List[int], int]:
    """Offsets (in floats) of each tensor inside one flat fp32 buffer; complex tensors take 2 floats per element."""
    offs, off = [], 0
    for p in params:
        n = p.numel() * (2 if p.is_complex() else 1)
        offs.append(off)
        off += (n + align - 1) // align * align
    return offs, off


class FlatParams:
    """One contiguous fp32 buffer holding every parameter; the module's Parameters become views of it."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = list(params)
        dev = self.params[0].device
        self.offsets, self.numel = flatten_layout(self.params)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views: List[Tensor] = []
        self.grad_views: List[Tensor] = []
        for p, off in zip(self.params, self.offsets):
            n = p.numel() * (2 if p.is_complex() else 1)
            sl = self.data[off:off + n]
            gsl = self.grad[off:off + n]
            if p.is_complex():
                sl.view(*p.shape, 2).copy_(torch.view_as_real(p.detach().contiguous()))
                p.data = torch.view_as_complex(sl.view(*p.shape, 2))
            else:
                sl.view(p.shape).copy_(p.detach())
                p.data = sl.view(p.shape)
            self.views.append(sl)
            self.grad_views.append(gsl)

    def ptrs(self, grad: bool = False) -> List[int]:
        return [(g if grad else v).data_ptr() for v, g in zip(self.views, self.grad_views)]


# The process group the data-parallel exchanges use when none is given: the default group, or the one
# harness/dist_util.overlapping_comm_group picked (a communicator whose stream really overlaps the compute stream).
_dp_group = None


def set_default_group(group) -> None:
    global _dp_group
    _dp_group = group


def _resolve_group(group):
    if group is not None or _dp_group is None:
        return group
    try:  # (a group of a process group that has been destroyed since is forgotten)
        dist.get_world_size(_dp_group)
        return _dp_group
    except Exception:  # noqa: BLE001
        set_default_group(None)
        return None


class GradSync:
    """Data-parallel gradient exchange: SUM all-reduce of the flat gradient buffer (RCCL over xGMI on GPUs, gloo in
    the CPU tests) in ``n_buckets`` contiguous slices; the 1/world_size factor is folded into the Adam kernel.
    Semantics = DistributedDataParallel's (per-rank loss normalisers, averaged gradients)."""

    def __init__(self, group=None, n_buckets: int = 1):
        live = dist.is_available() and dist.is_initialized()
        group = _resolve_group(group) if live else group
        self.group = group
        self.world = dist.get_world_size(group) if live else 1
        self.n_buckets = max(1, n_buckets)
        # A one-rank group has nothing to exchange and skips the collectives -- unless CFDBENCH_DP_ALWAYS_EXCHANGE=1 asks for them:
        # that is how a one-GPU box runs the RCCL path itself (communicator stream, event ordering, slices of the flat buffer
        # reduced while later backward phases still write it; tests/test_gpu_dp.py).  A SUM over one rank is the identity.
        self.exchange = self.world > 1 or (live and os.environ.get("CFDBENCH_DP_ALWAYS_EXCHANGE", "0") == "1")
        # RCCL ("nccl") reduces device buffers on its own stream, ordered by events: the asynchronous per-slice exchange below.
        # Any other backend (gloo: the CPU tests, and the 2-process tests that share one GPU) gets device slices through an
        # explicit, stream-ordered host staging in wait_all -- correct by construction, no overlap.
        self.device_native = self.exchange and dist.get_backend(group) == "nccl"

    def bucket_slices(self, numel: int) -> List[Tuple[int, int]]:
        per = (numel + self.n_buckets - 1) // self.n_buckets
        per = (per + 3) // 4 * 4
        return [(s, min(numel, s + per)) for s in range(0, numel, per)]

    def reduce_slice_async(self, flat_grad: Tensor, start: int, stop: int):
        """Start the SUM all-reduce of flat_grad[start:stop] (ordered after the work already enqueued on the current
        stream, running on the communicator's own stream); None when there is nothing to exchange."""
        if not self.exchange or stop <= start:
            return None
        if flat_grad.is_cuda and not self.device_native:
            return ("host", flat_grad, start, stop)
        return dist.all_reduce(flat_grad[start:stop], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_all(self, handles) -> float:
        """Make the current stream wait for the started reductions; returns the 1/world scale for Adam."""
        for h in handles:
            if isinstance(h, tuple):  # host-staged slice: D2H (ordered after every kernel enqueued so far), reduce, H2D
                _, flat, a, b = h
                t = flat[a:b].cpu()
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                flat[a:b].copy_(t)
            elif h is not None:
                h.wait()
        return 1.0 / self.world

    def all_reduce(self, flat_grad: Tensor) -> float:
        """Returns the scale to apply to the reduced gradient (1/world)."""
        if not self.exchange:
            return 1.0
        return self.wait_all([self.reduce_slice_async(flat_grad, a, b) for a, b in self.bucket_slices(flat_grad.numel())])


def backward_phase_slices(offsets: Sequence[int], numel: int, num_layers: int) -> List[Tuple[int, int]]:
    """[start, stop) of the flat gradient buffer written by backward phase k (``cfd_fno_backward_phase``): phase 0 = head
    (fc1, fc2: the last four tensors), phase 1 .. L = FnoBlock L-k (four tensors each), phase L+1 = fc0 (the first two).
    ``Fno2d.abi_parameters`` order is fc0, blocks 0 .. L-1, fc1, fc2, so every phase owns one contiguous slice and the
    slices tile the buffer."""
    end = lambda i: offsets[i + 1] if i + 1 < len(offsets) else numel
    sl = [(offsets[2 + 4 * num_layers], numel)]
    for k in range(1, num_layers + 1):
        l = num_layers - k
        sl.append((offsets[2 + 4 * l], end(2 + 4 * l + 3)))
    sl.append((0, end(1)))
    return sl


class FlatGradExchange:
    """Data-parallel gradient exchange of the AUTOGRAD training paths (U-Net, ResNet, the DeepONet family; eager or replayed from
    HIP graphs -- graph.GraphedTrainStep): after a backward pass every parameter gradient is packed, pre-scaled by 1 / world, into
    ONE flat fp32 buffer by one launch per 80 tensors (``cfd_scale_copy_multi``; complex gradients as (re, im) pairs), the buffer is
    SUM-all-reduced (RCCL over xGMI on GPUs; gloo in the CPU tests, staged through the host for device tensors like GradSync), and
    the parameters' ``.grad`` are re-pointed at views of the buffer, so the optimizer reads the reduced values where they are: no
    scatter-back.  DistributedDataParallel semantics (per-rank loss normalisers, averaged gradients).  (Rounds 1-4: ``torch.cat`` +
    all-reduce + ``mul_`` + one ``copy_`` launch per tensor, eager only.)"""

    def __init__(self, params: Sequence[torch.nn.Parameter], group=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradExchange: no trainable parameters")
        self.sync = GradSync(group)
        self.world = self.sync.world
        dev = self.params[0].device
        self.offsets, self.numel = flatten_layout(self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views: List[Tensor] = []  # what the optimizer reads: one view per parameter, the parameter's shape and dtype
        self._real_views: List[Tensor] = []
        for p_, off in zip(self.params, self.offsets):
            n = p_.numel() * (2 if p_.is_complex() else 1)
            sl = self.flat[off:off + n]
            self._real_views.append(sl)
            self.views.append(torch.view_as_complex(sl.view(*p_.shape, 2)) if p_.is_complex() else sl.view(p_.shape))
        self._table = None  # (key, ctypes tables) of the last pack

    def _sources(self) -> List[Tensor]:
        src = []
        for p_ in self.params:
            g = p_.grad
            if g is None:
                raise RuntimeError("FlatGradExchange.pack: a trainable parameter has no gradient (every parameter must take part in the "
                                   "loss: the flat buffer is exchanged whole)")
            g = torch.view_as_real(g) if g.is_complex() else g
            src.append(g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous())
        return src

    def pack(self) -> None:
        """``.grad`` of every parameter (wherever autograd left it) -> the flat buffer, times 1 / world."""
        src = self._sources()
        scale = 1.0 / self.world
        if self.flat.is_cuda:
            key = tuple(t.data_ptr() for t in src)
            if self._table is None or self._table[0] != key:
                col = lambda vals, ty: (ty * len(vals))(*vals)  # noqa: E731
                self._table = (key, (len(src), col(list(key), ctypes.c_void_p), col([v.data_ptr() for v in self._real_views], ctypes.c_void_p),
                                     col([t.numel() for t in src], ctypes.c_size_t)))
            n, sp, dp, nn = self._table[1]
            self._keep = src  # (temporaries of non-contiguous gradients live until the next pack)
            _lib.api().call("cfd_scale_copy_multi", n, sp, dp, nn, float(scale), torch.cuda.current_stream().cuda_stream)
        else:  # host tensors (the gloo tests of the exchange logic)
            with torch.no_grad():
                for d, t in zip(self._real_views, src):
                    d.copy_(t.reshape(-1)).mul_(scale)

    def reduce(self) -> None:
        """SUM all-reduce of the flat buffer (ordered after the work enqueued on the current stream)."""
        self.sync.wait_all([self.sync.reduce_slice_async(self.flat, a, b) for a, b in self.sync.bucket_slices(self.numel)])

    def install(self) -> None:
        """``p.grad`` = the parameter's view of the flat buffer (what ``optimizer.step()`` then reads)."""
        for p_, v in zip(self.params, self.views):
            p_.grad = v

    def exchange(self) -> None:
        self.pack()
        self.reduce()
        self.install()


def sync_gradients(params: Sequence[torch.nn.Parameter], group=None) -> None:
    """Data-parallel gradient averaging for the eager autograd training path (any model): FlatGradExchange on the parameters that
    received a gradient -- one pack launch per 80 tensors, ONE all-reduce, ``.grad`` left as views of the reduced flat buffer.
    Same semantics as DistributedDataParallel.  No-op outside a process group / in a one-rank group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(_resolve_group(group)) == 1 and os.environ.get("CFDBENCH_DP_ALWAYS_EXCHANGE", "0") != "1":
        return
    ps = [p_ for p_ in params if p_.requires_grad and p_.grad is not None]
    if not ps:
        return
    # One exchange object per (parameter set, group, world, backend), hung off the FIRST parameter so that it dies with the model
    # (a module-global dict kept every model's parameters and a model-sized flat buffer alive: ADVICE r5), and re-validated against
    # the live process group: after destroy_process_group + a new init in the same process a cached object would carry a stale
    # world size, backend and 1 / world scale.
    group = _resolve_group(group)
    world, backend = dist.get_world_size(group), dist.get_backend(group)
    key = tuple(id(p_) for p_ in ps) + (id(group), world, backend)
    slot = getattr(ps[0], "_cfd_grad_exchange", None)
    ex = slot[1] if slot is not None and slot[0] == key else None
    if ex is None or any(a is not b for a, b in zip(ex.params, ps)) or ex.sync.world != world:
        ex = FlatGradExchange(ps, group)
        ps[0]._cfd_grad_exchange = (key, ex)
    ex.exchange()


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the items owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class FnoTrainEngine:
    def __init__(self, model, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, loss_name: str = "nmse", group=None, grad_buckets: int = 1,
                 overlap: bool = True, fused_head: bool = True, act_dtype: str = "fp32"):
        """``act_dtype`` = "bf16": bf16-storage training (src/args.py:77-80 of the fork's other trainers): the saved
        activations a_0 .. a_L are rounded to bf16 when stored -- half the bytes of what the backward pass re-reads -- and the
        backward pass reads those rounded values; master weights, gradients, accumulation and Adam stay fp32
        (cfd_fno_forward_train_ex / cfd_fno_backward_phase_ex).  Needs the one-pass head (fused_head)."""
        if loss_name not in _LOSS_IDS:
            raise ValueError(f"loss_name must be one of {sorted(_LOSS_IDS)}")
        if act_dtype not in _ACT_DTYPES:
            raise ValueError(f"act_dtype must be one of {sorted(_ACT_DTYPES)}")
        self.act_dtype = _ACT_DTYPES[act_dtype]
        if self.act_dtype and not fused_head:
            raise ValueError("bf16 activation storage runs the one-pass training head: fused_head must be True")
        self.api = _lib.api()
        self.model = model
        params = model.abi_parameters()
        if not params[0].is_cuda:
            raise RuntimeError("FnoTrainEngine: move the model to the GPU first (no CPU fallback)")
        self.device = params[0].device
        self.flat = FlatParams(params)
        self.cfg = model.abi_config()
        self.L = self.cfg["num_layers"]
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.loss_id = _LOSS_IDS[loss_name]
        self.step_count = 0
        self.sync = GradSync(group, grad_buckets)
        self.overlap = overlap  # DP: reduce each backward phase's gradients while the next phase computes
        # the projection head in ONE pass for both directions (cfd_fno_forward_train: the loss is fixed, so its gradient's
        # coefficient is known from the labels before the head runs); False = cfd_fno_forward + cfd_loss_coef + head backward
        self.fused_head = fused_head
        self.pstruct = _param_struct(self.flat.ptrs(False), self.L)
        self.gstruct = _param_struct(self.flat.ptrs(True), self.L)
        # Round 6: the single-GPU step folds its three tiny launches into launches that exist anyway (include/cfdbench_amd.h,
        # CFD_TRAIN_DEFER_*: the nMSE normaliser into Adam, the head's reduction into backward phase 1, the fc0 combine into Adam's
        # launch).  Data-parallel steps keep them: a rank's gradients must be final and normalised by ITS labels before the all-reduce.
        # With the flags, `flat.grad` after train_step holds the gradients of sum d^2 * upstream / n (nmse); `gradients()` rescales.
        self.defer_flags = 0 if (self.sync.exchange or not fused_head) else 7
        self.sums = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.coef = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.scores_buf = torch.zeros(4, dtype=torch.float32, device=self.device)
        self._shape_key = None
        self._graph = None

    # ------------------------------------------------------------------------------------------------
    def _prepare(self, inputs: Tensor, case_params: Tensor):
        B, in_chan, H, W = inputs.shape
        key = (B, in_chan, H, W, case_params.shape[1])
        if key == self._shape_key:
            return
        c = self.cfg
        self.plan = _lib.plan(H, W, c["modes1"], c["modes2"], self.device.index)
        self.shape = FnoShape(B, H, W, in_chan, c["out_chan"], case_params.shape[1], c["hidden"], self.L, c["modes1"],
                              c["modes2"], c["head"])
        nbytes = self.api.size("cfd_fno_workspace_bytes_ex", self.plan, ctypes.byref(self.shape), 1, self.act_dtype)
        if nbytes == 0:
            raise RuntimeError("cfd_fno_workspace_bytes_ex returned 0 for this shape / activation type")
        self.ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)
        self.preds = torch.empty((B, c["out_chan"], H, W), dtype=torch.float32, device=self.device)
        self._shape_key = key
        self._graph = None

    def forward_backward(self, inputs: Tensor, label: Tensor, case_params: Tensor, mask: Optional[Tensor] = None, flags: int = 0):
        """Enqueue forward + loss + backward into the flat gradient buffer (every gradient is overwritten).  ``flags`` =
        CFD_TRAIN_DEFER_* (train_step passes ``defer_flags``): the gradients are then complete only after ``optimizer_step``'s launch."""
        for t in (inputs, label, case_params, mask):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError("FnoTrainEngine expects contiguous float32 CUDA tensors")
        self._prepare(inputs, case_params)
        st = torch.cuda.current_stream().cuda_stream
        a, sp = self.api, ctypes.byref(self.shape)
        mp = None if mask is None else mask.data_ptr()
        self._last = (inputs, case_params, mask, flags)  # optimizer_step finishes what the flags deferred
        if self.fused_head:
            a.call("cfd_fno_forward_train_f", self.plan, sp, ctypes.byref(self.pstruct), ctypes.byref(self.gstruct),
                   inputs.data_ptr(), case_params.data_ptr(), mp, label.data_ptr(), self.preds.data_ptr(), self.sums.data_ptr(),
                   self.coef.data_ptr(), self.ws.data_ptr(), self.loss_id, 1.0, self.act_dtype, flags, st)
            for phase in range(1, self.L + 2):  # phase 0 (the head) left the fused kernel already
                a.call("cfd_fno_backward_phase_f", self.plan, sp, ctypes.byref(self.pstruct), ctypes.byref(self.gstruct),
                       inputs.data_ptr(), case_params.data_ptr(), mp, label.data_ptr(), self.preds.data_ptr(), None,
                       self.coef.data_ptr(), self.sums.data_ptr(), self.ws.data_ptr(), phase, self.loss_id, self.act_dtype, flags, st)
            return
        if flags:
            raise RuntimeError("FnoTrainEngine: the deferred training step needs the one-pass head (fused_head)")
        a.call("cfd_fno_forward", self.plan, sp, ctypes.byref(self.pstruct), inputs.data_ptr(), case_params.data_ptr(), mp,
               label.data_ptr(), self.preds.data_ptr(), self.sums.data_ptr(), self.ws.data_ptr(), 1, st)
        a.call("cfd_loss_coef", self.sums.data_ptr(), self.coef.data_ptr(), self.loss_id, 1.0, st)
        a.call("cfd_fno_backward", self.plan, sp, ctypes.byref(self.pstruct), ctypes.byref(self.gstruct), inputs.data_ptr(),
               case_params.data_ptr(), mp, label.data_ptr(), self.preds.data_ptr(), None, self.coef.data_ptr(),
               self.ws.data_ptr(), st)

    # ---- data-parallel step: backward phase by phase, each phase's gradient slice reduced while the next computes ----
    def phase_slices(self) -> List[Tuple[int, int]]:
        """[start, stop) of the flat gradient buffer written by backward phase k (cfd_fno_backward_phase): phase 0 = head
        (fc1, fc2: the last four tensors), phase 1 .. L = FnoBlock L-k (four tensors each), phase L+1 = fc0 (the first
        two).  ``abi_parameters`` order is fc0, blocks 0 .. L-1, fc1, fc2, so every phase owns one contiguous slice."""
        return backward_phase_slices(self.flat.offsets, self.flat.numel, self.L)

    def forward_backward_overlapped(self, inputs: Tensor, label: Tensor, case_params: Tensor,
                                    mask: Optional[Tensor] = None) -> float:
        """forward_backward with the gradient exchange overlapped: after each backward phase is enqueued, the SUM
        all-reduce of that phase's (final) gradient slice is started asynchronously -- RCCL runs it on its own stream
        while the following phases compute (the head's 12 KB first, then one 0.9-MB bucket per FnoBlock, during the next
        block's DFT / mix / fused block kernels).  Returns the 1/world scale for Adam.  Bitwise the same gradients as
        forward_backward + GradSync.all_reduce (same kernels, same reduction per element)."""
        for t in (inputs, label, case_params, mask):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise RuntimeError("FnoTrainEngine expects contiguous float32 CUDA tensors")
        self._prepare(inputs, case_params)
        self._last = (inputs, case_params, mask, 0)  # (a data-parallel pass defers nothing: optimizer_step runs the plain flat Adam)
        st = torch.cuda.current_stream().cuda_stream
        a, sp = self.api, ctypes.byref(self.shape)
        mp = None if mask is None else mask.data_ptr()
        if self.fused_head:
            a.call("cfd_fno_forward_train_ex", self.plan, sp, ctypes.byref(self.pstruct), ctypes.byref(self.gstruct),
                   inputs.data_ptr(), case_params.data_ptr(), mp, label.data_ptr(), self.preds.data_ptr(), self.sums.data_ptr(),
                   self.coef.data_ptr(), self.ws.data_ptr(), self.loss_id, 1.0, self.act_dtype, st)
        else:
            a.call("cfd_fno_forward", self.plan, sp, ctypes.byref(self.pstruct), inputs.data_ptr(), case_params.data_ptr(), mp,
                   label.data_ptr(), self.preds.data_ptr(), self.sums.data_ptr(), self.ws.data_ptr(), 1, st)
            a.call("cfd_loss_coef", self.sums.data_ptr(), self.coef.data_ptr(), self.loss_id, 1.0, st)
        handles = []
        for phase, (s0, s1) in enumerate(self.phase_slices()):
            if not (self.fused_head and phase == 0):  # the fused forward has produced the head's gradients already
                a.call("cfd_fno_backward_phase_ex", self.plan, sp, ctypes.byref(self.pstruct), ctypes.byref(self.gstruct),
                       inputs.data_ptr(), case_params.data_ptr(), mp, label.data_ptr(), self.preds.data_ptr(), None,
                       self.coef.data_ptr(), self.ws.data_ptr(), phase, self.act_dtype, st)
            handles.append(self.sync.reduce_slice_async(self.flat.grad, s0, s1))
        return self.sync.wait_all(handles)

    def optimizer_step(self, grad_scale: float = 1.0):
        self.step_count += 1
        inputs, case_params, mask, flags = getattr(self, "_last", (None, None, None, 0))
        if flags:  # the pass left work to this launch: nMSE normaliser from sums[2:4], the lifting layer's gradient rows
            self.api.call("cfd_fno_adam_step", self.plan, ctypes.byref(self.shape), ctypes.byref(self.pstruct), ctypes.byref(self.gstruct),
                          inputs.data_ptr(), case_params.data_ptr(), None if mask is None else mask.data_ptr(), self.sums.data_ptr(),
                          self.ws.data_ptr(), self.flat.data.data_ptr(), self.flat.grad.data_ptr(), self.exp_avg.data_ptr(),
                          self.exp_avg_sq.data_ptr(), self.flat.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                          self.step_count, grad_scale, self.loss_id, self.act_dtype, flags, torch.cuda.current_stream().cuda_stream)
            self._last = (inputs, case_params, mask, 0)  # (a second optimizer_step on the same gradients must not redo the deferred work)
            self._grad_pending_scale = bool(flags & 1) and self.loss_id == _LOSS_IDS["nmse"]
            return
        self._grad_pending_scale = False
        self.api.call("cfd_adam_flat", self.flat.data.data_ptr(), self.flat.grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), self.flat.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.weight_decay, self.step_count, grad_scale, torch.cuda.current_stream().cuda_stream)

    def gradients(self) -> Tensor:
        """The flat gradient of the last step in the loss's own units.  After a deferred train_step (``defer_flags``) the buffer holds the
        gradients of sum d^2 / n -- this returns them times n / sum (label*mask)^2 (what Adam applied); otherwise the buffer itself."""
        if getattr(self, "_grad_pending_scale", False):
            return self.flat.grad * (self.sums[3] / self.sums[2])
        return self.flat.grad

    def train_step(self, inputs: Tensor, label: Tensor, case_params: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        """One optimisation step; returns the device tensor [sum sq err, sum abs err, sum sq label, n] of THIS rank's
        batch (no host sync -- read it with .tolist() only when logging)."""
        if self.sync.exchange and self.overlap:
            scale = self.forward_backward_overlapped(inputs, label, case_params, mask)
        else:
            self.forward_backward(inputs, label, case_params, mask, self.defer_flags)
            scale = self.sync.all_reduce(self.flat.grad)
        self.optimizer_step(scale)
        return self.sums

    # ---- optimiser state for full-state checkpoints (the reference saves weights only) ------------
    def state_dict(self) -> Dict[str, object]:
        """Adam moments (flat, in ``abi_parameters`` order), step count and learning rate.  The weights themselves live in
        the model's ``state_dict`` (the flat buffer is what the model's parameters view)."""
        return dict(exp_avg=self.exp_avg.detach().clone(), exp_avg_sq=self.exp_avg_sq.detach().clone(),
                    step_count=int(self.step_count), lr=float(self.lr), numel=int(self.flat.numel))

    def load_state_dict(self, state: Dict[str, object]) -> None:
        if int(state["numel"]) != int(self.flat.numel):
            raise ValueError(f"optimiser state is for {state['numel']} parameters, the model has {self.flat.numel}")
        self.exp_avg.copy_(state["exp_avg"].to(self.device))
        self.exp_avg_sq.copy_(state["exp_avg_sq"].to(self.device))
        self.step_count = int(state["step_count"])
        self.lr = float(state["lr"])

    def scores(self) -> Dict[str, float]:
        """{mse, rmse, mae, nmse} of the last step (loss.py:27-35).  Synchronises."""
        self.api.call("cfd_loss_scores", self.sums.data_ptr(), self.scores_buf.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
        v = self.scores_buf.tolist()
        return dict(mse=v[0], rmse=v[1], mae=v[2], nmse=v[3])

    # ---- optional HIP-graph replay of forward+backward (static buffers) -------------------------------
    def capture(self, inputs: Tensor, label: Tensor, case_params: Tensor, mask: Optional[Tensor] = None):
        """Capture forward+loss+backward for this batch shape into a HIP graph.  Afterwards ``train_step_graph``
        copies a batch into the static buffers and replays; the all-reduce and Adam stay outside the graph."""
        self._static = [t.clone() if t is not None else None for t in (inputs, label, case_params, mask)]
        self._prepare(inputs, case_params)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.forward_backward(*self._static, flags=self.defer_flags)  # warm-up outside capture
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        from .graph import CAPTURE_MODE
        torch.cuda.synchronize()  # (no collective of an earlier step in flight: see graph.CAPTURE_MODE)
        # Round 6: over RCCL the per-phase all-reduces are captured with the pass (forked onto the communicator's stream and joined back
        # by events), so a replayed data-parallel step makes no host call into the process group; host-staged backends reduce after it.
        self._graph_exchanges = bool(self.sync.exchange and self.sync.device_native and self.overlap
                                     and os.environ.get("CFDBENCH_DP_ONE_GRAPH", "1") != "0")
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            if self._graph_exchanges:
                self._graph_scale = self.forward_backward_overlapped(*self._static)
            else:
                self.forward_backward(*self._static, flags=self.defer_flags)
        self._graph = g

    def train_step_graph(self, inputs: Tensor, label: Tensor, case_params: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        key = (*inputs.shape, case_params.shape[1], mask is not None)
        if self._graph is None or key != getattr(self, "_graph_key", None):  # a new batch shape needs a new capture
            self.capture(inputs, label, case_params, mask)
            self._graph_key = key
        for dst, src in zip(self._static, (inputs, label, case_params, mask)):
            if dst is not None and src is not dst:
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        self._last = (self._static[0], self._static[2], self._static[3], self.defer_flags)  # (the replayed pass used the static buffers)
        scale = self._graph_scale if self._graph_exchanges else self.sync.all_reduce(self.flat.grad)
        self.optimizer_step(scale)
        return self.sums
