"""``torch.optim.Adam`` (amsgrad=False, maximize=False; src/train_auto.py:213) for the autograd training paths, as ONE kernel launch
per 80 parameter tensors (``cfd_adam_multi``).  The reference's optimizer is stock ``torch.optim.Adam``; its fused multi-tensor form
costs 3 x 31 us per U-Net step (136 tensors, 4.4 MB) and 40 us per Auto-DeepONet step (33 tensors) on MI355X -- 7 % of the latter --
where the data would need ~3 us.  Same update rule, same state layout (``step`` as an fp32 device scalar per parameter, ``exp_avg``,
``exp_avg_sq``), so ``state_dict`` / ``load_state_dict``, LR schedulers and ``graph.GraphedTrainStep`` work unchanged; bias
corrections are formed in fp32 on the device like torch's capturable path.  One difference: the parameters of a group that receive
a gradient share ONE step count (the first one's) -- they do in every model here, where a parameter either always or never has one.  Real fp32 CUDA parameters only (the FNO's complex
weights stay with torch's Adam or the fused engine's flat one).  Meant for captured steps (`--graph 1`, bench legs): in an eager
loop its per-step Python (136 parameters: checks, pointer-table lookup) costs more host time than torch's C++ fused path saves on
the GPU -- measured 6.50 vs 6.15 ms per eager U-Net step on a slow-host box -- so the eager default keeps the stock optimizer."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, capturable: bool = True):
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0) or eps < 0 or weight_decay < 0:
            raise ValueError("cfdbench_amd.optim.Adam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=capturable))
        self._tables = {}

    @staticmethod
    def supports(params) -> bool:
        return all(p.is_cuda and p.dtype == torch.float32 and not p.is_complex() for p in params)

    def __getstate__(self):  # (pointer tables are per-process caches)
        d = super().__getstate__()
        d.pop("_tables", None)
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}  # the moment tensors were replaced

    def _table(self, gi, ps):
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps) + (self.state[ps[0]]["exp_avg"].data_ptr(),)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1]
        st = [self.state[p] for p in ps]
        col = lambda vals, ty: (ty * len(vals))(*vals)
        tab = (len(ps), col([p.data_ptr() for p in ps], ctypes.c_void_p), col([p.grad.data_ptr() for p in ps], ctypes.c_void_p),
               col([s["exp_avg"].data_ptr() for s in st], ctypes.c_void_p), col([s["exp_avg_sq"].data_ptr() for s in st], ctypes.c_void_p),
               col([p.numel() for p in ps], ctypes.c_size_t))
        self._tables[gi] = (key, tab)
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        api = _lib.api()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32):
                    raise RuntimeError("cfdbench_amd.optim.Adam: real fp32 CUDA parameters only (use torch.optim.Adam otherwise)")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("cfdbench_amd.optim.Adam: parameters and gradients must be contiguous")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["step"].device != p.device or st["step"].dtype != torch.float32:  # (a state_dict written by a CPU-step Adam)
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
            steps = [self.state[p]["step"] for p in ps]
            torch._foreach_add_(steps, 1.0)
            lr = group["lr"]
            lr_dev = lr if torch.is_tensor(lr) else None
            if lr_dev is not None and not (lr_dev.is_cuda and lr_dev.dtype == torch.float32):
                lr, lr_dev = float(lr_dev), None
            n, pp, gp, mp, vp, nn = self._table(gi, ps)
            api.call("cfd_adam_multi", n, pp, gp, mp, vp, nn, None if lr_dev is None else lr_dev.data_ptr(),
                     0.0 if lr_dev is not None else float(lr), steps[0].data_ptr(), 0.0, float(group["betas"][0]), float(group["betas"][1]),
                     float(group["eps"]), float(group["weight_decay"]), 1.0, torch.cuda.current_stream().cuda_stream)
        return loss
