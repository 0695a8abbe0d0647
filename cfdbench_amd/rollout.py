"""Batched multi-step rollout x_{t+1} = Fno2d(x_t) (Fno2d.generate_many, src/models/fno/fno2d.py:269-295) replayed from
ONE HIP graph: every step writes straight into its slot of a (steps+1, B, c, H, W) frame buffer, so the whole horizon is
a single graph launch with no per-step host work (the reference rebuilds coordinate grids on the host and launches ~40
ATen kernels per step and per case).  Results are bitwise those of ``Fno2d.generate_many`` (same kernels, same order).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._capi import FnoShape
from .functional import _creal, _param_struct


ACT_DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


class FnoRollout:
    """``dtype`` = storage type of the activations between kernels: "f32" (bitwise ``Fno2d.generate_many``) or "bf16"
    (BASELINE.json configs[4]: the lifting layer's output and every FnoBlock's pre-activation are rounded to bf16 when
    stored -- half the activation traffic; frames, weights, kept modes and all arithmetic stay fp32)."""

    def __init__(self, model, dtype: str = "f32"):
        if dtype not in ACT_DTYPES:
            raise ValueError(f"dtype must be one of {sorted(ACT_DTYPES)}")
        self.model = model
        self.dtype = dtype
        self.act_dtype = ACT_DTYPES[dtype]
        self.api = _lib.api()
        self._cache: Dict[Tuple, dict] = {}

    def _build(self, B: int, c: int, H: int, W: int, P: int, steps: int, has_mask: bool, device) -> dict:
        cfg = self.model.abi_config()
        if c != cfg["out_chan"]:
            raise RuntimeError("rollout feeds predictions back as inputs: in_chan must equal out_chan")
        plan = _lib.plan(H, W, cfg["modes1"], cfg["modes2"], device.index)
        shape = FnoShape(B, H, W, c, cfg["out_chan"], P, cfg["hidden"], cfg["num_layers"], cfg["modes1"], cfg["modes2"],
                         cfg["head"])
        st = dict(plan=plan, shape=shape,
                  frames=torch.empty((steps + 1, B, c, H, W), dtype=torch.float32, device=device),
                  cp=torch.empty((B, P), dtype=torch.float32, device=device),
                  mask=torch.empty((B, 1, H, W), dtype=torch.float32, device=device) if has_mask else None,
                  ws=torch.empty(max(self.api.size("cfd_fno_workspace_bytes_ex", plan, ctypes.byref(shape), 0, self.act_dtype), 16),
                                 dtype=torch.uint8, device=device))
        # parameters are read through their current storage: re-capture if they are re-allocated (e.g. .to())
        flat = [(_creal(p.detach()) if p.is_complex() else p.detach().contiguous()) for p in self.model.abi_parameters()]
        st["flat"] = flat
        st["pstruct"] = _param_struct([t.data_ptr() for t in flat], cfg["num_layers"])

        def run():
            s = torch.cuda.current_stream().cuda_stream
            for t in range(steps):
                self.api.call("cfd_fno_forward_ex", plan, ctypes.byref(shape), ctypes.byref(st["pstruct"]),
                              st["frames"][t].data_ptr(), st["cp"].data_ptr(),
                              None if st["mask"] is None else st["mask"].data_ptr(), None,
                              st["frames"][t + 1].data_ptr(), None, st["ws"].data_ptr(), 0, self.act_dtype, s)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            st["frames"][0].zero_()
            st["cp"].zero_()
            if st["mask"] is not None:
                st["mask"].fill_(1.0)
            run()  # warm-up outside capture (plan tables, module load)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        from .graph import CAPTURE_MODE
        with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
            run()
        st["graph"] = g
        return st

    @torch.no_grad()
    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor], steps: int) -> List[Tensor]:
        """Same argument conventions and return value as Fno2d.generate_many (unbatched inputs get a batch dimension):
        ``steps`` frames, copied out of the graph's frame buffer in ONE device copy."""
        frames = self.generate_frames(inputs, case_params, mask, steps)
        return list(frames[1:].clone().unbind(0))

    @torch.no_grad()
    def generate_frames(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor], steps: int) -> Tensor:
        """The rollout as the graph's own (steps + 1, B, c, H, W) frame buffer (frame 0 = the input): no copy at all.  The
        buffer is overwritten by the next call with the same shapes."""
        assert len(inputs.shape) == len(case_params.shape) + 2
        if inputs.dim() == 3:
            inputs, case_params = inputs.unsqueeze(0), case_params.unsqueeze(0)
            mask = mask.unsqueeze(0) if mask is not None else None
        if mask is not None and mask.dim() == 3:
            mask = mask.unsqueeze(1)
        if not inputs.is_cuda:
            raise RuntimeError("cfdbench_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback)")
        B, c, H, W = inputs.shape
        ptrs = tuple(p.data_ptr() for p in self.model.abi_parameters())
        key = (B, c, H, W, case_params.shape[1], steps, mask is not None, inputs.device.index, ptrs)
        st = self._cache.get(key)
        if st is None:
            st = self._cache[key] = self._build(B, c, H, W, case_params.shape[1], steps, mask is not None, inputs.device)
        st["frames"][0].copy_(inputs)
        st["cp"].copy_(case_params)
        if mask is not None:
            st["mask"].copy_(mask)
        st["graph"].replay()
        return st["frames"]
