"""Whole-train-step HIP graph: ``model(**batch) -> loss[name].backward() -> optimizer.step()`` captured once and replayed
(the reference's loop body, src/train_auto.py:231-257, without its per-op host work).  The kernels are enqueued through
the C ABI on PyTorch's capture stream, allocations inside the capture come from the graph's private pool, and the
optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``).  Batches are copied into static input
buffers before each replay; shapes are fixed per instance."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch: Dict[str, Optional[Tensor]], loss_name: str = "nmse", warmup: int = 3):
        self.model, self.optimizer, self.loss_name = model, optimizer, loss_name
        self.static = {k: (v.clone() if v is not None else None) for k, v in example_batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # lazy initialisation (optimizer state, plans, allocator pools) outside the capture
                optimizer.zero_grad(set_to_none=True)
                out = model(**self.static)
                out["loss"][loss_name].backward()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            out = model(**self.static)
            self.loss = {k: v for k, v in out["loss"].items()}
            self.preds = out["preds"]
            self.loss[loss_name].backward()
            optimizer.step()

    def __call__(self, **batch) -> Dict[str, Tensor]:
        """One optimisation step on ``batch`` (same shapes as the example); returns the static loss tensors."""
        for k, v in batch.items():
            dst = self.static.get(k)
            if dst is not None and v is not None and v.data_ptr() != dst.data_ptr():
                dst.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.loss
