"""Whole-train-step HIP graph: ``model(**batch) -> loss[name].backward() -> optimizer.step()`` captured once and replayed
(the reference's loop body, src/train_auto.py:231-257, without its per-op host work).  The kernels are enqueued through
the C ABI on PyTorch's capture stream, allocations inside the capture come from the graph's private pool, and the
optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``).  Batches are copied into static input
buffers before each replay; shapes are fixed per instance.

``restore_state=True`` (the harness, ``train_auto --graph 1``): the warm-up steps that precede the capture must not count as
training, so parameters, buffers and optimiser state are snapshotted first and restored IN PLACE afterwards (the captured graph
holds their addresses); optimiser state that the warm-up created is reset to its initial value (zeros)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch: Dict[str, Optional[Tensor]], loss_name: str = "nmse", warmup: int = 3,
                 restore_state: bool = False):
        self.model, self.optimizer, self.loss_name = model, optimizer, loss_name
        self.static = {k: (v.clone() if v is not None else None) for k, v in example_batch.items()}
        self.shapes = {k: (tuple(v.shape) if v is not None else None) for k, v in example_batch.items()}
        snap = None
        if restore_state:
            snap = ([t.detach().clone() for t in list(model.parameters()) + list(model.buffers())],
                    {id(t): t.detach().clone() for st in optimizer.state.values() for t in st.values() if torch.is_tensor(t)})
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
        if snap is not None:
            with torch.no_grad():
                for t, saved in zip(list(model.parameters()) + list(model.buffers()), snap[0]):
                    t.copy_(saved)
                for st in optimizer.state.values():
                    for t in st.values():
                        if torch.is_tensor(t):
                            t.copy_(snap[1][id(t)]) if id(t) in snap[1] else t.zero_()
            optimizer.zero_grad(set_to_none=False)  # the captured backward accumulates into these gradient tensors

    def matches(self, batch: Dict[str, Optional[Tensor]]) -> bool:
        """True when ``batch`` has the captured shapes (a short last batch of an epoch has not: run it eagerly)."""
        return all((v is None and self.shapes.get(k) is None) or (v is not None and tuple(v.shape) == self.shapes.get(k))
                   for k, v in batch.items())

    def __call__(self, **batch) -> Dict[str, Tensor]:
        """One optimisation step on ``batch`` (same shapes as the example); returns the static loss tensors."""
        for k, v in batch.items():
            dst = self.static.get(k)
            if dst is not None and v is not None and v.data_ptr() != dst.data_ptr():
                dst.copy_(v, non_blocking=True)
        self.graph.replay()
        return self.loss
