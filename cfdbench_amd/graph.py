"""Whole-train-step HIP graph: ``model(**batch) -> loss[name].backward() -> optimizer.step()`` captured once and replayed
(the reference's loop body, src/train_auto.py:231-257, without its per-op host work).  The kernels are enqueued through
the C ABI on PyTorch's capture stream, allocations inside the capture come from the graph's private pool, and the
optimizer must be capturable (``torch.optim.Adam(..., capturable=True)``).  Batches are copied into static input
buffers before each replay; shapes are fixed per instance.

``restore_state=True`` (the harness, ``train_auto --graph 1``): the warm-up steps that precede the capture must not count as
training, so parameters, buffers and optimiser state are snapshotted first and restored IN PLACE afterwards (the captured graph
holds their addresses); optimiser state that the warm-up created is reset to its initial value (zeros).

Data parallel (round 5; SURVEY 8 a-11 / 8e for configs[2] and configs[3], whose models train through autograd): inside a process
group of more than one rank the step is TWO graphs with the exchange between them --
    graph A   forward + backward + one pack launch per 80 tensors: every parameter gradient, pre-scaled by 1 / world, into ONE flat buffer
    exchange  SUM all-reduce of that buffer (RCCL over xGMI; ``engine.FlatGradExchange``), enqueued eagerly between the replays
    graph B   optimizer.step() reading the reduced gradients through ``.grad`` views of the flat buffer
-- DistributedDataParallel semantics, no per-tensor copies, no host synchronisation.  Round 6: over RCCL (backend "nccl") the collective
is CAPTURED with the rest (``one_graph``, the default there; CFDBENCH_DP_ONE_GRAPH=0 restores the two graphs): a step is ONE replay --
forward, backward, pack, all-reduce on the communicator's stream (forked from and joined to the capture by events), optimizer -- and no
host call per step touches the process group.  Host-staged backends (gloo: the CPU tests, two processes on one GPU) keep the two graphs.
Every rank must construct the object and call
it the same number of times (the warm-up steps exchange too, so that replicas which are not restored stay identical).
``capture=False`` runs the same three stages eagerly (batches of another shape; host tensors in the gloo tests of the logic)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor


# Stream captures in this package run in "thread_local" error mode: in the default ("global") mode ANY thread's potentially unsafe HIP call
# fails while a capture is open -- and inside a process group the RCCL watchdog thread polls hipEventQuery on the collectives it still
# tracks.  A capture that began before the watchdog had retired the last all-reduce of the warm-up steps killed the process ("operation
# not permitted when stream is capturing", one run in four of tests/test_gpu_dp.py on the GPU box).  The capturing thread itself keeps
# the strict checks.
CAPTURE_MODE = "thread_local"


class GraphedTrainStep:
    def __init__(self, model, optimizer, example_batch: Dict[str, Optional[Tensor]], loss_name: str = "nmse", warmup: int = 3,
                 restore_state: bool = False, group=None, capture: bool = True, one_graph: Optional[bool] = None):
        import os
        from .engine import GradSync
        self.model, self.optimizer, self.loss_name = model, optimizer, loss_name
        sync = GradSync(group)
        # the all-reduce inside the graph: only where the backend reduces device buffers on its own stream (RCCL)
        self.one_graph = sync.device_native and os.environ.get("CFDBENCH_DP_ONE_GRAPH", "1") != "0" if one_graph is None else bool(one_graph)
        if self.one_graph and not sync.device_native:
            raise ValueError("GraphedTrainStep: one_graph needs a backend that reduces device buffers on its own stream (nccl)")
        self.static = {k: (v.clone() if v is not None else None) for k, v in example_batch.items()}
        self.shapes = {k: (tuple(v.shape) if v is not None else None) for k, v in example_batch.items()}
        # data parallel: the exchange object is made after the first backward pass, over the parameters that received a gradient
        # (the ResNet keeps the reference's unused bn1 / bn2 parameters)
        self.dp, self.group, self.exchange = GradSync(group).exchange, group, None
        self.capture = bool(capture)
        self.graph = self.graph_opt = None
        self.loss = self.preds = None
        if not self.capture:
            return
        if self.dp and warmup < 1:
            raise ValueError("GraphedTrainStep: the data-parallel capture needs at least one warm-up step")
        snap = None
        if restore_state:
            snap = ([t.detach().clone() for t in list(model.parameters()) + list(model.buffers())],
                    {id(t): t.detach().clone() for st in optimizer.state.values() for t in st.values() if torch.is_tensor(t)})
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # lazy initialisation (optimizer state, plans, allocator pools) outside the capture
                self.eager_step(self.static)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        if self.dp:
            torch.cuda.synchronize()  # the warm-up steps' collectives have completed before the capture opens
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_MODE):
            out = model(**self.static)
            self.loss = {k: v for k, v in out["loss"].items()}
            self.preds = out["preds"]
            self.loss[loss_name].backward()
            if self.exchange is None:
                optimizer.step()
            else:
                self.exchange.pack()
                if self.one_graph:  # the collective and the optimizer in the same capture
                    self._raw_grads = [p.grad for p in model.parameters()]  # (the pack launch reads these on every replay)
                    self.exchange.reduce()
                    self.exchange.install()
                    optimizer.step()
        if not (self.exchange is not None and self.one_graph):
            self._raw_grads = [p.grad for p in model.parameters()]
        if self.exchange is not None and not self.one_graph:
            # the pack launch of graph A reads the gradient tensors autograd allocated during the capture (graph-private pool): they
            # must outlive the parameters' .grad, which from here on are views of the flat buffer
            self.exchange.reduce()  # (one eager exchange before the second capture: communicator warm-up on this stream)
            torch.cuda.synchronize()
            self.exchange.install()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode=CAPTURE_MODE):
                optimizer.step()
        if snap is not None:
            with torch.no_grad():
                for t, saved in zip(list(model.parameters()) + list(model.buffers()), snap[0]):
                    t.copy_(saved)
                for st in optimizer.state.values():
                    for t in st.values():
                        if torch.is_tensor(t):
                            t.copy_(snap[1][id(t)]) if id(t) in snap[1] else t.zero_()
            if self.exchange is None:
                optimizer.zero_grad(set_to_none=False)  # the captured backward accumulates into these gradient tensors

    def eager_step(self, batch: Dict[str, Optional[Tensor]]) -> Dict[str, Tensor]:
        """The same optimisation step without a graph (any batch shape): forward, backward, (exchange,) optimizer."""
        # one process, after the capture: the gradient tensors of the capture stay the parameters' .grad (zeroed, accumulated into)
        keep = self.graph is not None and self.exchange is None
        self.optimizer.zero_grad(set_to_none=not keep)
        out = self.model(**batch)
        out["loss"][self.loss_name].backward()
        if self.dp:
            if self.exchange is None:
                from .engine import FlatGradExchange
                self.exchange = FlatGradExchange([p for p in self.model.parameters() if p.grad is not None], self.group)
            self.exchange.exchange()
        self.optimizer.step()
        self._eager_preds = out["preds"]
        return out["loss"]

    def matches(self, batch: Dict[str, Optional[Tensor]]) -> bool:
        """True when ``batch`` has the captured shapes (a short last batch of an epoch has not: run it eagerly)."""
        return all((v is None and self.shapes.get(k) is None) or (v is not None and tuple(v.shape) == self.shapes.get(k))
                   for k, v in batch.items())

    def __call__(self, **batch) -> Dict[str, Tensor]:
        """One optimisation step on ``batch`` (same shapes as the example); returns the static loss tensors."""
        if not self.capture:
            loss = self.eager_step(batch)
            self.preds = self._eager_preds
            return loss
        for k, v in batch.items():
            dst = self.static.get(k)
            if dst is not None and v is not None and v.data_ptr() != dst.data_ptr():
                dst.copy_(v, non_blocking=True)
        self.graph.replay()
        if self.graph_opt is not None:  # two-graph form: the exchange between the replays
            self.exchange.reduce()
            self.graph_opt.replay()
        return self.loss
