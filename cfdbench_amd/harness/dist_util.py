"""Data-parallel plumbing shared by the two training entry points (one process per GPU, ``torch.distributed`` over RCCL;
the reference has no distributed path at all: src/train_auto.py:316-381 is a single-process script).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m cfdbench_amd.harness.train_auto --model fno --data cavity_prop_bc_geo --loss_name nmse --fused 1
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed(backend: str | None = None) -> Tuple[int, int]:
    """Join the process group the launcher describes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), bind this process to
    its GPU.  A plain ``python -m ...`` start (no WORLD_SIZE) stays single-process.  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"))
    return dist.get_rank(), dist.get_world_size()


def shard_indices(n: int, rank: int, world: int, batch_size: int = 1, seed: int = 0, epoch: int = 0) -> List[int]:
    """Frames owned by ``rank`` in ``epoch``: a permutation of range(n) seeded with seed + epoch (``DistributedSampler.set_epoch``
    semantics: a rank meets different frames every epoch and all frames over time, although each epoch drops the remainder) cut
    to a multiple of world * batch_size (so that EVERY rank
    sees the same number of frames and, with drop_last, runs the same number of steps -- a rank with one step more would
    pair its gradient all-reduce with the others' barrier) and split into contiguous, equal parts.  When n is smaller
    than world * batch_size only the multiple-of-world cut applies."""
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch)).tolist()
    unit = world * batch_size
    keep = (n // unit) * unit if n >= unit else (n // world) * world
    per = keep // world
    return perm[rank * per:(rank + 1) * per]


def broadcast_model_state(model: torch.nn.Module, src: int = 0) -> None:
    """Parameters AND buffers (BatchNorm running statistics, step counters) of rank ``src`` to every rank, so that the
    replicas start identical whatever each process' RNG did and rank 0's checkpoint speaks for all of them."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        if t.is_complex():
            dist.broadcast(torch.view_as_real(t.data), src)
        else:
            dist.broadcast(t.data, src)


def average_buffers(model: torch.nn.Module) -> None:
    """Mean of the floating-point buffers over the ranks (BatchNorm running_mean / running_var after an epoch of per-rank
    batch statistics -- DistributedDataParallel keeps rank 0's; the mean uses every shard's frames)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    for b in model.buffers():
        if b.is_floating_point():
            dist.all_reduce(b.data, op=dist.ReduceOp.SUM)
            b.data.mul_(1.0 / world)


def check_resume_state(state: dict, fused: bool, world: int) -> None:
    """A saved training state continues only the kind of run that wrote it (optimizer layout and shard sizes differ)."""
    if "fused" in state and bool(state["fused"]) != bool(fused):
        raise RuntimeError(f"train_state.pt was written with --fused {int(state['fused'])}, this run uses --fused {int(fused)}")
    if "world" in state and int(state["world"]) != int(world):
        raise RuntimeError(f"train_state.pt was written by a {state['world']}-process run, this run has {world} processes")
