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
        steer_comm_stream()
    return dist.get_rank(), dist.get_world_size()


COMM_OVERLAPS = None  # verdict of the last steer_comm_stream() (None: not probed -- no process group / not RCCL)


def steer_comm_stream() -> bool | None:
    """Pick a process group whose communicator stream overlaps the compute stream (overlapping_comm_group below) and make it the one the
    data-parallel exchanges use by default (engine.set_default_group).  Collective; a no-op outside an RCCL process group."""
    global COMM_OVERLAPS
    COMM_OVERLAPS = None
    if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()) or dist.get_backend() != "nccl":
        return None
    from ..engine import set_default_group
    grp, ok = overlapping_comm_group(None)
    set_default_group(grp)
    COMM_OVERLAPS = bool(ok)
    return COMM_OVERLAPS


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


def comm_stream_overlaps(group=None, device=None, work_passes: int = 40) -> bool:
    """Whether the collectives of ``group`` (RCCL, one communicator stream per group and device) run WHILE kernels of the current stream
    execute.  Not a given: HIP maps its streams round-robin onto four hardware queues and a stream that shares the compute stream's
    queue executes behind it (LABNOTES round 5, ``harness/data.py:overlapping_copy_stream``) -- ProcessGroupNCCL draws its stream from the
    same pool, so one placement in four would serialise every gradient all-reduce with the backward pass it is meant to hide behind.
    Probe: ~2 ms of kernels on the current stream; behind the same start event, on a helper stream that was itself probed to overlap, a
    small all-reduce; it overlaps when it finished in under half the kernels' time.  Collective: every rank of the group must call it,
    and every rank gets the same answer (the per-rank verdicts are combined with MIN)."""
    if not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available() or dist.get_backend(group) != "nccl":
        return False
    from .data import overlapping_copy_stream
    cur = torch.cuda.current_stream(device)
    dev = cur.device
    helper, helper_ok = overlapping_copy_stream(dev)
    x = torch.empty(32 << 20, dtype=torch.float32, device=dev)
    t = torch.zeros(1024, dtype=torch.float32, device=dev)
    dist.all_reduce(t, group=group)  # communicator (and its stream) exist
    start, k_end, c_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize(dev)
    start.record(cur)
    for _ in range(work_passes):
        x.mul_(1.0)
    k_end.record(cur)
    helper.wait_event(start)
    with torch.cuda.stream(helper):
        w = dist.all_reduce(t, group=group, async_op=True)  # the communicator's stream waits for `helper` (idle behind `start`), not for `cur`
        w.wait()
        c_end.record(helper)
    torch.cuda.synchronize(dev)
    ok = helper_ok and start.elapsed_time(c_end) < 0.5 * start.elapsed_time(k_end)
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0.5)


def overlapping_comm_group(group=None, device=None, tries: int = 4):
    """A process group over the same ranks as ``group`` whose communicator stream overlaps the compute stream on EVERY rank: ``group``
    itself when ``comm_stream_overlaps`` says so, otherwise up to ``tries`` fresh groups (``dist.new_group``: a new communicator whose
    stream is the pool's next entry, i.e. another hardware queue) until one does.  Returns ``(group, overlaps)``; collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
        return group, False
    if comm_stream_overlaps(group, device):
        return group, True
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
    for _ in range(tries):
        cand = dist.new_group(ranks=ranks, backend="nccl")
        if comm_stream_overlaps(cand, device):
            return cand, True
    return group, False
