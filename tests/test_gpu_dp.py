"""MI355X: the data-parallel training step on the REAL engine.  Two processes share cuda:0 and exchange gradients through
a host-staged (gloo) process group -- the only multi-rank set-up a 1-GPU box offers; on a multi-GPU node the same code
runs over RCCL (backend "nccl").  What is checked is the engine's overlapped path: per-phase asynchronous all-reduce of
slices of the flat gradient while later backward phases are still being enqueued (engine.forward_backward_overlapped)."""
import os
import socket

import numpy as np
import pytest

from oracle import synth

pytestmark = pytest.mark.gpu

C, L, P, H, W, B_RANK, STEPS = 20, 2, 5, 64, 64, 4, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model(torch):
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn
    params = synth.make_fno_params(301, C, L, 12, 12, P, spectral_gain=4.0)
    m = Fno2d(2, 2, P, loss_name_to_fn("nmse"), L, 12, 12, C).cuda()
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()})
    return m


def _batch(torch, step, rank, world):
    b = synth.make_batch(311 + step, B_RANK * world, H, W, P, border_mask=True)
    return {k: torch.from_numpy(v[rank * B_RANK:(rank + 1) * B_RANK].copy()).cuda() for k, v in b.items()}


def _worker(rank, world, port, overlap, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.engine import FnoTrainEngine
        eng = FnoTrainEngine(_make_model(torch), lr=1e-3, loss_name="nmse", overlap=overlap)
        assert eng.sync.world == world
        # The two ranks of this test share ONE GPU and run their kernels concurrently (round 2 had to serialise them with a
        # cross-process lock: see tests/test_gpu_concurrency.py for what was wrong and how it is kept out of the library).
        grads = []
        for step in range(STEPS):
            b = _batch(torch, step, rank, world)
            eng.train_step(b["inputs"], b["label"], b["case_params"], b["mask"])
            torch.cuda.synchronize()
            grads.append(eng.flat.grad.cpu().numpy().copy())
        q.put((rank, eng.flat.data.cpu().numpy().copy(), grads))
    finally:
        dist.destroy_process_group()


def _run_world2(overlap):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("overlap", [True, False])
def test_engine_train_step_world2_equals_ddp_emulation(overlap):
    import torch
    from cfdbench_amd.engine import FnoTrainEngine
    assert torch.cuda.is_available()
    res = _run_world2(overlap)
    # single-process emulation of DistributedDataParallel on the concatenated batch: per-rank loss normaliser, the ranks'
    # gradients summed, 1/world folded into the optimiser -- with the very same kernels
    eng = FnoTrainEngine(_make_model(torch), lr=1e-3, loss_name="nmse")
    for step in range(STEPS):
        parts = []
        for rank in range(2):
            b = _batch(torch, step, rank, 2)
            eng.forward_backward(b["inputs"], b["label"], b["case_params"], b["mask"])
            parts.append(eng.flat.grad.clone())
        eng.flat.grad.copy_(parts[0] + parts[1])
        torch.cuda.synchronize()
        expect = eng.flat.grad.cpu().numpy()
        for rank in range(2):
            got = res[rank][2][step]
            if not np.array_equal(got, expect):
                bad = np.nonzero(got != expect)[0]
                offs = np.asarray(eng.flat.offsets)
                tensors = sorted({int(np.searchsorted(offs, int(i), side="right") - 1) for i in bad[:5000]})
                names = [n for n, _ in eng.model.named_parameters()]
                raise AssertionError(f"step {step}: rank {rank}'s reduced gradient differs in {bad.size} of {got.size} entries, max abs "
                                     f"{np.abs(got - expect).max():.3e}; parameter tensors (abi order) {tensors}; first {bad[:5]}; {names[:3]}")
        eng.optimizer_step(0.5)
    torch.cuda.synchronize()
    final = eng.flat.data.cpu().numpy()
    assert np.array_equal(res[0][1], res[1][1]), "replicas diverged"
    assert np.array_equal(res[0][1], final)
    assert not np.array_equal(parts[0].cpu().numpy(), parts[1].cpu().numpy())  # the ranks really had different shards


def _rccl_worker(port, overlap, graph, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CFDBENCH_DP_ALWAYS_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      NCCL_SOCKET_IFNAME="lo")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from cfdbench_amd.engine import FnoTrainEngine
        eng = FnoTrainEngine(_make_model(torch), lr=1e-3, loss_name="nmse", overlap=overlap, grad_buckets=3)
        assert eng.sync.exchange and eng.sync.device_native
        grads = []
        for step in range(STEPS):
            b = _batch(torch, step, 0, 1)
            (eng.train_step_graph if graph else eng.train_step)(b["inputs"], b["label"], b["case_params"], b["mask"])
            torch.cuda.synchronize()
            grads.append(eng.flat.grad.cpu().numpy().copy())
        q.put((eng.flat.data.cpu().numpy().copy(), grads))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap,graph", [(True, False), (False, False), (False, True), (True, True)])  # (True, True): the per-phase all-reduces captured in the graph (round 6)
def test_engine_train_step_over_rccl_one_rank_group(overlap, graph):
    """The RCCL branch of GradSync itself (backend "nccl": asynchronous all-reduce of slices of the flat gradient on the
    communicator's stream, ordered by events, while later backward phases are still being enqueued) on the one GPU this box
    has: a one-rank group whose collectives are forced on (CFDBENCH_DP_ALWAYS_EXCHANGE=1).  A SUM over one rank is the
    identity, so gradients and parameters must equal the plain single-process engine's bit for bit -- any missing
    stream dependency between the kernels and the collective shows up as a torn slice."""
    import torch
    import torch.multiprocessing as mp
    from cfdbench_amd.engine import FnoTrainEngine
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_rccl_worker, args=(_free_port(), overlap, graph, q))
    proc.start()
    try:
        params, grads = q.get(timeout=300)
    finally:
        proc.join(timeout=120)
        if proc.is_alive():
            proc.kill()
    assert proc.exitcode == 0
    eng = FnoTrainEngine(_make_model(torch), lr=1e-3, loss_name="nmse")
    eng.defer_flags = 0  # (a data-parallel engine keeps the step's five own launches: bitwise the same kernels as this one)
    for step in range(STEPS):
        b = _batch(torch, step, 0, 1)
        eng.train_step(b["inputs"], b["label"], b["case_params"], b["mask"])
        torch.cuda.synchronize()
        assert np.array_equal(grads[step], eng.flat.grad.cpu().numpy()), f"step {step}: gradient after the RCCL exchange differs"
    assert np.array_equal(params, eng.flat.data.cpu().numpy())


# ---- data parallelism for the graph-replayed models: U-Net (configs[2]) and Auto-DeepONet (configs[3]) --------
def _auto_model(torch, name):
    from cfdbench_amd.models.loss import loss_name_to_fn
    torch.manual_seed(17)
    if name == "unet":
        from cfdbench_amd.models.unet import UNet
        return UNet(2, 2, loss_name_to_fn("nmse"), n_case_params=3, insert_case_params_at="input", dim=4).cuda().train()
    from cfdbench_amd.models.auto_deeponet import AutoDeepONet
    return AutoDeepONet(16 * 16 + 3, 2, loss_name_to_fn("nmse"), num_label_samples=64, width=12, trunk_depth=3, branch_depth=3).cuda().train()


def _auto_batch(torch, step, rank):
    b = synth.make_smooth_batch(500 + 7 * step + rank, 4, 16, 16, 3)
    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in b.items()}


def _auto_optimizer(torch, m):
    from cfdbench_amd.optim import Adam
    return Adam(m.parameters(), lr=1e-3)


def _graph_dp_worker(rank, world, port, backend, name, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    if world == 1:
        os.environ["CFDBENCH_DP_ALWAYS_EXCHANGE"] = "1"
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        from cfdbench_amd.graph import GraphedTrainStep
        m = _auto_model(torch, name)
        opt = _auto_optimizer(torch, m)
        gs = GraphedTrainStep(m, opt, _auto_batch(torch, 0, rank), "nmse", restore_state=True)
        assert gs.dp and gs.exchange is not None
        # round 6: over RCCL the all-reduce is captured with the rest (ONE graph per step); host-staged backends keep the two graphs
        assert gs.one_graph == (backend == "nccl" and os.environ.get("CFDBENCH_DP_ONE_GRAPH", "1") != "0")
        assert (gs.graph_opt is None) == gs.one_graph
        losses = [float(gs(**_auto_batch(torch, s, rank))["nmse"].item()) for s in range(STEPS)]
        losses.append(float(gs.eager_step(_auto_batch(torch, STEPS, rank))["nmse"].item()))  # a step outside the graphs exchanges too
        losses.append(float(gs(**_auto_batch(torch, STEPS + 1, rank))["nmse"].item()))      # ... and the graphs still replay after it
        torch.cuda.synchronize()
        q.put((rank, [p.detach().cpu().numpy().copy() for p in m.parameters()], losses))
    finally:
        dist.destroy_process_group()


def _graph_dp_reference(torch, name, world):
    """One replica, the ranks' shards one after the other, gradients averaged by hand, the same one-launch Adam."""
    m = _auto_model(torch, name)
    opt = _auto_optimizer(torch, m)
    ps = [p for p in m.parameters()]
    for s in range(STEPS + 2):
        grads = []
        for r in range(world):
            opt.zero_grad(set_to_none=True)
            m(**_auto_batch(torch, s, r))["loss"]["nmse"].backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in ps])
        for i, p in enumerate(ps):
            if grads[0][i] is not None:
                p.grad = sum(g[i] for g in grads) / world
        opt.step()
    torch.cuda.synchronize()
    return [p.detach().cpu().numpy() for p in ps]


@pytest.mark.parametrize("name", ["unet", "auto_deeponet"])
def test_graphed_train_step_world2_equals_ddp_emulation(name):
    """Two processes on one GPU (gloo, host-staged exchange): forward + backward + gradient pack replayed from one graph, ONE all-reduce
    of the flat gradient, Adam replayed from a second graph reading the reduced gradients in place -- against the single-process
    emulation of DistributedDataParallel.  The captured Adam is the same kernel; BatchNorm statistics are per rank (DDP)."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_dp_worker, args=(r, 2, port, "gloo", name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = _graph_dp_reference(torch, name, 2)
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b), "replicas diverged"
    for got, ref in zip(res[0][1], want):
        assert np.allclose(got, ref, rtol=2e-4, atol=2e-6)  # (sum order of the two shards' gradients; ReLU nets)
    assert res[0][2] != res[1][2]


def test_graphed_train_step_over_rccl_one_rank_group():
    """The same two-graph step with the exchange on RCCL (backend "nccl", one-rank group with the collectives forced on): the
    all-reduce is enqueued between the two replays on the communicator's stream.  SUM over one rank = identity, so the parameters
    follow the single-process path."""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_graph_dp_worker, args=(0, 1, _free_port(), "nccl", "unet", q))
    proc.start()
    try:
        _, params, losses = q.get(timeout=300)
    finally:
        proc.join(timeout=120)
        if proc.is_alive():
            proc.kill()
    assert proc.exitcode == 0
    want = _graph_dp_reference(torch, "unet", 1)
    for got, ref in zip(params, want):
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-7)
    assert len(set(losses)) == len(losses)


def test_graphed_train_step_two_graphs_over_rccl_still_work(monkeypatch):
    """CFDBENCH_DP_ONE_GRAPH=0: the round-5 form (graph A, eager all-reduce, graph B) stays selectable."""
    import torch
    import torch.multiprocessing as mp
    monkeypatch.setenv("CFDBENCH_DP_ONE_GRAPH", "0")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_graph_dp_worker, args=(0, 1, _free_port(), "nccl", "auto_deeponet", q))
    proc.start()
    try:
        _, params, losses = q.get(timeout=300)
    finally:
        proc.join(timeout=120)
        if proc.is_alive():
            proc.kill()
    assert proc.exitcode == 0
    want = _graph_dp_reference(torch, "auto_deeponet", 1)
    for got, ref in zip(params, want):
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-7)


def _comm_probe_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from cfdbench_amd.harness.dist_util import comm_stream_overlaps, overlapping_comm_group
        first = comm_stream_overlaps(None)
        grp, ok = overlapping_comm_group(None)
        t = torch.arange(8, dtype=torch.float32, device="cuda")
        dist.all_reduce(t, group=grp)
        torch.cuda.synchronize()
        q.put((bool(first), bool(ok), t.cpu().tolist()))
    finally:
        dist.destroy_process_group()


def test_communicator_stream_is_probed_and_steered_onto_an_overlapping_queue():
    """harness/dist_util.overlapping_comm_group: HIP deals streams onto four hardware queues and the communicator's stream is drawn from
    the same pool -- one placement in four would serialise every gradient all-reduce behind the backward pass.  The probe times a small
    all-reduce beside 2 ms of kernels; a group whose stream does not overlap is replaced by a fresh one (next pool entry) until one does."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    proc = ctx.Process(target=_comm_probe_worker, args=(_free_port(), q))
    proc.start()
    try:
        first, ok, vals = q.get(timeout=300)
    finally:
        proc.join(timeout=120)
        if proc.is_alive():
            proc.kill()
    assert proc.exitcode == 0
    assert ok, f"no overlapping communicator stream found (the first group's verdict was {first})"
    assert vals == [float(i) for i in range(8)]


def test_bench_only_leg_as_a_data_parallel_job_over_a_one_rank_rccl_group():
    """`bench.py --only unet` in its data-parallel form (graph A -> all-reduce of the flat gradient over RCCL -> graph B), on the one GPU
    of this box as a one-rank group with the collectives forced on: the contract-style line of `--gpus N --only LEG`."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    env = dict(os.environ, CFDBENCH_DP_ALWAYS_EXCHANGE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_SOCKET_IFNAME="lo", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--only", "unet", "--steps", "5", "--warmup", "2"], env=env, cwd=repo,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["config"]["parallelism"] == "dp1"
    assert line["config"]["flat_gradient_bytes"] >= 4 * 1095362  # the U-Net's 1.1 M parameters (each tensor padded to 16 bytes)
    assert 0 < line["ms_per_step"] < 50 and line["value"] > 0 and np.isfinite(line["final_nmse"])
