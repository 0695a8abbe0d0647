"""CPU: host-side logic -- the built C-ABI library loads and exports every declared symbol (no compute calls),
flat-buffer layout, sharding, and the data-parallel gradient exchange on a 2-rank gloo group."""
import ctypes
import os
import re
import socket
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent


def test_header_and_binding_declare_the_same_symbols():
    from cfdbench_amd._capi import exported_symbols
    hdr = (REPO / "include" / "cfdbench_amd.h").read_text()
    declared = set(re.findall(r"\b(cfd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(exported_symbols())


def test_built_library_exports_every_symbol():
    from cfdbench_amd import build
    from cfdbench_amd._capi import CApi, exported_symbols
    lib = build.build()  # no-op when up to date; hipcc cross-compiles gfx950 without a GPU
    api = CApi(ctypes.CDLL(str(lib)), require_all=True)
    assert api.missing == []
    assert api.version() >= 100
    assert len(exported_symbols()) >= 30


def test_built_device_code_has_no_vulnerable_packed_fp32_form(tmp_path):
    """Round 3 root cause of the k_head_fwd corruption: v_pk_{fma,mul,add}_f32 with op_sel:[0,1,..] (low result from src0.lo and
    src1.HI) return a wrong low half in lanes 48-63 while certain other kernels are resident on the GPU.  The build refuses such
    instructions (cfdbench_amd/build.py:lint_object); here: every shipped device object is clean, and the lint does catch one."""
    import shutil
    import subprocess
    from cfdbench_amd import build
    build.build()
    objs = sorted((REPO / "cfdbench_amd" / "_C").glob("*.hip.o"))
    assert len(objs) >= 5
    for o in objs:
        assert build.lint_object(o) == [], o.name
    if shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists():
        pytest.skip("no hipcc: cannot build the positive control")
    src = tmp_path / "bad.hip"
    src.write_text(
        "#include <hip/hip_runtime.h>\n"
        "typedef float f2 __attribute__((ext_vector_type(2)));\n"
        "__global__ void k_bad(f2* p) { f2 a = p[threadIdx.x], b = p[threadIdx.x + 64], c = p[threadIdx.x + 128];\n"
        "  asm volatile(\"v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]\" : \"+v\"(c) : \"v\"(a), \"v\"(b));\n"
        "  asm volatile(\"v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\" : \"+v\"(c) : \"v\"(a), \"v\"(b));\n"
        "  p[threadIdx.x] = c; }\n")
    obj = tmp_path / "bad.hip.o"
    subprocess.run([build.hipcc(), *build.DEVICE_FLAGS, "-fPIC", "-x", "hip", "-c", str(src), "-o", str(obj)], check=True, capture_output=True)
    hits = build.lint_object(obj)
    assert len(hits) == 1 and "k_bad" in hits[0][0] and "op_sel:[0,1,0]" in hits[0][1], hits


def test_flat_layout_and_shard_range():
    import torch
    from cfdbench_amd.engine import flatten_layout, shard_range
    ps = [torch.zeros(3, 5), torch.zeros(2, 2, dtype=torch.complex64), torch.zeros(7)]
    offs, total = flatten_layout(ps)
    assert offs == [0, 16, 24] and total == 32
    for n, world in [(10, 3), (7, 8), (256, 8), (0, 2)]:
        spans = [shard_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.engine import GradSync, shard_range
        from oracle import fno_oracle as O
        from oracle import synth
        C, L, B = 4, 1, 4
        params = synth.make_fno_params(1, C, L, 12, 12, 5, dtype=np.float64, spectral_gain=4.0)
        batch = synth.make_batch(2, B, 64, 64, 5, dtype=np.float64)
        a, b = shard_range(B, rank, world)
        out = O.fno_forward(params, batch["inputs"][a:b], batch["case_params"][a:b], batch["mask"][a:b], batch["label"][a:b], L)
        gp = O.loss_grad_wrt_preds(out["cache"]["preds"], out["cache"]["label"], "nmse")
        g = O.fno_backward(params, out["cache"], gp, L)
        flat = np.concatenate([np.ascontiguousarray(g[k]).view(np.float64).reshape(-1) for k in params])
        t = torch.from_numpy(flat.copy())
        scale = GradSync(None, n_buckets=3).all_reduce(t)
        # the overlapped trainer's exchange: one asynchronous reduction per backward phase, in phase order
        from cfdbench_amd.engine import backward_phase_slices
        sizes = [np.ascontiguousarray(g[k]).view(np.float64).size for k in params]
        offs = [int(v) for v in np.concatenate([[0], np.cumsum(sizes)[:-1]])]
        slices = backward_phase_slices(offs, int(sum(sizes)), L)
        assert sorted(slices) == [(a0, b0) for a0, b0 in sorted(slices)] and sum(b0 - a0 for a0, b0 in slices) == sum(sizes)
        assert slices[0][1] == sum(sizes) and slices[-1][0] == 0 and len(slices) == L + 2
        t2 = torch.from_numpy(flat.copy())
        sync = GradSync(None)
        scale2 = sync.wait_all([sync.reduce_slice_async(t2, a0, b0) for a0, b0 in slices])
        assert scale2 == scale and torch.equal(t2, t)
        q.put((rank, (t * scale).numpy(), flat))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2():
    """N-rank exchange == single-process emulation of DDP semantics (per-rank loss normaliser, averaged grads)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    expected = (res[0][2] + res[1][2]) / 2.0
    for _, synced, _ in res:
        assert np.allclose(synced, expected, rtol=1e-12, atol=1e-15)
    assert not np.allclose(res[0][2], res[1][2])  # the ranks really had different shards


def _alone_worker(port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from cfdbench_amd.engine import GradSync
        quiet = GradSync(None, n_buckets=3)
        os.environ["CFDBENCH_DP_ALWAYS_EXCHANGE"] = "1"
        forced = GradSync(None, n_buckets=3)
        t = torch.arange(1000, dtype=torch.float32)
        handles = [forced.reduce_slice_async(t, a, b) for a, b in forced.bucket_slices(t.numel())]
        q.put((quiet.exchange, forced.exchange, quiet.reduce_slice_async(t, 0, 10) is None, sum(h is not None for h in handles),
               forced.wait_all(handles), bool(torch.equal(t, torch.arange(1000, dtype=torch.float32)))))
    finally:
        dist.destroy_process_group()


def test_one_rank_group_runs_the_collectives_only_when_asked():
    """CFDBENCH_DP_ALWAYS_EXCHANGE=1: a one-rank group still issues its all-reduces (how the RCCL branch is exercised on a one-GPU
    box, tests/test_gpu_dp.py); by default it skips them."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_alone_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert res == (False, True, True, 3, 1.0, True)


class _TinyAuto:  # (built inside the workers: a torch module with the AutoCfdModel call interface)
    @staticmethod
    def make(torch):
        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                torch.manual_seed(3)
                self.a = torch.nn.Linear(6, 5)
                self.b = torch.nn.Linear(5, 6)
                self.unused = torch.nn.Parameter(torch.zeros(4))  # never receives a gradient (the ResNet's bn1 / bn2)

            def forward(self, inputs, label=None):
                preds = self.b(torch.tanh(self.a(inputs)))
                d = preds - label
                return dict(preds=preds, loss=dict(nmse=(d * d).mean() / (label * label).mean(), mse=(d * d).mean()))
        return M()


def _graph_dp_batches(torch, rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [dict(inputs=torch.randn(8, 6, generator=g), label=torch.randn(8, 6, generator=g)) for _ in range(4)]


def _graph_dp_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.graph import GraphedTrainStep
        m = _TinyAuto.make(torch)
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
        batches = _graph_dp_batches(torch, rank)
        gs = GraphedTrainStep(m, opt, batches[0], "nmse", capture=False)  # host tensors: the three stages run eagerly
        assert gs.dp and gs.exchange is None
        losses = [float(gs(**b)["nmse"]) for b in batches]
        ex = gs.exchange
        assert len(ex.params) == 4 and m.unused.grad is None
        # the optimizer read the reduced gradients where they are: .grad is a view of the flat buffer
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(ex.params, ex.views))
        q.put((rank, [p.detach().numpy().copy() for p in ex.params], losses))
    finally:
        dist.destroy_process_group()


def test_graph_step_data_parallel_gloo_world2_equals_ddp_emulation():
    """GraphedTrainStep inside a two-rank group (pack every gradient pre-scaled into one flat buffer -> ONE
    all-reduce -> optimizer on views of that buffer) follows the single-process emulation of DistributedDataParallel: per-rank loss,
    averaged gradients, replicated Adam.  (`capture=False` runs the stages of the two graphs eagerly on host tensors; the captured
    form runs in tests/test_gpu_dp.py.)"""
    import torch
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # emulation: one replica, both shards, gradients averaged by hand
    m = _TinyAuto.make(torch)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    shards = [_graph_dp_batches(torch, r) for r in range(2)]
    ps = [m.a.weight, m.a.bias, m.b.weight, m.b.bias]
    for k in range(4):
        grads = []
        for r in range(2):
            opt.zero_grad(set_to_none=True)
            m(**shards[r][k])["loss"]["nmse"].backward()
            grads.append([p.grad.clone() for p in ps])
        for p, g0, g1 in zip(ps, *grads):
            p.grad = (g0 + g1) / 2
        opt.step()
    for r in range(2):
        for got, want in zip(res[r][1], ps):
            assert np.allclose(got, want.detach().numpy(), rtol=1e-5, atol=1e-7)
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))  # replicas stay identical bit for bit
    assert res[0][2] != res[1][2]  # (different shards)


def _sync_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.engine import sync_gradients
        g = torch.Generator().manual_seed(10 + rank)
        params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(2, 2, dtype=torch.cfloat)),
                  torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(7))]
        params[0].grad = torch.randn(3, 4, generator=g)
        params[1].grad = torch.complex(torch.randn(2, 2, generator=g), torch.randn(2, 2, generator=g))
        params[2].grad = torch.randn(5, generator=g)  # params[3] has no gradient (unused bn1/bn2-style parameter)
        before = [p.grad.clone() for p in params[:3]]
        sync_gradients(params)
        q.put((rank, [torch.view_as_real(b).numpy() if b.is_complex() else b.numpy() for b in before],
               [torch.view_as_real(p.grad).numpy() if p.grad.is_complex() else p.grad.numpy() for p in params[:3]],
               params[3].grad is None))
    finally:
        dist.destroy_process_group()


def test_autograd_path_gradient_averaging_gloo_world2():
    """sync_gradients (the DP exchange of the autograd training path, any model) == mean of the per-rank gradients."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    for i in range(3):
        mean = (res[0][1][i] + res[1][1][i]) / 2
        for r in res:
            assert np.allclose(r[2][i], mean, rtol=1e-6, atol=1e-7)
    assert res[0][3] and res[1][3]


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from cfdbench_amd import _lib
    from cfdbench_amd._capi import CfdError
    monkeypatch.setattr(_lib, "_LIB_PATH", tmp_path / "nope.so")
    monkeypatch.setattr(_lib, "_api", None)
    with pytest.raises(CfdError, match="no CPU/PyTorch fallback"):
        _lib.api()


def test_shard_indices_give_every_rank_the_same_number_of_steps():
    """with shards differing by one frame and drop_last, N = 31 / world 2 / batch 8 gave 2 steps on one rank and 1
    on the other -- the extra gradient all-reduce then pairs with the other rank's barrier."""
    from cfdbench_amd.harness.dist_util import shard_indices
    for n, world, bs in [(31, 2, 8), (47, 2, 8), (256, 8, 16), (1000, 8, 16), (10, 4, 8), (7, 2, 1)]:
        shards = [shard_indices(n, r, world, bs) for r in range(world)]
        sizes = {len(s) for s in shards}
        assert len(sizes) == 1, (n, world, bs, sizes)
        assert len({len(s) // bs for s in shards}) == 1
        flat = [i for s in shards for i in s]
        assert len(flat) == len(set(flat)) and all(0 <= i < n for i in flat)
        if n >= world * bs:
            assert len(flat) == (n // (world * bs)) * world * bs and len(shards[0]) % bs == 0


def _loop_worker(rank, world, port, n, bs, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.utils.data import DataLoader, Subset, TensorDataset
        from cfdbench_amd.harness.dist_util import average_buffers, broadcast_model_state, shard_indices
        data = TensorDataset(torch.arange(n, dtype=torch.float32))
        loader = DataLoader(Subset(data, shard_indices(n, rank, world, bs)), batch_size=bs, shuffle=True, drop_last=True)
        model = torch.nn.Sequential(torch.nn.Linear(1, 1), torch.nn.BatchNorm1d(1))
        with torch.no_grad():
            model[0].weight.fill_(float(rank + 1))
            model[1].running_mean.fill_(float(10 * rank))
        broadcast_model_state(model)
        w0, rm0 = float(model[0].weight), float(model[1].running_mean)
        steps = 0
        for _ in range(2):  # the epoch structure of harness/train_auto.py: one all-reduce per step, one barrier per epoch
            for (x,) in loader:
                t = x.sum().reshape(1)
                dist.all_reduce(t)
                steps += 1
            dist.barrier()
        with torch.no_grad():
            model[1].running_mean.fill_(float(rank))
        average_buffers(model)
        q.put((rank, steps, w0, rm0, float(model[1].running_mean), int(model[1].num_batches_tracked)))
    finally:
        dist.destroy_process_group()


def test_uneven_frame_count_runs_in_lockstep_gloo_world2():
    """N % (world * batch) != 0: both ranks run the same number of steps (no collective is left unpaired), replicas start
    from rank 0's weights and BatchNorm statistics, float buffers are averaged for the checkpoint."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, 31, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 2  # 31 frames -> 16 kept -> 8 per rank -> 1 step per epoch
    assert res[0][2] == res[1][2] == 1.0 and res[0][3] == res[1][3] == 0.0  # rank 0's state everywhere
    assert res[0][4] == res[1][4] == 0.5 and res[0][5] == res[1][5] == 0


class _ToyRollout:
    """Stand-in for an autoregressive model on the CPU (the product models are GPU-only): x_{t+1} = 0.9 x_t * mask + 0.01 p_0."""

    def modules(self):
        return []

    def generate_many(self, inputs, case_params, mask, steps):
        cur, out = inputs, []
        for _ in range(steps):
            cur = (0.9 * cur + 0.01 * case_params[:, :1, None, None]) * mask.unsqueeze(1)
            out.append(cur)
        return out


def _toy_cases(n_cases, steps):
    import torch
    g = torch.Generator().manual_seed(5)
    feats, cps = [], []
    for c in range(n_cases):
        f = torch.randn(steps + (c % 3), 3, 8, 9, generator=g)  # channels u, v, mask; cases of different lengths
        f[:, -1] = (torch.rand(8, 9, generator=g) > 0.2).float()
        feats.append(f[:max(steps, 1)] if f.shape[0] >= steps else f)
        cps.append(torch.randn(4, generator=g))
    return feats, cps


def _rollout_worker(rank, world, port, n_cases, steps, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.harness.test_multistep import infer, shard_cases
        feats, cps = _toy_cases(n_cases, steps)
        mine = shard_cases(n_cases, rank, world)
        got = infer(_ToyRollout(), [feats[i] for i in mine], [cps[i] for i in mine], steps, n_total_cases=n_cases)
        q.put((rank, mine, got))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cases", [5, 2, 1])
def test_rollout_metrics_shard_over_ranks_gloo_world2(n_cases):
    """SURVEY 8e, second half: test cases shard over the ranks with no data-path collective; one all-reduce of the (steps, 3)
    metric sums gives every rank the single-process numbers of src/test_multistep.py:135-177 (odd case count, and fewer cases
    than ranks: one rank holds nothing)."""
    import torch.multiprocessing as mp
    from cfdbench_amd.harness.test_multistep import get_metrics, infer
    steps = 6
    feats, cps = _toy_cases(n_cases, steps)
    single = infer(_ToyRollout(), feats, cps, steps)
    # the reference's own formulation: per case, per step, get_metrics on the masked u channel, then the mean over cases
    model = _ToyRollout()
    ref = [[] for _ in range(steps)]
    for f, cp in zip(feats, cps):
        preds = model.generate_many(f[0, :-1].unsqueeze(0), cp.unsqueeze(0), f[0, -1].unsqueeze(0), steps)
        for k in range(steps):
            ref[k].append(get_metrics(preds[k][0, 0] * f[k, -1], f[k, 0] * f[k, -1]))
    for k in range(steps):
        for name in ("mse", "nmse", "mae"):
            assert abs(single[k][name] - np.mean([m[name] for m in ref[k]])) <= 2e-6 * abs(single[k][name])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rollout_worker, args=(r, 2, port, n_cases, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0][1] + res[1][1]) == list(range(n_cases)) and not set(res[0][1]) & set(res[1][1])
    for _, _, got in res:
        for k in range(steps):
            for name in ("mse", "nmse", "mae"):
                assert abs(got[k][name] - single[k][name]) <= 1e-12 * abs(single[k][name]), (k, name)


class _ToyLoss:
    def get_score_names(self):
        return ["mse", "nmse"]

    def __call__(self, preds, labels):
        import torch
        mse = torch.mean((preds - labels) ** 2)
        return dict(mse=mse, nmse=mse / torch.mean(labels ** 2))


class _ToyAutoModel:
    """CPU stand-in for an AutoCfdModel (the product models run on the GPU only): preds = 0.9 u + 0.01 p_0, masked."""
    loss_fn = _ToyLoss()

    def eval(self):
        return self

    def __call__(self, inputs, label, case_params, mask):
        preds = (0.9 * inputs[:, :1] + 0.01 * case_params[:, :1, None, None]) * mask
        return dict(preds=preds, loss=self.loss_fn(preds=preds, labels=label[:, :1]))


def _eval_worker(rank, world, port, n_cases, bs, out_dir, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.harness.data import SyntheticAutoDataset
        from cfdbench_amd.harness.train_auto import evaluate
        data = SyntheticAutoDataset(n_cases=n_cases, n_frames=4, height=8, width=9, seed=3)
        got = evaluate(_ToyAutoModel(), data, Path(out_dir), batch_size=bs, plot_interval=0, sharded=True)
        q.put((rank, None if got is None else (got["preds"].numpy(), got["scores"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_cases,bs", [(3, 2), (1, 4), (4, 3)])
def test_evaluate_shards_batches_over_ranks_gloo_world2(tmp_path, n_cases, bs):
    """The periodic evaluation of a data-parallel run: batch k goes to rank k % world, rank 0 gets the single-process result back
    (same batches, same order: per-batch score lists, their means and the prediction tensor are identical) -- odd batch counts,
    a short last batch, and fewer batches than ranks."""
    import torch.multiprocessing as mp
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import evaluate
    data = SyntheticAutoDataset(n_cases=n_cases, n_frames=4, height=8, width=9, seed=3)
    (tmp_path / "single").mkdir()
    (tmp_path / "dp").mkdir()
    single = evaluate(_ToyAutoModel(), data, tmp_path / "single", batch_size=bs, plot_interval=0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, n_cases, bs, str(tmp_path / "dp"), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    preds, scores = res[0]
    assert np.array_equal(preds, single["preds"].numpy())
    assert scores == single["scores"]
    assert len(scores["all"]["nmse"]) == (len(data) + bs - 1) // bs


def test_shard_indices_reshuffle_per_epoch():
    """DistributedSampler.set_epoch semantics: the partition of the frames over the ranks changes with the epoch, stays a
    partition, keeps equal sizes, and is reproducible."""
    from cfdbench_amd.harness.dist_util import shard_indices
    n, world, bs = 100, 4, 8
    e0 = [shard_indices(n, r, world, bs, epoch=0) for r in range(world)]
    e1 = [shard_indices(n, r, world, bs, epoch=1) for r in range(world)]
    assert e0 == [shard_indices(n, r, world, bs) for r in range(world)]  # default epoch 0 = the round-2 behaviour
    assert e1 == [shard_indices(n, r, world, bs, epoch=1) for r in range(world)]
    assert any(set(a) != set(b) for a, b in zip(e0, e1))
    for shards in (e0, e1):
        flat = [i for s in shards for i in s]
        assert len(flat) == len(set(flat)) == 96 and len({len(s) for s in shards}) == 1


def test_committed_bench_line_keeps_the_driver_contract():
    """The newest profiles/r*_bench.json (a bench.py line measured on MI355X) carries every key of the bench contract, the
    roofline object of the dominant kernel and the CPU baseline -- a guard against editing a key away."""
    import json
    files = sorted((REPO / "profiles").glob("r*_bench.json"))
    assert files, "no committed bench line"
    line = json.loads(files[-1].read_text().strip().splitlines()[-1])
    for key, typ in dict(metric=str, value=(int, float), unit=str, n_gpus=int, steps=int, warmup=int, ms_per_step=(int, float),
                         higher_is_better=bool, scaling=str, dtype=str, data=str, config=dict).items():
        assert isinstance(line[key], typ), key
    assert "vs_baseline" in line and line["scaling"] == "weak" and line["higher_is_better"] is True
    assert "workload" in line["config"] and "model" not in line["config"]
    base = json.loads((REPO / "BASELINE.json").read_text())
    assert "train frames/sec" in base["metric"] and line["metric"].startswith("train frames/sec") and line["unit"] == "frames/s"
    rl = line["roofline"]
    assert rl["bound"] in ("hbm", "mfma") and rl["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3 and "traffic" in rl
    # round 6: the north-star figure and the step fraction as SCALAR members of `roofline` (the driver's record keeps scalars only)
    for key in ("spectral_conv2d_us", "spectral_conv2d_frac", "step_frac", "step_ms"):
        assert isinstance(rl[key], (int, float)), key
    assert isinstance(rl["kernel_symbol"], str) and abs(rl["spectral_conv2d_frac"] - line["roofline_spectral_conv2d"]["frac"]) < 1e-9
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert cb["steps"] >= 10 and cb["warmup"] >= 3 and isinstance(cb["cpu_model"], str) and cb["min_s"] <= cb["median_s"] <= cb["max_s"]
    assert line["value"] > 1000 * cb["value"]  # a sanity bound, not a target: the GPU path is the product


def _run_bench(*flags, env=None, timeout=600):
    import json
    import subprocess
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    r = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), *flags], env=e, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun and no rendezvous environment: bench.py spawns its two ranks, they meet over
    gloo, exchange the model's flat gradient buffer the way the engine does, and rank 0 alone prints the one JSON line."""
    r, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--dry-dp", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_dp"] is True
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 512
    assert d["config"]["backend"] == "gloo" and d["config"]["self_launched"] is True
    assert d["gradient_mean_ok"] is True and d["config"]["exchange_slices"] == 6


def test_bench_under_an_external_launcher_does_not_spawn():
    """With WORLD_SIZE in the environment (torch.distributed.run's contract) the process IS a rank: a one-rank dry run prints its
    line with dp1 and never re-launches itself."""
    r, lines = _run_bench("--gpus", "1", "--dry-dp", "--steps", "2", "--warmup", "0",
                          env=dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["config"]["parallelism"] == "dp1"
    assert lines[0]["config"]["self_launched"] is False


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies takes the job down with a non-zero exit code instead of leaving its peers in the rendezvous: without a GPU
    the measured (non --dry-dp) path exits 1 in every rank, and the launcher hands that code back."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPUs")
    r, lines = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert r.returncode == 1 and not lines
    assert "no CPU fallback" in r.stderr


def test_ffn_routes_its_layers_to_the_stack_and_chain_kernels():
    """Host logic of models/ffn.py: which runs of Linear(+activation) layers go to the fused stack kernel (all widths <= 128), which to
    the Linear-chain node (wider runs of >= 2 layers), which stay single calls -- the partition the DeepONet variants rely on."""
    import torch
    from cfdbench_amd.models.act_fn import get_act_fn
    from cfdbench_amd.models.ffn import Ffn

    def partition(dims, act_norm=False, act_on_output=False):
        f = Ffn(dims, act_fn=get_act_fn("relu", act_norm), act_on_output=act_on_output)
        mods, i, out = list(f.layers), 0, []
        while i < len(mods):
            run = f._fusable_run(mods, i)
            if run is not None:
                out.append(("stack", len(run[0])))
                i = run[4]
                continue
            chain = f._chain_run(mods, i)
            if chain is not None:
                out.append(("chain", len(chain[0])))
                i = chain[3]
                continue
            out.append(("single", 1))
            i += 2 if i + 1 < len(mods) and not isinstance(mods[i + 1], torch.nn.Linear) else 1
        return out

    assert partition([4295] + [100] * 8) == [("single", 1), ("stack", 7)]         # Auto-DeepONet branch: a wide first layer, then the stack
    assert partition([2] + [100] * 8) == [("stack", 8)]                            # its trunk
    assert partition([4101] + [200] * 8 + [1]) == [("chain", 9)]                   # Auto-FFN: wider than the stack kernel takes
    assert partition([512] * 3 + [1]) == [("chain", 3)]                            # the CNN variant's output FFN
    assert partition([300, 300, 100, 100, 100]) == [("chain", 2), ("stack", 2)]    # the narrow tail of a wide run belongs to the stack kernel
    assert partition([100, 100, 100], act_norm=True) == [("single", 1), ("single", 1)]  # NormAct: statistics over the sample, no fusion
    assert partition([64, 1]) == [("single", 1)]


def test_resnet_dropout_step_counter_is_training_state_not_a_checkpoint_key():
    """The ResNet's dropout step counter (a non-persistent buffer since round 4: it follows .to(), GraphedTrainStep restores it, the
    kernels read it on the device) stays out of the state_dict -- the reference's checkpoint keys -- and round-trips through the
    harness's extra training state."""
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.resnet import ResNet
    m = ResNet(2, 2, 3, loss_name_to_fn("nmse"), hidden_chan=4, num_blocks=1, kernel_size=7, padding=3)
    assert not any("drop_step" in k for k in m.state_dict())
    assert any(b is m._drop_step for b in m.buffers()) and not getattr(m, "graph_unsafe", False)
    assert m.extra_train_state() == dict(train_steps=0)
    m.load_extra_train_state(dict(train_steps=41))
    assert m.extra_train_state() == dict(train_steps=41) and int(m._drop_step) == 41
