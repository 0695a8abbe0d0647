"""MI355X: the harness end to end on a synthetic dataset (train both paths, checkpoint tree, test, multi-step inference)
and the HIP-graph rollout against Fno2d.generate_many and the golden rollouts of the reference."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import fno_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _args(tmp_path, **kw):
    from cfdbench_amd.harness.args import Args
    return Args(model="fno", data_name="cavity_bc", loss_name="nmse", fno_hidden_dim=8, fno_depth=2, lr=5e-3,
                output_dir=str(tmp_path), num_epochs=4, batch_size=4, eval_batch_size=4, eval_interval=2, log_interval=5,
                plot_interval=0, **kw)


@pytest.mark.parametrize("fused", [0, 1])
def test_train_eval_test_artifacts(torch, tmp_path, fused):
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.common import get_best_ckpt, get_output_dir, load_best_ckpt, load_json
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import test, train
    args = _args(tmp_path, fused=fused)
    out = get_output_dir(args, is_auto=True)
    tr = SyntheticAutoDataset(n_cases=6, n_frames=6, height=64, width=64, seed=0)
    dev = SyntheticAutoDataset(n_cases=2, n_frames=4, height=64, width=64, seed=1)
    torch.manual_seed(0)
    model = init_model(args).cuda()
    losses = train(model, tr, dev, out, num_epochs=args.num_epochs, lr=args.lr, lr_step_size=args.lr_step_size,
                   lr_gamma=args.lr_gamma, batch_size=args.batch_size, eval_batch_size=args.eval_batch_size,
                   log_interval=args.log_interval, eval_interval=args.eval_interval, fused=bool(fused), plot_interval=0)
    assert len(losses) == 4 * ((len(tr) + 3) // 4)
    assert np.mean(losses[-6:]) < 0.9 * np.mean(losses[:6]), "training does not reduce the nMSE"
    for ep in (1, 3):
        d = out / f"ckpt-{ep}"
        assert (d / "model.pt").exists() and (d / "dev_scores.json").exists() and (d / "train_loss.json").exists()
        sc = load_json(d / "scores.json")
        assert set(sc) == {"ep", "train_loss", "dev_loss", "time"}
        ds = load_json(d / "dev_scores.json")
        assert set(ds) == {"mean", "all"} and set(ds["all"]) == {"mse", "rmse", "mae", "nmse"}
        assert "input_nmse" in ds["mean"]
    assert (out / "train_losses.json").exists()
    # checkpoint = bare state_dict with the reference's keys / dtypes
    sd = torch.load(get_best_ckpt(out) / "model.pt", map_location="cpu")
    assert sd["blocks.0.conv0.weights1"].dtype == torch.complex64 and "fc2.bias" in sd
    m2 = init_model(args).cuda()
    load_best_ckpt(m2, out)
    res = test(m2, dev, out / "test", batch_size=1, plot_interval=10)
    assert (out / "test" / "preds.pt").exists() and (out / "test" / "scores.json").exists()
    assert res["preds"].shape == (2 * len(dev), 1, 64, 64)


@pytest.mark.parametrize("fused", [0, 1])
def test_resume_reproduces_the_uninterrupted_run(torch, tmp_path, fused):
    """4 epochs in one go == 2 epochs, process 'restart', resume to 4 (weights, loss history, bitwise)."""
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import train
    tr = SyntheticAutoDataset(n_cases=4, n_frames=5, height=64, width=64, seed=0)
    dev = SyntheticAutoDataset(n_cases=1, n_frames=4, height=64, width=64, seed=1)

    def run(out, num_epochs, resume):
        args = _args(out, fused=fused)
        torch.manual_seed(0)
        model = init_model(args).cuda()
        if resume:  # a restarted process knows nothing of the first run: different init, different generator state
            torch.manual_seed(123)
            for p_ in model.parameters():
                p_.data.mul_(0.5)
        losses = train(model, tr, dev, out, num_epochs=num_epochs, lr=args.lr, lr_step_size=1, lr_gamma=0.8,
                       batch_size=4, eval_batch_size=4, log_interval=100, eval_interval=2, fused=bool(fused), plot_interval=0,
                       resume=resume)
        return model, losses

    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    ma, la = run(tmp_path / "a", 4, False)
    run(tmp_path / "b", 2, False)
    assert (tmp_path / "b" / "train_state.pt").exists()
    mb, lb = run(tmp_path / "b", 4, True)
    assert la == lb
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(pa, pb), k


@pytest.mark.parametrize("model_name,kw", [("unet", dict(unet_dim=4)), ("auto_deeponet", dict(deeponet_width=24, branch_depth=3, trunk_depth=3)),
                                           ("fno", dict())])
def test_train_auto_graph_option_follows_the_eager_run(torch, tmp_path, model_name, kw):
    """``--graph 1``: the autograd step captured once as a HIP graph and replayed (the warm-up steps before the capture are rolled
    back; the short last batch of an epoch runs eagerly on the same capturable Adam).  Same kernels, same data order: the loss
    curve follows the eager run to the rounding of the optimiser's fp32 step arithmetic, and the artefacts are the same."""
    from cfdbench_amd.harness.args import Args, is_args_valid
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.common import get_output_dir
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import train
    tr = SyntheticAutoDataset(n_cases=7, n_frames=6, height=64, width=64, seed=0)   # 35 frames: batches of 4 + a short one of 3
    dev = SyntheticAutoDataset(n_cases=2, n_frames=4, height=64, width=64, seed=1)
    curves = {}
    for graph in (0, 1):
        args = Args(model=model_name, data_name="cavity_bc", loss_name="nmse", fno_hidden_dim=8, fno_depth=2, lr=2e-3,
                    output_dir=str(tmp_path / f"g{graph}"), num_epochs=3, batch_size=4, eval_batch_size=4, eval_interval=3,
                    log_interval=4, plot_interval=0, lr_step_size=1, graph=graph, **kw)
        is_args_valid(args)
        out = get_output_dir(args, is_auto=True)
        torch.manual_seed(0)
        model = init_model(args).cuda()
        curves[graph] = np.array(train(model, tr, dev, out, num_epochs=3, lr=args.lr, lr_step_size=1, lr_gamma=0.5, batch_size=4,
                                       eval_batch_size=4, log_interval=4, eval_interval=3, plot_interval=0, graph=bool(graph)))
        assert (out / "ckpt-2" / "model.pt").exists() and (out / "train_state.pt").exists()
    e, g = curves[0], curves[1]
    assert len(e) == len(g) == 3 * 9
    print(model_name, "eager", np.round(e[[0, 8, 17, 26]], 5), "graph", np.round(g[[0, 8, 17, 26]], 5), "max rel", np.max(np.abs(e - g) / e))
    assert abs(e[0] - g[0]) <= 1e-6 * e[0]            # the first step sees the untouched initial weights (warm-up rolled back)
    assert np.max(np.abs(e - g) / e) < 2e-2           # same trajectory (the step arithmetic of Adam differs in rounding)


def test_resume_with_graph_option_reproduces_the_uninterrupted_run(torch, tmp_path):
    """--graph 1 + --resume 1 (U-Net): the capture's warm-up steps are rolled back over the LOADED optimiser state and BatchNorm
    buffers, so 4 epochs in one go == 2 epochs, restart, resume to 4 -- bitwise (the replayed graph is deterministic)."""
    from cfdbench_amd.harness.args import Args
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import train
    tr = SyntheticAutoDataset(n_cases=5, n_frames=5, height=64, width=64, seed=0)  # 20 frames: 5 full batches of 4
    dev = SyntheticAutoDataset(n_cases=1, n_frames=4, height=64, width=64, seed=1)

    def run(out, num_epochs, resume):
        args = Args(model="unet", data_name="cavity_bc", loss_name="nmse", unet_dim=4, lr=2e-3, output_dir=str(out), graph=1)
        torch.manual_seed(0)
        model = init_model(args).cuda()
        if resume:
            torch.manual_seed(123)
            for p_ in model.parameters():
                p_.data.mul_(0.5)
        losses = train(model, tr, dev, out, num_epochs=num_epochs, lr=args.lr, lr_step_size=1, lr_gamma=0.8, batch_size=4,
                       eval_batch_size=4, log_interval=100, eval_interval=2, plot_interval=0, resume=resume, graph=True)
        return model, losses

    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    ma, la = run(tmp_path / "a", 4, False)
    run(tmp_path / "b", 2, False)
    mb, lb = run(tmp_path / "b", 4, True)
    assert la == lb
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(pa, pb), k


@pytest.mark.parametrize("model_name", ["deeponet", "ffn"])
def test_nonauto_train_graph_option(torch, tmp_path, model_name):
    """train.py --graph 1 (non-autoregressive DeepONet / FFN): the captured step draws fresh query points on every replay (torch's
    CUDA generator is graph-safe), the loss falls and the artefacts are the eager loop's.  (No curve comparison: the warm-up
    before the capture consumes random numbers, so the two runs see different query points.)"""
    from cfdbench_amd.harness.args import Args
    from cfdbench_amd.harness.common import get_output_dir, load_json
    from cfdbench_amd.harness.train import SyntheticDataset, init_model, train
    args = Args(model=model_name, data_name="cavity_prop_bc_geo", loss_name="nmse", deeponet_width=32, branch_depth=3,
                trunk_depth=3, ffn_width=32, ffn_depth=3, lr=2e-3, output_dir=str(tmp_path), num_epochs=6, batch_size=4,
                eval_interval=3, log_interval=5, plot_interval=0, graph=1)
    out = get_output_dir(args)
    tr, dev = SyntheticDataset(4, 6, 16, 16, seed=0), SyntheticDataset(2, 3, 16, 16, seed=1)
    torch.manual_seed(0)
    model = init_model(args).cuda()
    model.num_label_samples = 200
    losses = train(model, tr, dev, out, num_epochs=args.num_epochs, lr=args.lr, batch_size=args.batch_size,
                   eval_interval=args.eval_interval, log_interval=args.log_interval, plot_interval=0, graph=True)
    assert len(losses) == 6 * 6 and np.all(np.isfinite(losses))
    assert len(set(np.round(losses[:6], 9))) == 6, "every replay must see freshly drawn query points"
    assert np.mean(losses[-6:]) < np.mean(losses[:6]), "training does not reduce the nMSE"
    assert (out / "ckpt-5" / "model.pt").exists() and set(load_json(out / "ckpt-5" / "scores.json")) == {"ep", "train_loss", "dev_loss", "time"}


def test_graph_option_is_refused_where_it_cannot_hold(torch, tmp_path):
    from cfdbench_amd.harness.args import Args, is_args_valid
    for bad in (dict(model="fno", fused=1), dict(model="unet", gradient_accumulation_steps=2)):
        with pytest.raises(AssertionError):
            is_args_valid(Args(data_name="cavity_bc", graph=1, **bad))
    # (round 5) the ResNet's dropout stream counter lives on the device since round 4: its step is capturable and no longer refused
    is_args_valid(Args(data_name="cavity_bc", graph=1, model="resnet"))


def test_train_with_device_batch_loader(torch, tmp_path):
    """the fused trainer fed by on-device batch assembly (no DataLoader, no per-item host work)."""
    import time
    from torch.utils.data import DataLoader
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import DeviceBatchLoader, SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import collate_fn, train
    args = _args(tmp_path, fused=1)
    tr = SyntheticAutoDataset(n_cases=6, n_frames=6, height=64, width=64, seed=0)
    dev = SyntheticAutoDataset(n_cases=2, n_frames=4, height=64, width=64, seed=1)
    torch.manual_seed(0)
    model = init_model(args).cuda()
    losses = train(model, tr, dev, tmp_path / "run", num_epochs=4, lr=args.lr, batch_size=4, eval_batch_size=4,
                   log_interval=100, eval_interval=2, fused=True, plot_interval=0, device_loader=True)
    assert len(losses) == 4 * ((len(tr) + 3) // 4)
    assert np.mean(losses[-6:]) < 0.9 * np.mean(losses[:6])
    # same batches as the reference's collate when the order is the same
    ref = list(DataLoader(tr, batch_size=7, shuffle=False, collate_fn=collate_fn))
    got = list(DeviceBatchLoader(tr, 7, shuffle=False))
    for a, b in zip(ref, got):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # loader-only rate (printed with -s): item-by-item collate vs on-device gathers
    big = SyntheticAutoDataset(n_cases=16, n_frames=65, height=64, width=64, seed=2)  # 1024 frames
    rates = {}
    for name, mk in (("DataLoader+collate_fn", lambda: DataLoader(big, batch_size=256, shuffle=True, collate_fn=collate_fn)),
                     ("DeviceBatchLoader", lambda: DeviceBatchLoader(big, 256, shuffle=True))):
        ld = mk()
        for b in ld:  # warm-up epoch
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(5):
            for b in ld:
                n += b["inputs"].shape[0]
        torch.cuda.synchronize()
        rates[name] = n / (time.perf_counter() - t0)
        print(f"{name}: {rates[name]:.0f} frames/s")
    assert rates["DeviceBatchLoader"] > 5 * rates["DataLoader+collate_fn"]


def test_fused_and_autograd_paths_agree(torch, tmp_path):
    """Same data order, same init: FnoTrainEngine's steps == autograd + torch.optim.Adam steps."""
    from cfdbench_amd.engine import FnoTrainEngine
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import collate_fn
    args = _args(tmp_path)
    ds = SyntheticAutoDataset(n_cases=2, n_frames=5, height=64, width=64, seed=3)
    batches = [collate_fn([ds[i] for i in range(k, k + 4)]) for k in (0, 4)]
    torch.manual_seed(0)
    ma = init_model(args).cuda()
    mb = init_model(args).cuda()
    mb.load_state_dict(ma.state_dict())
    opt = torch.optim.Adam(ma.parameters(), lr=1e-3)
    eng = FnoTrainEngine(mb, lr=1e-3, loss_name="nmse")
    for b in batches * 2:
        ma(**b)["loss"]["nmse"].backward()
        opt.step()
        opt.zero_grad()
        eng.train_step(b["inputs"], b["label"], b["case_params"], b["mask"])
    for (k, pa), (_, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        a = torch.view_as_real(pa) if pa.is_complex() else pa
        bb = torch.view_as_real(pb) if pb.is_complex() else pb
        assert O.rel_nmse(bb.cpu().numpy(), a.cpu().numpy()) < 1e-9, k


@pytest.mark.parametrize("fused", [0, 1])
def test_training_options_schedules_and_early_stopping(torch, tmp_path, fused):
    """SURVEY.md 8f-4 options: cosine / plateau learning-rate schedules on both training paths (the fused engine follows the
    torch scheduler bit for bit through a shadow optimizer), early stopping on the validation loss, and (autograd path)
    gradient accumulation == one step on the concatenated micro-batches' averaged gradients."""
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.common import get_output_dir
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import train
    tr = SyntheticAutoDataset(n_cases=4, n_frames=5, height=64, width=64, seed=0)
    dev = SyntheticAutoDataset(n_cases=2, n_frames=3, height=64, width=64, seed=1)
    finals = {}
    for kind in ("cosine", "plateau"):
        args = _args(tmp_path / kind, fused=fused)
        out = get_output_dir(args, is_auto=True)
        torch.manual_seed(0)
        model = init_model(args).cuda()
        losses = train(model, tr, dev, out, num_epochs=6, lr=5e-3, batch_size=4, eval_batch_size=4, eval_interval=1,
                       fused=bool(fused), plot_interval=0, lr_scheduler_kind=kind, lr_scheduler_factor=0.5, lr_scheduler_patience=0,
                       early_stopping_patience=2, early_stopping_delta=10.0)  # delta so large that nothing counts as improvement
        # the first evaluation sets the best value, the next two are "no improvement" -> stop after epoch 2
        assert len(losses) == 3 * (len(tr) // 4), (kind, len(losses))
        state = torch.load(out / "train_state.pt", map_location="cpu", weights_only=False)
        assert state["scheduler"]["kind"] == kind and state["early_stopping"]["bad"] == 2
        if kind == "plateau":  # patience 0: the rate halves at every evaluation without improvement
            assert abs(state["scheduler"]["lr"] - 5e-3 * 0.25) < 1e-12
        else:
            assert abs(state["scheduler"]["lr"] - 5e-3 * 0.5 * (1 + np.cos(np.pi * 3 / 6))) < 1e-9
        finals[kind] = losses
    if not fused:
        from torch.utils.data import DataLoader
        from cfdbench_amd.harness.train_auto import collate_fn
        args = _args(tmp_path / "accum", fused=0)
        n_micro = len(tr) // 4
        assert n_micro >= 2
        torch.manual_seed(0)
        m1 = init_model(args).cuda()
        train(m1, tr, dev, get_output_dir(args, is_auto=True), num_epochs=1, lr=1e-3, batch_size=4, eval_interval=10, plot_interval=0,
              gradient_accumulation_steps=n_micro)
        # by hand: ONE Adam step on the mean of the micro-batch gradients (same seed -> same initial weights, same shuffle)
        torch.manual_seed(0)
        m3 = init_model(args).cuda()
        opt = torch.optim.Adam(m3.parameters(), lr=1e-3)
        m3.train()
        for batch in DataLoader(tr, batch_size=4, shuffle=True, collate_fn=collate_fn):
            (m3(**batch)["loss"]["nmse"] / n_micro).backward()
        opt.step()
        for (k, a), b in zip(m1.state_dict().items(), m3.state_dict().values()):
            assert torch.equal(a, b), k


def test_multistep_inference_metrics(torch, tmp_path):
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.test_multistep import get_metrics, infer, infer_case, prepare_cases
    args = _args(tmp_path)
    torch.manual_seed(1)
    model = init_model(args).cuda()
    data = SyntheticAutoDataset(n_cases=3, n_frames=4, height=64, width=64, seed=5, border_mask=True)
    steps = 6  # > n_frames: cases are padded with their last frame
    feats, cps = prepare_cases(data, steps)
    assert all(f.shape[0] == steps for f in feats)
    metrics = infer(model, feats, cps, steps)
    assert len(metrics) == steps and set(metrics[0]) == {"mse", "nmse", "mae"}
    # the reference's per-case / per-step formulation (test_multistep.py:144-176)
    ref = []
    preds = [infer_case(model, f, c, steps) for f, c in zip(feats, cps)]
    for s in range(steps):
        ms = []
        for c in range(3):
            msk = feats[c][s][-1]
            ms.append(get_metrics(preds[c][s][0][0] * msk, feats[c][s][0] * msk))
        ref.append({k: float(np.mean([m[k] for m in ms])) for k in ms[0]})
    for a, b in zip(metrics, ref):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-12, (k, a[k], b[k])


def _multistep_rank(rank, world, port, out_dir, steps, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks on the box's one GPU: host-side collective
    try:
        from cfdbench_amd.harness.autoregressive import init_model
        from cfdbench_amd.harness.data import SyntheticAutoDataset
        from cfdbench_amd.harness.test_multistep import infer, prepare_cases, shard_cases
        args = _args(Path(out_dir))
        torch.manual_seed(1)
        model = init_model(args).cuda()
        data = SyntheticAutoDataset(n_cases=5, n_frames=4, height=64, width=64, seed=5, border_mask=True)
        mine = shard_cases(5, rank, world)
        feats, cps = prepare_cases(data, steps, only=set(mine))
        assert len(feats) == len(mine)
        q.put((rank, mine, infer(model, feats, cps, steps, n_total_cases=5)))
    finally:
        dist.destroy_process_group()

def _evaluate_rank(rank, world, port, out_dir, device_loader, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cfdbench_amd.harness.autoregressive import init_model
        from cfdbench_amd.harness.data import SyntheticAutoDataset
        from cfdbench_amd.harness.train_auto import evaluate
        torch.manual_seed(1)
        model = init_model(_args(Path(out_dir))).cuda()
        data = SyntheticAutoDataset(n_cases=3, n_frames=4, height=64, width=64, seed=5, border_mask=True)
        got = evaluate(model, data, Path(out_dir), batch_size=2, plot_interval=0, device_loader=bool(device_loader), sharded=True)
        q.put((rank, None if got is None else (got["preds"].numpy(), got["scores"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("device_loader", [0, 1])
def test_evaluate_sharded_over_two_ranks(torch, tmp_path, device_loader):
    """The periodic evaluation of a data-parallel run (train_auto.py:61-148 is single-process): batch k on rank k % world, rank 0
    assembles the single-process result -- bitwise, because every batch is the same launch on the same frames (5 batches of 2, 2, 2,
    2, 1 frames over two ranks)."""
    import socket
    import torch.multiprocessing as mp
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.train_auto import evaluate
    torch.manual_seed(1)
    model = init_model(_args(tmp_path)).cuda()
    data = SyntheticAutoDataset(n_cases=3, n_frames=4, height=64, width=64, seed=5, border_mask=True)
    single = evaluate(model, data, tmp_path, batch_size=2, plot_interval=0, device_loader=bool(device_loader))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_evaluate_rank, args=(r, 2, port, str(tmp_path), device_loader, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[1] is None
    preds, scores = res[0]
    assert len(scores["all"]["nmse"]) == 5
    assert np.array_equal(preds, single["preds"].numpy())
    assert scores == single["scores"]



def test_multistep_inference_sharded_over_two_ranks(torch, tmp_path):
    """rollout inference shards the test cases over the ranks (r, r + world, ...), every rank rolls its cases out from one
    HIP graph, and ONE all-reduce of the (steps, 3) metric sums gives the single-process metrics (src/test_multistep.py:135-177)."""
    import socket
    import torch.multiprocessing as mp
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.test_multistep import infer, prepare_cases
    steps = 6
    args = _args(tmp_path)
    torch.manual_seed(1)
    model = init_model(args).cuda()
    data = SyntheticAutoDataset(n_cases=5, n_frames=4, height=64, width=64, seed=5, border_mask=True)
    feats, cps = prepare_cases(data, steps)
    single = infer(model, feats, cps, steps)
    with torch.no_grad():  # the graph rollout behind infer() is bitwise the step-by-step generate_many
        start = torch.stack([f[0, :-1] for f in feats])
        plain = model.generate_many(inputs=start, case_params=torch.stack(cps), mask=torch.stack([f[0, -1] for f in feats]), steps=steps)
    from cfdbench_amd.harness.test_multistep import rollout_frames
    graph = rollout_frames(model, start, torch.stack(cps), torch.stack([f[0, -1] for f in feats]), steps)
    assert all(torch.equal(a, b) for a, b in zip(plain, graph))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_multistep_rank, args=(r, 2, port, str(tmp_path), steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for _, _, got in res:
        for a, b in zip(got, single):
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-6 * abs(b[k]) + 1e-12, (k, a[k], b[k])  # fp32 sums on the device; batch 3 + 2 against batch 5


def test_multistep_unet_in_train_mode_follows_the_reference_per_case(torch, tmp_path):
    """the reference never calls model.eval() in test_multistep.py and rolls out ONE case at a time, so a U-Net's
    BatchNorm normalises with that case's own statistics.  ``infer`` must not pool the statistics over the cases: with the
    model in training mode its metrics equal the per-case formulation, and differ from a batched train-mode rollout; in
    eval mode (no batch dependence left) the batched rollout is used and equals the per-case one as well."""
    from cfdbench_amd.harness.data import SyntheticAutoDataset
    from cfdbench_amd.harness.test_multistep import batch_dependent, get_metrics, infer, infer_case, prepare_cases
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.unet import UNet
    torch.manual_seed(3)
    model = UNet(2, 2, loss_name_to_fn("nmse"), 5, insert_case_params_at="input", bilinear=False, dim=2).cuda()
    data = SyntheticAutoDataset(n_cases=3, n_frames=4, height=32, width=32, seed=6, border_mask=True)
    steps = 3
    feats, cps = prepare_cases(data, steps)

    def per_case():
        preds = [infer_case(model, f, c, steps) for f, c in zip(feats, cps)]
        out = []
        for s in range(steps):
            ms = [get_metrics(preds[c][s][0][0] * feats[c][s][-1], feats[c][s][0] * feats[c][s][-1]) for c in range(3)]
            out.append({k: float(np.mean([m[k] for m in ms])) for k in ms[0]})
        return out

    for mode in ("train", "eval"):
        getattr(model, mode)()
        state = {k: v.clone() for k, v in model.state_dict().items()}
        assert batch_dependent(model) == (mode == "train")
        got = infer(model, feats, cps, steps)
        model.load_state_dict(state)  # train mode updates the running statistics: same start for the comparison run
        ref = per_case()
        model.load_state_dict(state)
        for a, b in zip(got, ref):
            for k in a:
                assert abs(a[k] - b[k]) <= 1e-5 * abs(b[k]) + 1e-12, (mode, k, a[k], b[k])
    model.train()
    with torch.no_grad():  # what r1 did: all cases as one train-mode batch -> pooled BatchNorm statistics
        start = torch.stack([f[0, :-1] for f in feats])
        pooled = model.generate_many(inputs=start, case_params=torch.stack(list(cps)), mask=torch.stack([f[0, -1] for f in feats]),
                                     steps=steps)
        single = infer_case(model, feats[0], cps[0], steps)
    assert not torch.allclose(pooled[-1][:1], single[-1], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", ["rollout_small_64x64", "rollout_small_66x65"])
def test_graph_rollout_matches_generate_many_and_reference_golden(torch, golden_dir, name):
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.rollout import FnoRollout
    from oracle import synth
    g = np.load(golden_dir / f"{name}.npz")
    pseed, bseed, B, C, L, H, W, p, steps, border = [int(v) for v in g["meta"]]
    params = synth.make_fno_params(pseed, C, L, 12, 12, p, spectral_gain=float(g["gain"]))
    batch = synth.make_smooth_batch(bseed, B, H, W, p)
    if border:
        batch["mask"][:, :, 0, :] = 0
        batch["mask"][:, :, -1, :] = 0
        batch["mask"][:, :, :, 0] = 0
    model = Fno2d(2, 2, p, loss_name_to_fn("nmse"), L, 12, 12, C).cuda()
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()})
    x, cp, mask = (torch.from_numpy(batch[k]).cuda() for k in ("inputs", "case_params", "mask"))
    with torch.no_grad():
        plain = model.generate_many(x, cp, mask, steps)
    ro = FnoRollout(model)
    fast = ro.generate_many(x, cp, mask, steps)
    again = ro.generate_many(x, cp, mask, steps)  # replay from the cached graph
    for t in range(steps):
        assert torch.equal(plain[t], fast[t]) and torch.equal(fast[t], again[t])
    # vs the reference's own rollout (oracle/make_golden.py: gen_rollout), north-star tolerance and far tighter
    assert O.rel_nmse(fast[0].cpu().numpy(), g["first"]) < 1e-9
    assert O.rel_nmse(fast[-1].cpu().numpy(), g["last"]) < 1e-7
    # unbatched entry point
    one = ro.generate_many(x[0], cp[0], mask[0, 0], 2)
    assert tuple(one[0].shape) == (1, 2, H, W) and O.rel_nmse(one[1].cpu().numpy(), fast[1][:1].cpu().numpy()) < 1e-10


@pytest.mark.parametrize("model_name", ["deeponet", "ffn"])
def test_nonauto_train_eval_test_artifacts(torch, tmp_path, model_name):
    """src/train.py's loop on the DeepONet / FfnModel drop-ins: loss decreases, reference file tree, reloadable ckpt."""
    from cfdbench_amd.harness.args import Args
    from cfdbench_amd.harness.common import get_best_ckpt, get_output_dir, load_best_ckpt, load_json
    from cfdbench_amd.harness.train import SyntheticDataset, init_model, test, train
    args = Args(model=model_name, data_name="cavity_prop_bc_geo", loss_name="nmse", deeponet_width=32, branch_depth=3,
                trunk_depth=3, ffn_width=32, ffn_depth=3, lr=2e-3, output_dir=str(tmp_path), num_epochs=6, batch_size=4,
                eval_interval=3, log_interval=5, plot_interval=0)
    out = get_output_dir(args)
    tr, dev = SyntheticDataset(4, 6, 16, 16, seed=0), SyntheticDataset(2, 3, 16, 16, seed=1)
    torch.manual_seed(0)
    model = init_model(args).cuda()
    model.num_label_samples = 200
    losses = train(model, tr, dev, out, num_epochs=args.num_epochs, lr=args.lr, batch_size=args.batch_size,
                   eval_interval=args.eval_interval, log_interval=args.log_interval, plot_interval=0)
    assert len(losses) == 6 * 6 and np.all(np.isfinite(losses))
    assert np.mean(losses[-6:]) < np.mean(losses[:6]), "training does not reduce the nMSE"
    for ep in (2, 5):
        d = out / f"ckpt-{ep}"
        assert (d / "model.pt").exists() and (d / "train_loss.json").exists()
        assert set(load_json(d / "scores.json")) == {"ep", "train_loss", "dev_loss", "time"}
        assert set(load_json(d / "dev_loss.json")["mean"]) == {"mse", "rmse", "mae", "nmse"}
    m2 = init_model(args).cuda()
    load_best_ckpt(m2, out)
    res = test(m2, dev, out / "test", batch_size=1, plot_interval=10)
    assert (out / "test" / "preds.pt").exists() and len(res["preds"]) == len(dev)
    assert tuple(res["preds"][0].shape) == (1, 3, 16, 16)
    assert get_best_ckpt(out) is not None


def test_train_on_native_cavity_loader_device_resident(torch, tmp_path):
    """CFDBench-layout cavity files -> native loader with the frames resident on the GPU -> train_auto loop."""
    from cfdbench_amd.harness.args import Args
    from cfdbench_amd.harness.autoregressive import init_model
    from cfdbench_amd.harness.cavity import get_cavity_auto_datasets
    from cfdbench_amd.harness.common import get_output_dir
    from cfdbench_amd.harness.train_auto import train
    from oracle import synth
    root = synth.write_cavity_tree(tmp_path / "data", 5, h=16, w=16)
    tr, dev, te = get_cavity_auto_datasets(root / "cavity", "prop_bc_geo", norm_props=True, norm_bc=True, device="cuda")
    assert tr.inputs.is_cuda and len(tr) > 50
    args = Args(model="fno", data_name="cavity_prop_bc_geo", loss_name="nmse", fno_hidden_dim=8, fno_depth=2,
                fno_modes_x=4, fno_modes_y=4, num_rows=16, num_cols=16, lr=3e-3, output_dir=str(tmp_path / "out"),
                num_epochs=3, batch_size=8, eval_batch_size=8, eval_interval=3, log_interval=50, plot_interval=0)
    torch.manual_seed(0)
    model = init_model(args).cuda()
    losses = train(model, tr, dev, get_output_dir(args, is_auto=True), num_epochs=3, lr=args.lr, batch_size=8,
                   eval_batch_size=8, eval_interval=3, log_interval=50, plot_interval=0)
    assert np.all(np.isfinite(losses)) and np.mean(losses[-5:]) < np.mean(losses[:5])


def _train_graph_rank(rank, world, port, out_dir, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # two ranks on the box's one GPU: host-staged exchange
    try:
        from cfdbench_amd.harness.args import Args
        from cfdbench_amd.harness.autoregressive import init_model
        from cfdbench_amd.harness.data import SyntheticAutoDataset
        from cfdbench_amd.harness.dist_util import broadcast_model_state
        from cfdbench_amd.harness.train_auto import train
        args = Args(model="unet", data_name="cavity_bc", loss_name="nmse", unet_dim=4, lr=2e-3, output_dir=str(Path(out_dir) / "dp"),
                    num_epochs=2, batch_size=4, eval_batch_size=4, eval_interval=2, log_interval=100, plot_interval=0, graph=1)
        torch.manual_seed(100 + rank)  # different initial replicas: train() must broadcast rank 0's
        model = init_model(args).cuda()
        tr = SyntheticAutoDataset(n_cases=9, n_frames=6, height=64, width=64, seed=0)   # 45 frames -> 5 full batches of 4 per rank
        dev = SyntheticAutoDataset(n_cases=2, n_frames=4, height=64, width=64, seed=1)
        losses = train(model, tr, dev, Path(out_dir) / "dp", num_epochs=2, lr=args.lr, batch_size=4, eval_batch_size=4, log_interval=100,
                       eval_interval=2, plot_interval=0, graph=True, device_loader=True)
        torch.cuda.synchronize()
        q.put((rank, [p.detach().cpu().numpy().copy() for p in model.parameters()], [float(v) for v in losses]))
    finally:
        dist.destroy_process_group()


def test_train_auto_graph_option_on_two_ranks(torch, tmp_path):
    """VERDICT r4 next #3 at the harness level: `train_auto --graph 1` under a two-rank process group (the refusal of round 4 is gone): every
    rank replays forward + backward + gradient pack, the flat gradient is all-reduced, Adam replays from the second graph; the replicas --
    started from DIFFERENT seeds, i.e. rank 0's weights are broadcast -- stay bitwise identical through two epochs, the losses are finite and fall."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_graph_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for a, b in zip(res[0][1], res[1][1]):
        assert np.array_equal(a, b), "replicas diverged"
    for _, _, losses in res:
        assert len(losses) >= 8 and np.all(np.isfinite(losses))
    assert np.mean(res[0][2][-3:]) < np.mean(res[0][2][:3])


def test_overlapping_copy_stream_hides_uploads_behind_kernels(torch):
    """harness/data.py:overlapping_copy_stream -- the stream it returns copies pinned host batches to the device WHILE the current stream's
    kernels run (one stream in four shares the compute queue and cannot); double-buffered uploads through it deliver every batch intact."""
    from cfdbench_amd.harness.data import overlapping_copy_stream
    dev = torch.device("cuda", 0)
    side, overlaps = overlapping_copy_stream(dev)
    assert overlaps, "no copy stream overlapped the current stream's kernels"
    cur = torch.cuda.current_stream()
    # the same measurement again on the returned stream: 40 passes over 128 MB on the current stream, a 4-MB copy beside them
    x = torch.empty(32 << 20, device=dev)
    h = torch.arange(1 << 20, dtype=torch.float32).pin_memory()
    d = torch.empty(1 << 20, device=dev)
    d.copy_(h)  # (the first copy out of freshly pinned pages pays for their mapping: 7.8 ms for these 4 MB)
    start, k_end, c_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    start.record(cur)
    for _ in range(40):
        x.mul_(1.0)
    k_end.record(cur)
    side.wait_event(start)
    with torch.cuda.stream(side):
        d.copy_(h, non_blocking=True)
    c_end.record(side)
    torch.cuda.synchronize()
    assert start.elapsed_time(c_end) < 0.5 * start.elapsed_time(k_end)
    assert torch.equal(d.cpu(), h)
    # two buffer sets, batch i + 1 uploaded while "step" i reads batch i: every step sees its own batch
    sets = [torch.empty(1 << 18, device=dev) for _ in range(2)]
    host = [torch.full((1 << 18,), float(i)).pin_memory() for i in range(6)]
    ready, free = [torch.cuda.Event() for _ in range(2)], [torch.cuda.Event() for _ in range(2)]
    for e in free:
        e.record(cur)

    def upload(i):
        side.wait_event(free[i & 1])
        with torch.cuda.stream(side):
            sets[i & 1].copy_(host[i], non_blocking=True)
        ready[i & 1].record(side)
    seen = torch.zeros(6, device=dev)
    upload(0)
    for i in range(6):
        if i + 1 < 6:
            upload(i + 1)
        cur.wait_event(ready[i & 1])
        x.mul_(1.0)  # (something for the upload to hide behind)
        seen[i] = sets[i & 1].mean()
        free[i & 1].record(cur)
    torch.cuda.synchronize()
    assert seen.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]
