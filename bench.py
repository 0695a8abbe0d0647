#!/usr/bin/env python
"""Headline benchmark: Auto-FNO training frames/s on synthetic (B,2,64,64) batches (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + nMSE loss + backward + (N>1: RCCL all-reduce of the flat gradient) + Adam on one batch that is
already resident in HBM.  Workload = BASELINE.json configs[1]: Fno2d(in=2,out=2,p=5,L=4,hidden=20,modes=12), B=256 per
GPU, 64x64, fp32.  Weak scaling: every rank processes its own B frames; value = N*B*K / max-over-ranks time.
Rank 0 prints ONE JSON line (with the `roofline` and `cpu_baseline` objects).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (BASELINE configs[1]: 256)")
    ap.add_argument("--hidden", type=int, default=20)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--n-case-params", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay forward+backward from a HIP graph")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--rollout-batch", type=int, default=64)
    ap.add_argument("--rollout-steps", type=int, default=200)
    return ap.parse_args()


def kernel_models(B, C, HW, L, M, head=128, co=2):
    """Algorithmic HBM bytes and flops of ONE launch of each kernel."""
    N = B * C * HW * 4          # one activation tensor
    Mb = B * C * M * 8          # one kept-mode tensor (complex64)
    Wb = 2 * C * C * (M // 2) * 8  # weights1+weights2 of one layer
    px = B * HW
    io = px * 4 * (2 + 1 + 2 + co)  # inputs, mask, label, preds of the head
    return {
        "k_stem_fwd": dict(bytes=px * 4 * 3 + N, flops=2 * px * C * 10),
        "k_dft_fwd": dict(bytes=N + Mb, flops=2 * B * C * (36 * 64 * 32 + 64 * 32 * 32)),
        "k_dft_fwd_act": dict(bytes=N + Mb, flops=2 * B * C * (36 * 64 * 32 + 64 * 32 * 32)),
        "k_mix": dict(bytes=2 * Mb + Wb, flops=8 * B * C * C * M),
        "k_mix_adj": dict(bytes=2 * Mb + Wb, flops=8 * B * C * C * M),
        "k_chanmix": dict(bytes=2 * N, flops=2 * px * C * C),
        "k_chanmix_act": dict(bytes=2 * N, flops=2 * px * C * C),
        "k_chanmix_t": dict(bytes=2 * N, flops=2 * px * C * C),
        "k_block_fwd": dict(bytes=2 * N + Mb, flops=2 * px * C * C + 2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_block_fwd_act": dict(bytes=2 * N + Mb, flops=2 * px * C * C + 2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_block_bwd": dict(bytes=2 * N + Mb, flops=2 * px * C * C + 2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_block_bwd_dgelu": dict(bytes=3 * N + Mb, flops=2 * px * C * C + 2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_idft": dict(bytes=N + Mb, flops=2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_idft_add": dict(bytes=2 * N + Mb, flops=2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_idft_add_dgelu": dict(bytes=3 * N + Mb, flops=2 * B * C * (28 * 32 * 64 + 24 * 64 * 64)),
        "k_spec_wgrad_part": dict(bytes=2 * Mb + Wb, flops=8 * B * C * C * M),
        "k_mixadj_wgrad": dict(bytes=3 * Mb + 2 * Wb, flops=16 * B * C * C * M),  # adjoint mix + weight gradient, one launch
        "k_spec_wgrad_reduce": dict(bytes=2 * Wb, flops=0),
        "k_chan_wgrad": dict(bytes=2 * N, flops=2 * px * C * C),
        "k_chan_wgrad_stem": dict(bytes=N + px * 12, flops=2 * px * C * 10),
        "k_wgrad_reduce": dict(bytes=0, flops=0),
        "k_head_fwd": dict(bytes=N + io, flops=2 * px * (C * head + head * co)),
        "k_head_bwd": dict(bytes=2 * N + io, flops=2 * px * (3 * C * head + 2 * head * co)),
        "k_adam": dict(bytes=0, flops=0),
    }


def roofline_of(name, avg_ms, model):
    m = model.get(name)
    if not m or avg_ms <= 0:
        return None
    t = avg_ms * 1e-3
    gbs = m["bytes"] / t / 1e9
    tfs = m["flops"] / t / 1e12
    t_hbm = m["bytes"] / (HBM_PEAK_GBS * 1e9)
    t_mfma = m["flops"] / (FP32_MFMA_PEAK_TF * 1e12)
    if t_mfma > t_hbm:
        # fp32-equivalent flops of the kernel's GEMMs against the 157.3 TF fp32 ceiling (fp32 MFMA == VALU fp32 rate on
        # gfx950, tools/exp/valu_rate.hip).  The head kernels run those GEMMs as 3-term split-bf16 MFMAs and are bound
        # by the VALU work of GELU / gelu', which shares that ceiling.
        return dict(kernel=name, bound="mfma", achieved=round(tfs, 3), peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s",
                    frac=round(tfs / FP32_MFMA_PEAK_TF, 4), avg_us=round(avg_ms * 1e3, 2), traffic=None,
                    note="fp32-equivalent GEMM flops vs the fp32 MFMA/VALU ceiling; GEMMs issue as split-bf16 MFMA, "
                         "the kernel is VALU(GELU)-bound (profiles/*_step_busy_counters.txt)")
    return dict(kernel=name, bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(gbs / HBM_PEAK_GBS, 4), avg_us=round(avg_ms * 1e3, 2), traffic=None)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback", file=sys.stderr)
        sys.exit(1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from cfdbench_amd import _lib
    from cfdbench_amd.engine import FnoTrainEngine
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn

    B, C, L, H, W, p = args.batch, args.hidden, args.layers, args.height, args.width, args.n_case_params
    torch.manual_seed(0)  # identical weights on every rank
    model = Fno2d(2, 2, p, loss_name_to_fn("nmse"), L, 12, 12, C).to(dev)
    eng = FnoTrainEngine(model, lr=1e-3, loss_name="nmse")
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)  # SURVEY.md 8(d) synthetic inputs; a shard per rank
    inputs = torch.randn(B, 2, H, W, generator=g).to(dev)
    label = (inputs.cpu() + 0.1 * torch.randn(B, 2, H, W, generator=g)).to(dev)
    cp = torch.randn(B, p, generator=g).to(dev)
    mask = torch.ones(B, 1, H, W, device=dev)
    step = eng.train_step_graph if args.graph else eng.train_step

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step(inputs, label, cp, mask)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(inputs, label, cp, mask)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final = eng.scores()

    result = {
        "metric": "train frames/sec (64x64x2), Auto-FNO cavity", "value": round(world * B * args.steps / elapsed, 1),
        "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: Auto-FNO train step (fwd+nMSE+bwd+Adam), Fno2d(L={L},hidden={C},modes=12,p={p}), "
                               f"{H}x{W}, batch {B}/GPU, fp32, random-init weights",
                   "precision": "fp32 storage and accumulation; DFT / 1x1-weight-gradient / head contractions as 3-term "
                                "split-bf16 MFMA products (rel. error <= 2^-16, measured nMSE vs fp64 oracle <= 5e-11)",
                   "global_batch": world * B, "parallelism": f"dp{world}", "graph": bool(args.graph)},
        "final_nmse": round(final["nmse"], 6),
    }

    # ---- roofline leg: the same K steps again with per-kernel HIP events on the launch stream -------------------
    # EVERY rank runs these steps (they contain the gradient all-reduce, a collective); only rank 0 records events.
    if not args.no_roofline:
        import ctypes
        api = _lib.api()
        if rank == 0:
            api.call("cfd_prof_begin")
        for _ in range(args.steps):
            eng.train_step(inputs, label, cp, mask)
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        buf = ctypes.create_string_buffer(1 << 16)
        api.call("cfd_prof_end", buf, len(buf))
        rows = []
        for line in buf.value.decode().splitlines():
            name, cnt, tot = line.split()
            rows.append((name, int(cnt), float(tot)))
        model_bytes = kernel_models(B, C, H * W, L, 2 * 12 * 12)
        rows.sort(key=lambda r: -r[2])
        tot_ms = sum(r[2] for r in rows)
        kern = []
        for name, cnt, tot in rows:
            rl = roofline_of(name, tot / cnt, model_bytes)
            kern.append(dict(kernel=name, launches_per_step=cnt // args.steps, avg_us=round(tot / cnt * 1e3, 2),
                             share=round(tot / tot_ms, 4), bound=rl and rl["bound"], frac=rl and rl["frac"]))
        result["kernels"] = kern
        dom = rows[0]
        result["roofline"] = roofline_of(dom[0], dom[2] / dom[1], model_bytes)
        # HBM traffic per launch of the dominant kernel: rocprofv3 PMC passes of this same command
        # (tools/profile_step.sh: FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE), committed under profiles/
        try:
            pmc = json.loads(sorted((REPO / "profiles").glob("r*_pmc_traffic.json"))[-1].read_text())
            match = [v for k, v in pmc.items() if k.startswith(dom[0] + "<") or k == dom[0]]
            if match and B == 256 and C == 20:
                result["roofline"]["traffic"] = int(sum(m["traffic_bytes"] for m in match) / len(match))
                result["roofline"]["traffic_source"] = "profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, per launch)"
        except Exception:  # noqa: BLE001
            pass
        # north-star kernel group: SpectralConv2d forward+backward alone, algorithmic bytes 5N + 3Wb
        plan = _lib.plan(H, W, 12, 12, dev.index)
        x = torch.randn(B, C, H, W, device=dev)
        gy = torch.randn(B, C, H, W, device=dev)
        w1 = torch.view_as_real(torch.rand(C, C, 12, 12, dtype=torch.cfloat, device=dev) / (C * C)).contiguous()
        w2 = torch.view_as_real(torch.rand(C, C, 12, 12, dtype=torch.cfloat, device=dev) / (C * C)).contiguous()
        y, gx = torch.empty_like(x), torch.empty_like(x)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        xh = torch.empty(B, C, 24, 12, 2, device=dev)
        z = torch.empty(B, C, 24, 12, 2, device=dev)
        ws = torch.empty(api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, C, C), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def spec():
            api.call("cfd_spectral_conv2d_fwd", plan, x.data_ptr(), w1.data_ptr(), w2.data_ptr(), y.data_ptr(), xh.data_ptr(),
                     z.data_ptr(), B, C, C, st)
            api.call("cfd_spectral_conv2d_bwd", plan, gy.data_ptr(), xh.data_ptr(), w1.data_ptr(), w2.data_ptr(), gx.data_ptr(),
                     gw1.data_ptr(), gw2.data_ptr(), ws.data_ptr(), B, C, C, st)

        for _ in range(3):
            spec()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(args.steps, 20)
        e0.record()
        for _ in range(reps):
            spec()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        Nb = B * C * H * W * 4
        Wb = 2 * C * C * 144 * 8
        alg = 5 * Nb + 3 * Wb
        gbs = alg / (us * 1e-6) / 1e9
        result["roofline_spectral_conv2d"] = dict(
            what="SpectralConv2d fwd+bwd (dft, mix, idft | dft, mix_adj + wgrad in one launch, idft carrying the partial-sum reduction)", bound="hbm",
            algorithmic_bytes=alg, avg_us=round(us, 2), achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(gbs / HBM_PEAK_GBS, 4), traffic=None)

    # ---- rollout leg (rank 0): batched multi-step inference from one HIP graph (BASELINE metric's "rollout" half) ---
    if rank == 0 and world == 1 and not args.no_rollout:  # N=1 only: keeps the multi-rank runs short
        from cfdbench_amd.rollout import FnoRollout
        Br, steps_r = args.rollout_batch, args.rollout_steps
        ro = FnoRollout(model)
        x0 = inputs[:Br].contiguous()
        fr = ro.generate_many(x0, cp[:Br].contiguous(), mask[:Br].contiguous(), steps_r)  # builds + captures the graph
        torch.cuda.synchronize()
        t0r = time.perf_counter()
        reps_r = 3
        for _ in range(reps_r):
            fr = ro.generate_many(x0, cp[:Br].contiguous(), mask[:Br].contiguous(), steps_r)
        torch.cuda.synchronize()
        dtr = (time.perf_counter() - t0r) / reps_r
        with torch.no_grad():
            plain = model.generate_many(x0, cp[:Br].contiguous(), mask[:Br].contiguous(), 3)
        result["rollout"] = dict(
            what=f"FnoRollout.generate_many: {Br} cases x {steps_r} steps, {H}x{W}, fp32, one HIP graph per horizon",
            frames_per_s=round(Br * steps_r / dtr, 1), ms_per_step=round(dtr / steps_r * 1e3, 4),
            bitwise_equal_to_generate_many=bool(all(torch.equal(a, b) for a, b in zip(plain, fr[:3]))))

    # ---- CPU baseline leg (rank 0, N=1): the reference's ATen call sequence on the host cores ------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_port
        ncpu = os.cpu_count() or 1
        # ATen's CPU kernels do not scale to every hardware thread at this size (256 threads on the 2x64-core GPU box is
        # pathologically slow): probe two moderate thread counts with one timed step each, keep the faster, and stop
        # probing as soon as ~20 s are spent so the default run stays within minutes.
        cands = sorted({t for t in (16, 64) if t <= ncpu} or {ncpu})
        probe, t_probe0 = {}, time.perf_counter()
        for t in cands:
            probe[t] = torch_port.time_train_steps(args.cpu_batch, 1, warmup=1, C=C, L=L, H=H, W=W, p=p, threads=t)["frames_per_s"]
            if time.perf_counter() - t_probe0 > 20.0:
                break
        best_t = max(probe, key=probe.get)
        cb = torch_port.time_train_steps(args.cpu_batch, args.cpu_steps, warmup=1, C=C, L=L, H=H, W=W, p=p, threads=best_t)
        result["cpu_baseline"] = dict(
            value=round(cb["frames_per_s"], 1), unit="frames/s", cores=cb["threads"], kind="port",
            sample=f"same train step (fwd+nMSE+bwd+Adam) via oracle/torch_port.py (PyTorch-CPU ATen ops, fp32), batch "
                   f"{args.cpu_batch} x {args.cpu_steps} steps after 1 warm-up, median; best of thread counts "
                   f"{ {t: round(v, 1) for t, v in probe.items()} } on {ncpu} host cores")

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()  # the other ranks wait here while rank 0 finishes its (collective-free) extra legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
