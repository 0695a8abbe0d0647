#!/usr/bin/env python
"""A/B timing of the Auto-FNO train step (bench.py's headline workload) under tune-knob settings, same process, interleaved.

    python tools/exp/ab_step.py "side_stream=0" "side_stream=1" [--batch 256] [--rounds 4] [--steps 30] [--hw 64 64]
Each config is a comma-separated list of knob=value pairs ("" = defaults).  Prints best / median ms per step per config.
"""
import argparse
import statistics
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--hw", type=int, nargs=2, default=(64, 64))
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--prof", action="store_true", help="per-kernel HIP-event times of the first config (cfd_prof)")
    a = ap.parse_args()
    from cfdbench_amd import _lib
    from cfdbench_amd.engine import FnoTrainEngine
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn
    api = _lib.api()
    dev = torch.device("cuda", 0)
    B, (H, W) = a.batch, a.hw
    torch.manual_seed(0)
    model = Fno2d(2, 2, 5, loss_name_to_fn("nmse"), 4, 12, 12, a.hidden).to(dev)
    eng = FnoTrainEngine(model, lr=1e-3, loss_name="nmse")
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 2, H, W, generator=g).to(dev)
    y = (x.cpu() + 0.1 * torch.randn(B, 2, H, W, generator=g)).to(dev)
    cp = torch.randn(B, 5, generator=g).to(dev)
    mask = torch.ones(B, 1, H, W, device=dev)
    step = eng.train_step_graph if a.graph else eng.train_step

    def apply(cfg, reset=False):
        knobs = []
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            api.call("cfd_tune_set", k.encode(), -1 if reset else int(v))
            knobs.append(k)
        return knobs

    res = {c: [] for c in a.configs}
    for r in range(a.rounds):
        for c in a.configs:
            apply(c)
            if a.graph:
                eng._graph = None
            for _ in range(5):
                step(x, y, cp, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step(x, y, cp, mask)
            torch.cuda.synchronize()
            res[c].append((time.perf_counter() - t0) / a.steps * 1e3)
            apply(c, reset=True)
    for c, v in res.items():
        print(f"{c or '(defaults)':40s} best {min(v):.4f} ms  median {statistics.median(v):.4f} ms  all {[round(t, 4) for t in v]}", flush=True)
    print("nmse", eng.scores()["nmse"])
    if a.prof:
        import ctypes
        apply(a.configs[0])
        api.call("cfd_prof_begin")
        for _ in range(10):
            eng.train_step(x, y, cp, mask)
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 16)
        api.call("cfd_prof_end", buf, len(buf))
        rows = [ln.split() for ln in buf.value.decode().splitlines()]
        tot = sum(float(r[2]) for r in rows)
        print(f"# per-kernel times, config {a.configs[0]!r}: {tot / 10 * 1e3:.1f} us of kernels per step")
        for r in sorted(rows, key=lambda r: -float(r[2])):
            n = int(r[1])
            gbs = float(r[3]) / (float(r[2]) * 1e-3) / 1e9 if float(r[2]) > 0 else 0
            print(f"  {r[0]:22s} {n // 10:3d}/step {float(r[2]) / n * 1e3:8.2f} us  {gbs:7.0f} GB/s  {float(r[2]) / tot * 100:5.1f} %")


if __name__ == "__main__":
    main()
