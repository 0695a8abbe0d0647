#!/usr/bin/env python
"""Dev experiment (GPU box): time the mode-domain kernels (mix, adjoint mix, spectral weight gradient) and the
SpectralConv2d forward+backward pair under the dispatch knobs of csrc/tune.cpp (cfd_tune_set), all in one process.

    python tools/exp/mode_variants.py [--batch 256] [--reps 50]
"""
import argparse
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from cfdbench_amd import _lib  # noqa: E402

CONFIGS = [
    ("lane=mode mix, unfused", dict(mix_nwv=0, fused_variant=0)),
    ("default", dict()),
    ("mix lds nwv1", dict(mix_nwv=1)),
    ("mix lds nwv2", dict(mix_nwv=2)),
    ("mix lds nwv4", dict(mix_nwv=4)),
    ("mix lds nwv8", dict(mix_nwv=8)),
    ("unfused", dict(fused_variant=0)),
]
KEYS = ("wgrad_wg", "mix_nwv", "fused_variant")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--configs", type=str, default="", help="comma-separated indices into CONFIGS (default: all)")
    args = ap.parse_args()
    api = _lib.api()
    dev = torch.device("cuda", 0)
    B, C, H, W, m1, m2 = args.batch, 20, 64, 64, 12, 12
    plan = _lib.plan(H, W, m1, m2, 0)
    st = torch.cuda.current_stream().cuda_stream
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    x, gy, y, gx = f(B, C, H, W), f(B, C, H, W), f(B, C, H, W), f(B, C, H, W)
    xh, gh, z = f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2)
    w1, w2 = f(C, C, m1, m2, 2) / (C * C), f(C, C, m1, m2, 2) / (C * C)
    gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
    P = lambda t: t.data_ptr()  # noqa: E731

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3

    ref = {}
    sel = [int(i) for i in args.configs.split(",") if i] or range(len(CONFIGS))
    for name, env in [CONFIGS[i] for i in sel]:
        for k in KEYS:
            api.call("cfd_tune_set", k.encode(), int(env.get(k, -1)))
        ws = torch.empty(api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, C, C) + 256, dtype=torch.uint8, device=dev)
        cases = {
            "mix": lambda: api.call("cfd_spectral_mix", plan, P(xh), P(w1), P(w2), P(z), B, C, C, 0, st),
            "mix_adj": lambda: api.call("cfd_spectral_mix", plan, P(gh), P(w1), P(w2), P(z), B, C, C, 1, st),
            "wgrad": lambda: api.call("cfd_spectral_wgrad", plan, P(xh), P(gh), P(gw1), P(gw2), P(ws), B, C, C, st),
            "adj+wgrad": lambda: api.call("cfd_spectral_mix_adj_wgrad", plan, P(xh), P(gh), P(w1), P(w2), P(z), P(gw1), P(gw2),
                                          P(ws), B, C, C, st),
        }

        def pair():
            api.call("cfd_spectral_conv2d_fwd", plan, P(x), P(w1), P(w2), P(y), P(xh), P(z), B, C, C, st)
            api.call("cfd_spectral_conv2d_bwd", plan, P(gy), P(xh), P(w1), P(w2), P(gx), P(gw1), P(gw2), P(ws), B, C, C, st)

        row = {k: timed(fn) for k, fn in cases.items()}
        xh_keep = xh.clone()
        row["fwd+bwd"] = timed(pair)
        # results of this configuration vs the first one (same inputs): max relative deviation
        api.call("cfd_spectral_mix", plan, P(xh_keep), P(w1), P(w2), P(z), B, C, C, 0, st)
        api.call("cfd_spectral_wgrad", plan, P(xh_keep), P(gh), P(gw1), P(gw2), P(ws), B, C, C, st)
        torch.cuda.synchronize()
        cur = dict(z=z.clone(), gw1=gw1.clone(), gw2=gw2.clone())
        if not ref:
            ref = cur
        dev_ = max(float((cur[k] - ref[k]).norm() / ref[k].norm()) for k in cur)
        xh.copy_(xh_keep)
        alg = 5 * B * C * H * W * 4 + 3 * 2 * C * C * 144 * 8
        print(f"{name:38s} mix {row['mix']:6.2f}  mix_adj {row['mix_adj']:6.2f}  wgrad {row['wgrad']:6.2f}  adj+wgrad {row['adj+wgrad']:6.2f}  "
              f"fwd+bwd {row['fwd+bwd']:7.2f} us = {alg / row['fwd+bwd'] / 1e3 / 80:5.1f}% of 8 TB/s   rel.dev {dev_:.1e}", flush=True)


if __name__ == "__main__":
    main()
