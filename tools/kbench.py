#!/usr/bin/env python
"""Dev tool (GPU box): time every C-ABI entry point on its own at the BASELINE configs[1] shapes.

    python tools/kbench.py [--batch 256] [--hidden 20] [--reps 30] [--only dft,idft]

Prints one line per kernel: average microseconds (HIP events on the launch stream) and the algorithmic GB/s.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cfdbench_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=20)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--json", type=str, default="")
    ap.add_argument("--dump-out", type=int, default=0, help="print the first N int64 words of the output tensor after one call of --only (kernel timestamp experiments)")
    ap.add_argument("--zeros", action="store_true", help="all-zero activations / gradients / labels (DVFS probe: the same instruction stream at "
                                                         "the lowest switching power; MI355X_MICROARCH.md)")
    ap.add_argument("--tune", type=str, default="", help="dispatch knobs, e.g. general_b3=0,fused_variant=0 (cfd_tune_set)")
    args = ap.parse_args()
    api = _lib.api()
    for kv in [s for s in args.tune.split(",") if s]:
        k, v = kv.split("=")
        api.call("cfd_tune_set", k.encode(), int(v))
    dev = torch.device("cuda", 0)
    B, C, H, W = args.batch, args.hidden, args.height, args.width
    HW, m1, m2 = H * W, 12, 12
    M = 2 * m1 * m2
    plan = _lib.plan(H, W, m1, m2, 0)
    st = torch.cuda.current_stream().cuda_stream
    f = (lambda *s: torch.zeros(*s, device=dev)) if args.zeros else (lambda *s: torch.randn(*s, device=dev))  # noqa: E731
    a, a2, g, out = f(B, C, H, W), f(B, C, H, W), f(B, C, H, W), f(B, C, H, W)
    xh, gh, z = f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2)
    w1, w2 = f(C, C, m1, m2, 2) / (C * C), f(C, C, m1, m2, 2) / (C * C)
    gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
    w0, b0 = f(C, C) / C, f(C)
    gw0, gb0 = torch.empty_like(w0), torch.empty_like(b0)
    inputs, label, mask, cp = f(B, 2, H, W), f(B, 2, H, W), torch.ones(B, 1, H, W, device=dev), f(B, 5)
    fc0w, fc0b = f(C, 10), f(C)
    gfc0w, gfc0b = torch.empty_like(fc0w), torch.empty_like(fc0b)
    fc1w, fc1b, fc2w, fc2b = f(128, C) / C ** 0.5, f(128), f(2, 128) / 11.3, f(2)
    g1w, g1b, g2w, g2b = (torch.empty_like(t) for t in (fc1w, fc1b, fc2w, fc2b))
    preds, sums, coef = f(B, 2, H, W), torch.zeros(4, device=dev), torch.tensor([1e-6, 0.0], device=dev)
    ws_n = max(api.size("cfd_spectral_wgrad_workspace_bytes", plan, B, C, C),
               api.size("cfd_chan_wgrad_workspace_bytes", B, C, C, HW),
               api.size("cfd_fno_head_workspace_bytes", B, C, 128, 2, HW),
               api.size("cfd_fno_stem_bwd_workspace_bytes", plan, B, 2, 5, C),
               api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, C, C))
    ws = torch.empty(ws_n + 256, dtype=torch.uint8, device=dev)
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    N = B * C * HW * 4
    Mb = B * C * M * 8
    Wb = 2 * C * C * (M // 2) * 8
    px = B * HW

    cases = {
        "dft": (lambda: api.call("cfd_spectral_dft", plan, P(a), P(xh), B * C, 0, st), N + Mb),
        "dft_act": (lambda: api.call("cfd_spectral_dft", plan, P(a), P(xh), B * C, 1, st), N + Mb),
        "mix": (lambda: api.call("cfd_spectral_mix", plan, P(xh), P(w1), P(w2), P(z), B, C, C, 0, st), 2 * Mb + Wb),
        "mix_adj": (lambda: api.call("cfd_spectral_mix", plan, P(gh), P(w1), P(w2), P(z), B, C, C, 1, st), 2 * Mb + Wb),
        "spec_wgrad": (lambda: api.call("cfd_spectral_wgrad", plan, P(xh), P(gh), P(gw1), P(gw2), P(ws), B, C, C, st),
                       2 * Mb + Wb),
        "mix_adj_wgrad": (lambda: api.call("cfd_spectral_mix_adj_wgrad", plan, P(xh), P(gh), P(w1), P(w2), P(z), P(gw1), P(gw2),
                                           P(ws), B, C, C, st), 3 * Mb + 2 * Wb),
        "idft": (lambda: api.call("cfd_spectral_idft", plan, P(z), None, None, P(out), B * C, 0, st), N + Mb),
        "idft_add": (lambda: api.call("cfd_spectral_idft", plan, P(z), P(out), None, P(out), B * C, 1, st), 2 * N + Mb),
        "idft_add_dgelu": (lambda: api.call("cfd_spectral_idft", plan, P(z), P(out), P(a2), P(out), B * C, 2, st),
                           3 * N + Mb),
        "block_fwd": (lambda: api.call("cfd_fno_block_fwd", plan, P(a), P(z), P(w0), P(b0), P(out), B, C, C, 0, st), 2 * N + Mb),
        "block_fwd_act": (lambda: api.call("cfd_fno_block_fwd", plan, P(a), P(z), P(w0), P(b0), P(out), B, C, C, 1, st), 2 * N + Mb),
        "block_bwd": (lambda: api.call("cfd_fno_block_bwd_input", plan, P(g), P(z), P(w0), None, P(out), B, C, C, st), 2 * N + Mb),
        "block_bwd_dgelu": (lambda: api.call("cfd_fno_block_bwd_input", plan, P(g), P(z), P(w0), P(a2), P(out), B, C, C, st),
                            3 * N + Mb),
        "chanmix": (lambda: api.call("cfd_chanmix", P(a), P(w0), P(b0), P(out), B, C, C, HW, 0, 0, st), 2 * N),
        "chanmix_act": (lambda: api.call("cfd_chanmix", P(a), P(w0), P(b0), P(out), B, C, C, HW, 1, 0, st), 2 * N),
        "chanmix_t": (lambda: api.call("cfd_chanmix", P(g), P(w0), None, P(out), B, C, C, HW, 0, 1, st), 2 * N),
        "chan_wgrad": (lambda: api.call("cfd_chan_wgrad", P(g), P(a), P(gw0), P(gb0), P(ws), B, C, C, HW, 0, st), 2 * N),
        "chan_wgrad_act": (lambda: api.call("cfd_chan_wgrad", P(g), P(a), P(gw0), P(gb0), P(ws), B, C, C, HW, 1, st), 2 * N),
        "stem_fwd": (lambda: api.call("cfd_fno_stem_fwd", plan, P(inputs), P(mask), P(cp), P(fc0w), P(fc0b), P(out), B, 2, 5,
                                      C, st), N + px * 12),
        "stem_bwd": (lambda: api.call("cfd_fno_stem_bwd", plan, P(g), P(inputs), P(mask), P(cp), P(gfc0w), P(gfc0b), P(ws), B,
                                      2, 5, C, st), N + px * 12),
        "head_fwd": (lambda: api.call("cfd_fno_head_fwd", P(a), P(mask), P(label), P(fc1w), P(fc1b), P(fc2w), P(fc2b), P(preds),
                                      P(sums), P(ws), B, C, 128, 2, HW, 1, st), N + px * 28),
        "head_bwd": (lambda: api.call("cfd_fno_head_bwd", P(a), P(mask), P(label), P(preds), None, P(coef), P(fc1w), P(fc1b),
                                      P(fc2w), P(out), P(g1w), P(g1b), P(g2w), P(g2b), P(ws), B, C, 128, 2, HW, 1, st),
                     2 * N + px * 28),
        "head_train": (lambda: api.call("cfd_fno_head_train", P(a), P(mask), P(label), P(coef), P(fc1w), P(fc1b), P(fc2w), P(fc2b), P(preds),
                                        P(sums), P(out), P(g1w), P(g1b), P(g2w), P(g2b), P(ws), B, C, 128, 2, HW, 1, st), 2 * N + px * 28),
        "spectral_fwd": (lambda: api.call("cfd_spectral_conv2d_fwd", plan, P(a), P(w1), P(w2), P(out), P(xh), P(z), B, C, C, st),
                         2 * N + Wb),
        "spectral_bwd": (lambda: api.call("cfd_spectral_conv2d_bwd", plan, P(g), P(xh), P(w1), P(w2), P(out), P(gw1), P(gw2),
                                          P(ws), B, C, C, st), 3 * N + 2 * Wb),
    }
    def pair():
        cases["spectral_fwd"][0]()
        cases["spectral_bwd"][0]()
    cases["spectral_fwd+bwd"] = (pair, 5 * N + 3 * Wb)
    only = [s for s in args.only.split(",") if s]
    print(f"# B={B} C={C} {H}x{W} tune='{args.tune}'", flush=True)
    rows = {}
    for name, (fn, nbytes) in cases.items():
        if only and not any(name == o or name.startswith(o) for o in only):
            continue
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            rows[name] = dict(us=round(us, 2), gbs=round(nbytes / us / 1e3, 1), frac_hbm=round(nbytes / us / 1e3 / 8000, 3))
            print(f"{name:18s} {us:9.2f} us   {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / us / 1e3 / 80:5.1f}% of 8 TB/s)",
                  flush=True)
            if args.dump_out:
                words = (preds if name.startswith('head_fwd') else out).reshape(-1)[:2 * args.dump_out].view(torch.int64).cpu().tolist()
                print("dump:", " ".join(str(w) for w in words), flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{name:18s} FAILED: {e}", flush=True)
    if args.json:
        Path(args.json).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
