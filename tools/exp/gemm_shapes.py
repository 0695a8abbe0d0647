#!/usr/bin/env python
"""Times of the five stand-alone fp32 GEMMs of an Auto-DeepONet configs[3] train step (B = 512, 66x65, width 100) under forced split-K
counts (cfd_tune_set("gemm_splits", s)):   python tools/exp/gemm_shapes.py [--splits 0,8,16,32,64]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cfdbench_amd import _lib  # noqa: E402

SHAPES = [  # (name, M, N, K, trans_a, trans_b)
    ("branch_l1_fwd   x W1^T", 512, 100, 4295, 0, 1),
    ("inner           b t^T", 512, 4290, 100, 0, 1),
    ("g_branch        gP t", 512, 100, 4290, 0, 0),
    ("g_trunk         gP^T b", 4290, 100, 512, 1, 0),
    ("branch_l1_wgrad gZ^T x", 100, 4295, 512, 1, 0),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splits", default="0,4,8,16,32,64")
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    api = _lib.api()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    for name, M, N, K, ta, tb in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        line = f"{name:26s} M={M:5d} N={N:5d} K={K:5d}:"
        for s in [int(v) for v in a.splits.split(",")]:
            api.call("cfd_tune_set", b"gemm_splits", s if s > 0 else -1)
            ws_n = api.size("cfd_gemm_workspace_bytes", M, N, K)
            ws = torch.empty(max(ws_n, 256), dtype=torch.uint8, device=dev)
            fn = lambda: api.call("cfd_gemm", A.data_ptr(), B.data_ptr(), C.data_ptr(), ws.data_ptr() if ws_n else None, M, N, K,  # noqa: E731
                                  A.shape[1], B.shape[1], N, ta, tb, st)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            line += f"  s={s or 'auto'}:{e0.elapsed_time(e1) / a.reps * 1e3:6.1f}"
        api.call("cfd_tune_set", b"gemm_splits", -1)
        print(line, flush=True)


if __name__ == "__main__":
    main()
