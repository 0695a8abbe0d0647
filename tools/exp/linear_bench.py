#!/usr/bin/env python
"""GPU box: time of cfd_linear_fwd / the input-gradient half of cfd_linear_bwd at the tall shapes of the FFN-family legs, per gemm_b3 setting.
usage: linear_bench.py [M K N]..."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cfdbench_amd import _lib  # noqa: E402

api = _lib.api()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
shapes = [(131072, 200, 200), (256000, 100, 100), (131072, 512, 512)]
if len(sys.argv) > 3:
    shapes = [tuple(int(v) for v in sys.argv[1:4])]
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    line = f"M={M} K={K} N={N}:"
    for knob in (-1, 0):
        api.call("cfd_tune_set", b"gemm_b3", knob)
        ws = torch.empty(max(api.size("cfd_linear_fwd_workspace_bytes", M, K, N), 256), dtype=torch.uint8, device=dev)
        fn = lambda: api.call("cfd_linear_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, ws.data_ptr(), M, K, N, 1, st)  # noqa: E731
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        line += f"  gemm_b3={knob}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us"
    api.call("cfd_tune_set", b"gemm_b3", -1)
    print(line, flush=True)
