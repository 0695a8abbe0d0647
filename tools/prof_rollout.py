#!/usr/bin/env python
"""Dev tool (GPU box): per-kernel HIP-event times of one eager rollout step (cfd_prof), e.g.
    python tools/prof_rollout.py --cases 64 --hidden 32 --height 66 --width 65 [--dtype bf16]"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cfdbench_amd import _lib  # noqa: E402
from cfdbench_amd.models.fno.fno2d import Fno2d  # noqa: E402
from cfdbench_amd.models.loss import loss_name_to_fn  # noqa: E402
from cfdbench_amd.rollout import FnoRollout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=64)
ap.add_argument("--hidden", type=int, default=32)
ap.add_argument("--height", type=int, default=66)
ap.add_argument("--width", type=int, default=65)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
api = _lib.api()
torch.manual_seed(0)
m = Fno2d(2, 2, 5, loss_name_to_fn("nmse"), 4, 12, 12, a.hidden).cuda().eval()
x = torch.randn(a.cases, 2, a.height, a.width).cuda()
cp = torch.randn(a.cases, 5).cuda()
mask = torch.ones(a.cases, 1, a.height, a.width).cuda()
ro = FnoRollout(m, dtype=a.dtype)
st = ro._build(a.cases, 2, a.height, a.width, 5, a.steps, True, x.device)  # buffers + graph; the eager loop below re-runs its body
shape, plan = st["shape"], st["plan"]
s = torch.cuda.current_stream().cuda_stream
api.call("cfd_prof_begin")
for t in range(a.steps):
    api.call("cfd_fno_forward_ex", plan, ctypes.byref(shape), ctypes.byref(st["pstruct"]), st["frames"][t].data_ptr(), st["cp"].data_ptr(),
             st["mask"].data_ptr(), None, st["frames"][t + 1].data_ptr(), None, st["ws"].data_ptr(), 0, ro.act_dtype, s)
torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
api.call("cfd_prof_end", buf, len(buf))
rows = [ln.split() for ln in buf.value.decode().splitlines()]
tot = sum(float(r[2]) for r in rows)
print(f"# {a.cases} cases, hidden {a.hidden}, {a.height}x{a.width}, {a.dtype}: {tot / a.steps * 1e3:.1f} us of kernels per step")
for r in sorted(rows, key=lambda r: -float(r[2])):
    n = int(r[1])
    print(f"{r[0]:22s} {n // a.steps:3d} launches/step  {float(r[2]) / n * 1e3:8.2f} us each  {float(r[2]) / tot * 100:5.1f} %")
import time
t0 = time.perf_counter()
for _ in range(3):
    ro.generate_frames(x, cp, mask, a.steps)
torch.cuda.synchronize()
print(f"graph replay: {(time.perf_counter() - t0) / 3 / a.steps * 1e3:.1f} us per step")
