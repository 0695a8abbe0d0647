#!/usr/bin/env python
"""GPU box: one FNO rollout configuration, eager launches (so that rocprofv3 --kernel-trace sees every kernel), then the HIP-graph rate.
usage: rollout_trace.py H W hidden cases steps dtype"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cfdbench_amd.models.fno.fno2d import Fno2d  # noqa: E402
from cfdbench_amd.models.loss import loss_name_to_fn  # noqa: E402
from cfdbench_amd.rollout import FnoRollout  # noqa: E402

H, W, C, B, steps = (int(v) for v in sys.argv[1:6])
dtype = sys.argv[6] if len(sys.argv) > 6 else "f32"
torch.manual_seed(0)
m = Fno2d(2, 2, 5, loss_name_to_fn("nmse"), 4, 12, 12, C).cuda().eval()
g = torch.Generator().manual_seed(1)
x0, cp, mask = torch.randn(B, 2, H, W, generator=g).cuda(), torch.randn(B, 5, generator=g).cuda(), torch.ones(B, 1, H, W).cuda()
with torch.no_grad():
    for _ in range(2):
        m.generate_many(x0, cp, mask, 5)  # eager launches: what the kernel trace records
torch.cuda.synchronize()
ro = FnoRollout(m, **({} if dtype == "f32" else dict(dtype=dtype)))
ro.generate_frames(x0, cp, mask, steps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ro.generate_frames(x0, cp, mask, steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"rollout {H}x{W} C={C} cases={B} {dtype}: {dt / steps * 1e6:.1f} us/step, {B * steps / dt:.0f} frames/s")
