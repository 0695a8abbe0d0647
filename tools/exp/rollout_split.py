#!/usr/bin/env python
"""GPU box: a rollout of N cases as ONE graph against the same cases in K groups, each group's whole-horizon graph on its own stream
(cases are independent over the entire horizon: no fork / join inside).  usage: rollout_split.py H W hidden cases steps"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cfdbench_amd.models.fno.fno2d import Fno2d  # noqa: E402
from cfdbench_amd.models.loss import loss_name_to_fn  # noqa: E402
from cfdbench_amd.rollout import FnoRollout  # noqa: E402

H, W, C, B, steps = (int(v) for v in sys.argv[1:6])
torch.manual_seed(0)
m = Fno2d(2, 2, 5, loss_name_to_fn("nmse"), 4, 12, 12, C).cuda().eval()
g = torch.Generator().manual_seed(1)
x0, cp, mask = torch.randn(B, 2, H, W, generator=g).cuda(), torch.randn(B, 5, generator=g).cuda(), torch.ones(B, 1, H, W).cuda()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


one = FnoRollout(m)
dt = timed(lambda: one.generate_frames(x0, cp, mask, steps))
ref = one.generate_frames(x0, cp, mask, steps).clone()
print(f"one graph : {dt / steps * 1e6:7.1f} us/step  {B * steps / dt:9.0f} frames/s")
for K in (2, 4):
    per = B // K
    ros = [FnoRollout(m) for _ in range(K)]
    for trial in range(3):
        streams = [torch.cuda.Stream() for _ in range(K)]
        outs = [None] * K

        def run():
            cur = torch.cuda.current_stream()
            for k in range(K):
                streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    outs[k] = ros[k].generate_frames(x0[k * per:(k + 1) * per], cp[k * per:(k + 1) * per], mask[k * per:(k + 1) * per], steps)
            for k in range(K):
                cur.wait_stream(streams[k])
        dt = timed(run)
        got = torch.cat([o for o in outs], dim=1)
        same = torch.equal(got, ref)
        print(f"{K} streams (trial {trial}): {dt / steps * 1e6:7.1f} us/step  {B * steps / dt:9.0f} frames/s  bitwise equal to one graph: {same}")
