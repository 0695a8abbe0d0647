#!/usr/bin/env python
"""Dev/measurement tool (GPU box): ResNet (hidden 16, depth 4, 7x7) train step, 64x64, model(**batch) -> backward -> Adam."""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from cfdbench_amd import _lib
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.resnet import ResNet
    H, W, p, B = 64, 64, 5, a.batch
    torch.manual_seed(0)
    m = ResNet(2, 2, p, loss_name_to_fn("nmse"), hidden_chan=16, num_blocks=4, kernel_size=7, padding=3).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 2, H, W, generator=g).cuda()
    y = (x.cpu() + 0.1 * torch.randn(B, 2, H, W, generator=g)).cuda()
    cp = torch.randn(B, p, generator=g).cuda()
    mask = torch.ones(B, 1, H, W).cuda()

    def step():
        out = m(inputs=x, case_params=cp, label=y, mask=mask)
        out["loss"]["nmse"].backward()
        opt.step()
        opt.zero_grad()
        return out["loss"]["nmse"]

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    api = _lib.api()
    api.call("cfd_prof_begin")
    for _ in range(a.steps):
        step()
    buf = ctypes.create_string_buffer(1 << 16)
    api.call("cfd_prof_end", buf, len(buf))
    kern = {ln.split()[0]: round(float(ln.split()[2]) / a.steps * 1e3, 1) for ln in buf.value.decode().splitlines()}
    flops = 3 * 4.37e9 * B  # fwd + bwd (4.37 GFLOP per frame forward)
    print(json.dumps(dict(workload=f"ResNet(hidden 16, depth 4, k7) train step, B={B}, {H}x{W}, fp32", frames_per_s=round(B / dt, 1),
                          ms_per_step=round(dt * 1e3, 3), tflops=round(flops / dt / 1e12, 2), final_nmse=round(l.item(), 5),
                          hip_kernel_us_per_step=kern)))


if __name__ == "__main__":
    main()
