#!/usr/bin/env python
"""Dev/measurement tool (GPU box): per-layer timing of the replicate-padded conv kernels at the U-Net (dim 12, B=128,
64x64) and ResNet (hidden 16, k=7, B=32) layer shapes: forward, and backward (input + weight gradient)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from cfdbench_amd import functional as F_
    shapes = [("unet", 128, ci, co, hw, 3) for ci, co, hw in
              [(10, 12, 64), (12, 12, 64), (12, 24, 32), (24, 24, 32), (24, 48, 16), (48, 48, 16), (48, 96, 8), (96, 96, 8),
               (96, 192, 4), (192, 192, 4), (192, 96, 8), (96, 48, 16), (48, 24, 32), (24, 12, 64)]]
    # ResidualBlock(in, out, hidden = 64): conv1 in -> 64, conv2 64 -> out (src/models/resnet.py:35-55)
    shapes += [("resnet", 32, 8, 64, 64, 7), ("resnet", 32, 64, 16, 64, 7), ("resnet", 32, 16, 64, 64, 7), ("resnet", 32, 64, 2, 64, 7)]
    tot = [0.0, 0.0]
    if len(sys.argv) > 1:  # e.g. "1,5": only these rows (for counter collection)
        shapes = [shapes[int(i)] for i in sys.argv[1].split(",")]
    for fam, B, ci, co, hw, ks in shapes:
        x = torch.randn(B, ci, hw, hw, device="cuda", requires_grad=True)
        w = (torch.randn(co, ci, ks, ks, device="cuda") * 0.1).requires_grad_(True)
        b = torch.zeros(co, device="cuda", requires_grad=True)
        y = F_.Conv2dReplicateFn.apply(x, w, b)
        g = torch.randn_like(y)
        tf = timed(lambda: F_.Conv2dReplicateFn.apply(x, w, b))

        def bwd():
            yy = F_.Conv2dReplicateFn.apply(x, w, b)
            yy.backward(g)
        tb = timed(bwd) - tf
        xd = x.detach()  # no input gradient: weight + bias gradient only

        def bwd_w():
            yy = F_.Conv2dReplicateFn.apply(xd, w, b)
            yy.backward(g)
        tw = timed(bwd_w) - tf
        gf = 2.0 * B * hw * hw * ci * co * ks * ks / 1e9
        byts = 4.0 * B * hw * hw * (ci + co) / 1e6
        print(f"{fam:6s} B={B:4d} {ci:4d}->{co:4d} {hw:3d}x{hw:<3d} k{ks}  fwd {tf:8.1f} us ({gf / tf * 1e3:6.1f} TF, {byts / tf:6.2f} TB/s)"
              f"   bwd {tb:8.1f} us ({2 * gf / tb * 1e3:6.1f} TF)  of which wgrad+bias {tw:7.1f} us")
        if fam == "unet":
            tot[0] += tf
            tot[1] += tb
    print(f"unet conv total: fwd {tot[0]:.0f} us, bwd {tot[1]:.0f} us")


if __name__ == "__main__":
    main()
