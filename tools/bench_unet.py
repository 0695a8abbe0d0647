#!/usr/bin/env python
"""Dev/measurement tool (GPU box): U-Net train step at BASELINE configs[2] shapes (cylinder: p=8, dim 12, 64x64,
batch 128 per GPU): model(**batch) -> nmse.backward() -> Adam.step().  Prints frames/s and the HIP-event breakdown."""
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
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of cfdbench_amd.optim.Adam")
    a = ap.parse_args()
    from cfdbench_amd import _lib
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.models.unet import UNet
    H, W, p, B = 64, 64, 8, a.batch
    torch.manual_seed(0)
    m = UNet(2, 2, loss_name_to_fn("nmse"), p, insert_case_params_at="input", dim=12).cuda()
    from cfdbench_amd.optim import Adam as MultiTensorAdam
    opt = MultiTensorAdam(m.parameters(), lr=1e-3) if not a.torch_adam else torch.optim.Adam(m.parameters(), lr=1e-3, capturable=a.graph, fused=True)
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

    if a.graph:
        from cfdbench_amd.graph import GraphedTrainStep
        gs = GraphedTrainStep(m, opt, dict(inputs=x, case_params=cp, label=y, mask=mask))
        eager_step = step

        def step():  # noqa: F811
            return gs(inputs=x, case_params=cp, label=y, mask=mask)["nmse"]
    for _ in range(3):
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
        (eager_step if a.graph else step)()
    buf = ctypes.create_string_buffer(1 << 16)
    api.call("cfd_prof_end", buf, len(buf))
    kern = {ln.split()[0]: round(float(ln.split()[2]) / a.steps * 1e3, 1) for ln in buf.value.decode().splitlines()}
    res = dict(workload=f"U-Net(dim 12, p=8) train step, B={B}, {H}x{W}, fp32", frames_per_s=round(B / dt, 1), graph=bool(a.graph),
               ms_per_step=round(dt * 1e3, 3), final_nmse=round(l.item(), 5), hip_kernel_us_per_step=kern)
    if a.cpu:
        import torch.nn as nn
        import torch.nn.functional as F

        def dc(i, o):
            return nn.Sequential(nn.Conv2d(i, o, 3, padding=1, padding_mode="replicate"), nn.BatchNorm2d(o), nn.ReLU(),
                                 nn.Conv2d(o, o, 3, padding=1, padding_mode="replicate"), nn.BatchNorm2d(o), nn.ReLU())
        d = 12
        torch.set_num_threads(16)
        inc, downs = dc(2 + 1 + p, d), nn.ModuleList([dc(d * 2 ** i, d * 2 ** (i + 1)) for i in range(4)])
        ups = nn.ModuleList([nn.ConvTranspose2d(d * 2 ** (4 - i), d * 2 ** (3 - i), 2, 2) for i in range(4)])
        upc = nn.ModuleList([dc(d * 2 ** (4 - i), d * 2 ** (3 - i)) for i in range(4)])
        outc = nn.Conv2d(d, 2, 1)
        mods = nn.ModuleList([inc, downs, ups, upc, outc])
        o2 = torch.optim.Adam(mods.parameters(), lr=1e-3)
        Bc = min(B, 32)
        xc, yc, cpc, mc = x[:Bc].cpu(), y[:Bc].cpu(), cp[:Bc].cpu(), mask[:Bc].cpu()

        def cstep():  # unet.py:165-222 op for op
            h = torch.cat([xc, mc, cpc[:, :, None, None].expand(-1, -1, H, W)], 1)
            xs = [inc(h)]
            for dn in downs:
                xs.append(dn(F.max_pool2d(xs[-1], 2)))
            cur = xs[-1]
            for i in range(4):
                cur = upc[i](torch.cat([xs[3 - i], ups[i](cur)], 1))
            preds = (outc(cur) + xc) * mc
            lab = yc * mc
            loss = F.mse_loss(preds, lab) / torch.square(lab).mean()
            loss.backward()
            o2.step()
            o2.zero_grad()
        cstep()
        t0 = time.perf_counter()
        for _ in range(2):
            cstep()
        res["cpu_frames_per_s_16_threads_b32"] = round(Bc / ((time.perf_counter() - t0) / 2), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
