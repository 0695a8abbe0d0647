#!/usr/bin/env python
"""Dev/measurement tool (GPU box): Auto-DeepONet train step at BASELINE configs[3] shapes
(tube 66x65 lattice, branch_dim 4295, width 100, depth 8/8, batch 512): model(**batch) -> nmse.backward() -> Adam.step().
Prints frames/s, per-step ms and the per-kernel HIP-event breakdown; optional CPU baseline with the same ATen ops."""
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
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu", action="store_true")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of cfdbench_amd.optim.Adam")
    a = ap.parse_args()
    from cfdbench_amd import _lib
    from cfdbench_amd.models.auto_deeponet import AutoDeepONet
    from cfdbench_amd.models.loss import loss_name_to_fn
    H, W, p, B = 66, 65, 5, a.batch
    torch.manual_seed(0)
    m = AutoDeepONet(H * W + p, 2, loss_name_to_fn("nmse"), branch_depth=8, trunk_depth=8, width=100).cuda()
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
    for _ in range(5):
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
    res = dict(workload=f"Auto-DeepONet train step, B={B}, {H}x{W}, width 100, depth 8/8, fp32", frames_per_s=round(B / dt, 1), graph=bool(a.graph),
               ms_per_step=round(dt * 1e3, 3), final_nmse=round(l.item(), 5), hip_kernel_us_per_step=kern)
    if a.cpu:
        import torch.nn as nn

        class RefFfn(nn.Module):
            def __init__(s, dims):
                super().__init__()
                L = []
                for i in range(len(dims) - 2):
                    L += [nn.Linear(dims[i], dims[i + 1]), nn.ReLU()]
                L.append(nn.Linear(dims[-2], dims[-1]))
                s.layers = nn.Sequential(*L)

            def forward(s, v):
                return s.layers(v)

        torch.set_num_threads(16)
        br, tr = RefFfn([H * W + p] + [100] * 8), RefFfn([2] + [100] * 8)
        bias = nn.Parameter(torch.zeros(1))
        o2 = torch.optim.Adam(list(br.parameters()) + list(tr.parameters()) + [bias], lr=1e-3)
        xc, yc, cpc = x.cpu(), y.cpu(), cp.cpu()
        q = torch.tensor([(i, j) for i in range(H) for j in range(W)])

        def cstep():  # auto_deeponet.py:104-143 op for op
            u = xc[:, 0]
            xb = br(torch.cat([u.reshape(B, -1), cpc], 1))
            xt = tr((q.float() - 50) / 100)
            preds = torch.sum(xb.unsqueeze(1) * xt.unsqueeze(0), dim=-1) + bias + u[:, q[:, 0], q[:, 1]]
            lab = yc[:, 0][:, q[:, 0], q[:, 1]]
            loss = torch.nn.functional.mse_loss(preds, lab) / torch.square(lab).mean()
            loss.backward()
            o2.step()
            o2.zero_grad()
        cstep()
        t0 = time.perf_counter()
        for _ in range(3):
            cstep()
        res["cpu_frames_per_s_16_threads"] = round(B / ((time.perf_counter() - t0) / 3), 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
