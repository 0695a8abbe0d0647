"""GPU box: torch.optim.Adam foreach (default) vs fused=True on the U-Net / Auto-DeepONet parameter sets: time per step, agreement."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from cfdbench_amd.models.loss import loss_name_to_fn  # noqa: E402
from cfdbench_amd.models.unet import UNet  # noqa: E402

torch.manual_seed(0)
for name in ("unet",):
    outs = {}
    for fused in (False, True):
        torch.manual_seed(0)
        m = UNet(2, 2, loss_name_to_fn("nmse"), 8, insert_case_params_at="input", bilinear=False, dim=12).cuda()
        g = torch.Generator(device="cuda").manual_seed(1)
        for p in m.parameters():
            p.grad = torch.randn(p.shape, device="cuda", generator=g) * 1e-3
        opt = torch.optim.Adam(m.parameters(), lr=torch.tensor(1e-3, device="cuda"), capturable=True, fused=fused)
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            opt.step()
        e0.record()
        for _ in range(50):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        outs[fused] = torch.cat([p.detach().flatten() for p in m.parameters()])
        print(name, "fused" if fused else "foreach", f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us per step (graph replay)")
    d = (outs[True] - outs[False]).abs().max().item()
    print("max |fused - foreach| after 54 steps:", d)
