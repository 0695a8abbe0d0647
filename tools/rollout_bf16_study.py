#!/usr/bin/env python
"""GPU box: bf16 activation storage vs fp32 over a 200-step rollout (BASELINE.json configs[4]; the reference has no reduced-precision
path, so the yardstick is the fp32 rollout of the same network -- itself pinned to the reference by tests/test_gpu_fullsize.py).

    python tools/rollout_bf16_study.py [--steps 200] [--json profiles/r02_rollout_bf16.json]

Network: Fno2d(hidden 32, L 4) on the 66 x 65 grid with the near-identity propagator weights of oracle/synth.py (a stand-in for
a trained one-step model: round-off is carried from step to step instead of being contracted away), band-limited start
frames, tube / dam border mask.  Prints the per-step relative nMSE of the bf16-storage frames against the fp32 frames, its
growth law, and frames/s of both storage types at 64 and 1024 cases."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import synth  # noqa: E402  (measurement tool: the synthetic weights / fields live with the test infrastructure)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--hidden", type=int, default=32)
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    from cfdbench_amd.models.fno.fno2d import Fno2d
    from cfdbench_amd.models.loss import loss_name_to_fn
    from cfdbench_amd.rollout import FnoRollout
    C, L, H, W, p = a.hidden, 4, 66, 65, 5
    res = dict(network=f"Fno2d(hidden {C}, L {L}, modes 12), {H}x{W}, near-identity propagator (eps 0.05, gain 30, decay 0.04)", steps=a.steps)
    params, batch = synth.make_rollout_case(205, 215, 64, C, L, H, W, p, 0.05, 30.0, 0.04)
    m = Fno2d(2, 2, p, loss_name_to_fn("nmse"), L, 12, 12, C).cuda().eval()
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in params.items()})
    b = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
    f32 = FnoRollout(m).generate_frames(b["inputs"], b["case_params"], b["mask"], a.steps).clone()
    b16 = FnoRollout(m, dtype="bf16").generate_frames(b["inputs"], b["case_params"], b["mask"], a.steps).clone()
    num = (b16[1:] - f32[1:]).double().pow(2).mean(dim=(1, 2, 3, 4))
    den = f32[1:].double().pow(2).mean(dim=(1, 2, 3, 4))
    err = (num / den).cpu().numpy()
    # per-case worst step error (the metric averages over cases; a tolerance should hold for each case)
    num_c = (b16[1:] - f32[1:]).double().pow(2).mean(dim=(2, 3, 4))
    den_c = f32[1:].double().pow(2).mean(dim=(2, 3, 4))
    err_c = (num_c / den_c).max(dim=1).values.cpu().numpy()
    k = np.arange(1, a.steps + 1)
    slope = float(np.polyfit(np.log(k[4:]), np.log(err[4:]), 1)[0])
    res["nmse_bf16_vs_fp32_per_step"] = {int(s): float(err[s - 1]) for s in (1, 2, 5, 10, 20, 50, 100, 150, a.steps) if s <= a.steps}
    res["worst_case_nmse_per_step"] = {int(s): float(err_c[s - 1]) for s in (1, 10, 50, 100, a.steps) if s <= a.steps}
    res["growth_exponent"] = round(slope, 3)
    res["curve"] = [float(v) for v in err]
    res["rms_of_frames"] = {int(s): float(den[s - 1].sqrt()) for s in (1, 50, 100, a.steps) if s <= a.steps}
    print("step : nMSE(bf16-storage frames vs fp32 frames)")
    for s, v in res["nmse_bf16_vs_fp32_per_step"].items():
        print(f"{s:5d} : {v:.3e}   (worst case {err_c[s - 1]:.3e})")
    print(f"growth ~ step^{slope:.2f}")
    # throughput
    res["throughput"] = {}
    for B in (64, 1024):
        g = torch.Generator().manual_seed(5)
        x0 = torch.randn(B, 2, H, W, generator=g).cuda()
        cp = torch.randn(B, p, generator=g).cuda()
        mask = torch.ones(B, 1, H, W).cuda()
        mask[:, :, 0, :] = 0
        mask[:, :, -1, :] = 0
        mask[:, :, :, 0] = 0
        for dt in ("f32", "bf16"):
            steps = a.steps if B == 64 else 20
            ro = FnoRollout(m, dtype=dt)
            ro.generate_frames(x0, cp, mask, steps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ro.generate_frames(x0, cp, mask, steps)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / 3
            res["throughput"][f"B{B}_{dt}"] = dict(frames_per_s=round(B * steps / dtm, 1), ms_per_step=round(dtm / steps * 1e3, 4), steps=steps)
            print(f"B={B:5d} {dt:5s}: {B * steps / dtm:10.1f} frames/s  ({dtm / steps * 1e3:.3f} ms per step)")
            del ro
    if a.json:
        Path(a.json).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
