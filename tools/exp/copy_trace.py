#!/usr/bin/env python
"""GPU box: which Python lines of one eager train step of a model leg launch ATen copy / fill / cat kernels (torch.profiler with stacks).
usage: copy_trace.py auto_deeponet|unet|auto_edeeponet"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from cfdbench_amd.optim import Adam  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "auto_deeponet"
dev = torch.device("cuda:0")
captured = {}


def fake_leg(api, name, model, batch, *a, **k):
    captured.update(model=model, batch=batch)
    raise StopIteration


bench.model_train_leg = fake_leg
try:
    bench.MODEL_LEGS[leg][1](None, dev)
except StopIteration:
    pass
model, batch = captured["model"], captured["batch"]
opt = Adam(model.parameters(), lr=1e-3)


def step():
    out = model(**batch)
    out["loss"]["nmse"].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::cat", "aten::zero_", "aten::mul", "aten::add", "aten::add_", "aten::index_add_", "aten::contiguous", "aten::clone") and ev.device_time_total > 0:
        st = [s for s in (ev.stack or []) if "torch/" not in s and "<built-in" not in s][:4] or list(ev.stack or [])[:6]
        st = st + [f"shapes {ev.input_shapes}"] if getattr(ev, "input_shapes", None) else st
        key = (ev.name, tuple(st))
        seen[key] = seen.get(key, 0) + 1
for (name, st), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {name}")
    for s in st:
        print("       ", s[-140:])
