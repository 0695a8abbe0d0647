"""GPU box: one bench leg's train step as a HIP graph, N replays and nothing else after the capture (for rocprofv3 --kernel-trace
--stats: run with two values of N and difference the per-kernel call counts / times = what ONE replayed step holds).
usage: graph_only.py LEG N"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench
from cfdbench_amd.optim import Adam
from cfdbench_amd.graph import GraphedTrainStep

leg, n = sys.argv[1], int(sys.argv[2])
grabbed = {}


def grab(api, name, model, batch, *a, **k):
    grabbed.update(model=model, batch=batch)
    raise KeyboardInterrupt  # leave the leg before it times anything


bench.model_train_leg = grab
try:
    bench.MODEL_LEGS[leg][1](None, torch.device("cuda:0"))
except KeyboardInterrupt:
    pass
m, b = grabbed["model"], grabbed["batch"]
gs = GraphedTrainStep(m, Adam(m.parameters(), lr=1e-3), b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    gs(**b)
torch.cuda.synchronize()
print("replay ms/step", (time.perf_counter() - t0) / n * 1e3)
