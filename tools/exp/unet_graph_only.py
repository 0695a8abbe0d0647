"""GPU box: U-Net train step as a HIP graph, N replays (for rocprofv3 --kernel-trace --stats: run with two values of N and take
the difference of the per-kernel call counts / times = what ONE replayed step holds)."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import bench
from cfdbench_amd.models.loss import loss_name_to_fn
from cfdbench_amd.models.unet import UNet
from cfdbench_amd.optim import Adam
from cfdbench_amd.graph import GraphedTrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = UNet(2, 2, loss_name_to_fn("nmse"), 8, insert_case_params_at="input", dim=12).to(dev)
b = bench._fields(128, 64, 64, 8, torch.Generator(device="cpu").manual_seed(7), dev)
opt = Adam(m.parameters(), lr=1e-3)
gs = GraphedTrainStep(m, opt, b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    gs(**b)
torch.cuda.synchronize()
print("replay ms/step", (time.perf_counter() - t0) / n * 1e3)
