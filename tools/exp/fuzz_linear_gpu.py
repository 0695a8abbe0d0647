"""GPU box: random-shape sweep of the Linear layer on the three-piece bf16 kernels (k_rowgemm6 / k_rowwgrad6, forced on by gemm_b3 = 2)
against the fp64 layer: catches out-of-bounds accesses (memory faults) and plan / workspace mismatches that fixed test shapes miss."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import kernel_checks as K  # noqa: E402
from tests.backends import TorchBackend  # noqa: E402

be = TorchBackend()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(n):
    Kd = 4 * rnd.choice([4, 5, 9, 16, 25, 32, 40, 50, 64, 96, 128, 129])
    N = 4 * rnd.choice([4, 6, 13, 16, 25, 32, 50, 52, 53, 64, 104, 128, 130])
    M = rnd.choice([1, 7, 31, 32, 33, 255, 256, 257, 1000, 4097, 9000, 20000, 33000])
    if M * (Kd + N) > 3e7:
        M = int(3e7 / (Kd + N))
    act, in_act = rnd.choice([("tanh", "tanh"), ("none", None), ("gelu", "tanh"), ("tanh", None)])
    r = K.check_linear_rowgemm6(be, M, Kd, N, act, in_act, seed=it, force=True)
    worst = max(v for k, v in r.items())
    if not worst < 1e-10:
        bad += 1
        print("BAD", (M, Kd, N, act, in_act), r, flush=True)
print(f"{n} shapes, {bad} bad")
