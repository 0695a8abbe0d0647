"""GPU box: random-shape sweep of the conv entry points (the emulator twin lives in tests/test_emul_kernels.py): catches what only
the hardware shows -- out-of-bounds accesses (memory faults), alignment assumptions."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import kernel_checks as K  # noqa: E402
from tests.backends import TorchBackend  # noqa: E402

be = TorchBackend()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(n):
    ks = rnd.choice([3, 3, 5, 7])
    B = rnd.choice([1, 2, 3, 5, 8, 9, 17, 33, 128])
    Ci, Co = rnd.choice([1, 2, 5, 8, 9, 12, 16, 17, 24, 33, 48, 96, 192]), rnd.choice([1, 2, 7, 12, 16, 17, 31, 40, 64, 96, 192])
    H, W = rnd.choice([1, 2, 3, 4, 5, 8, 13, 16, 32, 33, 64]), rnd.choice([1, 2, 3, 4, 7, 8, 9, 16, 17, 32, 40, 64, 65])
    if B * (Ci + Co) * H * W > 4e7:
        B = max(1, int(4e7 / ((Ci + Co) * H * W)))
    grid = rnd.choice([-1, -1, 1, 2, 3, 8, 64])
    with K.tuned(be, conv6_grid=grid):
        r = K.check_conv2d(be, B, Ci, Co, H, W, ks, seed=it)
        s = K.check_conv_bn_stats(be, B, Ci, Co, H, W, ks, seed=it)
    worst = max(list(r.values()) + (list(s.values()) if s else []))
    if not worst < 1e-10:
        bad += 1
        print("BAD", (B, Ci, Co, H, W, ks, grid), r, s, flush=True)
print(f"{n} shapes, {bad} bad")
