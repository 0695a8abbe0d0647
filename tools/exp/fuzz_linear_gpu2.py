"""GPU box: the default routing (no knob) of tall Linear layers at random widths 160 .. 700 and row counts 4096 .. 60000 against the fp64 layer."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import kernel_checks as K  # noqa: E402
from tests.backends import TorchBackend  # noqa: E402

be = TorchBackend()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for it in range(n):
    Kd, N = 4 * rnd.randint(40, 175), 4 * rnd.randint(40, 175)
    M = rnd.choice([4096, 4097, 5000, 12345, 30000, 60000])
    r = K.check_linear_rowgemm6(be, M, Kd, N, "tanh", rnd.choice(["tanh", None]), seed=it, force=False)
    worst = max(v for k, v in r.items())
    print((M, Kd, N), f"{worst:.2e}", flush=True)
    if not worst < 1e-10:
        bad += 1
        print("BAD", r, flush=True)
print(f"{n} shapes, {bad} bad")
