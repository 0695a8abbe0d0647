"""GPU box: random-shape sweep of BatchNorm, pooling / transposed conv / residual epilogue and the fused Linear stacks (the kernels
touched in round 3 besides the convolutions); the emulator twin lives in tests/test_emul_kernels.py."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from tests import kernel_checks as K  # noqa: E402
from tests.backends import TorchBackend  # noqa: E402

be = TorchBackend()
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
R = rnd.choice
bad = 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for it in range(n):
    B, C, H, W = R([1, 2, 3, 5, 9, 64]), R([1, 2, 3, 7, 12, 17, 24, 33, 48, 192]), R([1, 2, 3, 4, 6, 9, 16, 64]), R([1, 2, 3, 4, 5, 8, 9, 16, 64])
    res = dict(K.check_batchnorm(be, B, C, H, W, R([True, False]), R([True, False]), seed=it))
    Ci, Co, h, w = R([1, 2, 3, 7, 12, 24, 50, 192]), R([1, 2, 3, 5, 12, 13, 48, 96]), R([2, 3, 4, 8, 9, 32]), R([2, 3, 4, 7, 16, 32])
    r2 = K.check_pool_convt_resid(be, R([1, 2, 5, 33]), Ci, Co, h, w, seed=it)
    r2 = {k: v for k, v in r2.items() if k not in ("pool", "pool_bwd")}
    dims = [R([1, 2, 3, 7, 16, 33, 100, 128]) for _ in range(R([2, 3, 4, 6, 9]))]
    r3 = K.check_ffn_stack(be, R([1, 5, 16, 17, 33, 70, 129, 4290]), dims, R(["relu", "gelu", "tanh"]), R([True, False]), R([True, False]), seed=it)
    for name, r in (("bn", res), ("convt", r2), ("ffn", r3)):
        worst = max(r.values())
        if not worst < 1e-9:
            bad += 1
            print("BAD", name, it, r, flush=True)
print(f"{n} rounds, {bad} bad")
