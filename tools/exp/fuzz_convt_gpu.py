#!/usr/bin/env python
"""GPU box: seeded random-shape sweep of the ConvTranspose2d(2, 2) kernels (convt6.hip and the fp32 fallbacks) against the oracle."""
import random
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests import kernel_checks as K  # noqa: E402
from tests.backends import TorchBackend  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    be = TorchBackend()
    rnd = random.Random(99)
    R = rnd.choice
    bad = 0
    for i in range(n):
        B, Ci, Co = R([1, 2, 3, 8, 31, 64]), R([1, 2, 7, 16, 17, 24, 48, 49, 96, 192]), R([1, 2, 3, 4, 5, 12, 13, 24, 48, 96])
        H, W = R([1, 2, 3, 4, 8, 9, 16, 32]), R([1, 3, 4, 8, 8, 16, 24, 32])
        res = K.check_convt(be, B, Ci, Co, H, W, seed=1000 + i)
        b = {k: v for k, v in res.items() if not (v < 1e-10)}
        if b:
            bad += 1
            print("BAD", (B, Ci, Co, H, W), b, flush=True)
    print(f"convT fuzz: {n} shapes, {bad} bad")


if __name__ == "__main__":
    main()
