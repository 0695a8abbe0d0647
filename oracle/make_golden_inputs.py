"""Input generators shared by oracle/make_golden.py and the tests (no reference import here, so the
tests can regenerate fixture inputs on the GPU box where /root/reference does not exist).
TEST INFRASTRUCTURE ONLY."""
import numpy as np


def spectral_case(seed, B, Cin, Cout, H, W, m1, m2):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    gy = rng.standard_normal((B, Cout, H, W)).astype(np.float32)
    sc = 1.0 / (Cin * Cout)
    w1 = (sc * (rng.random((Cin, Cout, m1, m2)) + 1j * rng.random((Cin, Cout, m1, m2)))).astype(np.complex64)
    w2 = (sc * (rng.random((Cin, Cout, m1, m2)) + 1j * rng.random((Cin, Cout, m1, m2)))).astype(np.complex64)
    return x, gy, w1, w2
