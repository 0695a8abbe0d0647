"""MI355X: every FNO-path entry point returns bit-identical results while OTHER kernels are resident on the same GPU -- from a second
stream of this process, and from a foreign process (an evaluation job beside a trainer, two ranks sharing a device).

Round 2 shipped a k_head_fwd that returned wrong predictions beside a foreign k_head_bwd.  Root cause (round 3,
DESIGN.md section 8, tools/exp/pkfma_cotenancy.hip): packed-fp32 instructions whose LOW result takes src0.lo with src1.HI
(`v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1,..]`) return a wrong low half in lanes 48-63 while waves of certain
other kernels (k_head_bwd, k_head_train, k_dft_fwd64_b3) are resident -- in another process OR on another stream.  The library no
longer contains such an instruction (cfdbench_amd/build.py:lint_object, tests/test_cpu_host.py); these tests are the behavioural side.
"""
import os
import subprocess
import sys
import time
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "tools"))

AGGRESSORS = ("head_bwd", "head_train", "dft")  # the kernels that triggered the wrong low halves in tools/exp/pkfma_cotenancy


def _stable(torch, fn, outs, reps, between=None):
    """Number of launches (of ``reps``) whose outputs differ bitwise from the first launch."""
    ref, bad = None, 0
    for _ in range(reps):
        for o in outs:
            o.fill_(float("nan"))
        if between is not None:
            between()
        fn()
        torch.cuda.synchronize()
        cur = [o.clone() for o in outs]
        if ref is None:
            ref = cur
        elif not all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(cur, ref)):
            bad += 1
    return bad


@pytest.mark.parametrize("B", [4, 64])
def test_entry_points_are_bitwise_stable_beside_a_second_stream(B):
    """Victim entry point on the current stream, the aggressors looping on a second stream of the SAME process (with the round-2
    library: head_fwd differs in ~100 % of the launches at B = 256, profiles/r03i_inproc_two_streams.txt)."""
    import torch
    import det_kernels as D
    side = torch.cuda.Stream()
    victims = D.build_cases(B)
    aggr = D.build_cases(37, stream=side.cuda_stream)
    torch.cuda.synchronize()

    def between():
        for name in AGGRESSORS:
            aggr[name][0]()

    failures = {}
    for name, (fn, outs) in victims.items():
        bad = _stable(torch, fn, outs, 60, between)
        if bad:
            failures[name] = bad
    torch.cuda.synchronize()
    assert not failures, f"entry points that are not bitwise reproducible beside a second stream (launches of 60): {failures}"


def test_entry_points_are_bitwise_stable_beside_a_foreign_process(tmp_path):
    """The two-process form: a second PROCESS loops the aggressor entry points on the same GPU while this one repeats every entry
    point (the configuration of tests/test_gpu_dp.py -- two ranks on one device -- and of an evaluation beside a training job)."""
    import torch
    import det_kernels as D
    env = dict(os.environ, ONLY=",".join(AGGRESSORS), REPS="1000000", BATCHES="37")
    log = open(tmp_path / "aggressor.log", "w")
    aggressor = subprocess.Popen([sys.executable, str(REPO / "tools" / "det_kernels.py")], env=env, stdout=log, stderr=subprocess.STDOUT)
    try:
        time.sleep(12)  # the foreign process imports torch, builds its inputs and starts looping
        assert aggressor.poll() is None, "the aggressor process exited early: " + (tmp_path / "aggressor.log").read_text()[-2000:]
        failures = {}
        for B in (4, 37, 256):
            for name, (fn, outs) in D.build_cases(B).items():
                bad = _stable(torch, fn, outs, 150)
                if bad:
                    failures[(B, name)] = bad
        assert aggressor.poll() is None, "the aggressor process died during the test"
    finally:
        aggressor.kill()
        aggressor.wait(timeout=60)
        log.close()
    assert not failures, f"entry points that are not bitwise reproducible beside a foreign process (launches of 150): {failures}"
