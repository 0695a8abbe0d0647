#!/usr/bin/env python
"""ISA lint (round 3): no packed-fp32 instruction may take its LOW result from (src0.lo, src1.HI), i.e. carry `op_sel:[0,1` .

Measured on MI355X (tools/exp/pkfma_cotenancy.hip, profiles/r03_pk_opsel_*.txt): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with
op_sel = [0,1,x] return a wrong LOW half in lanes 48-63 while waves of certain other kernels (k_head_bwd, k_head_train,
k_dft_fwd64_b3) are resident on the same GPU -- from another process OR another stream of the same process.  Every other
operand-select form tested (op_sel_hi variants, op_sel:[1,0,0], [1,1,0], [0,0,1]) is unaffected.  This was the cause of
k_head_fwd's wrong predictions beside a training job.

    python tools/scan_pk_opsel.py            # compiles every csrc/*.hip to ISA and lists offending kernels; exit status 1 if any
"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "cfdbench_amd" / "csrc"
BAD = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[0,1")


def isa_of(src: Path, extra=()) -> str:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize", f"-I{CSRC}",
                        f"-I{REPO / 'include'}", *extra, "-x", "hip", "-S", "--cuda-device-only", str(src), "-o", "-"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed for {src.name}:\n{r.stderr[-2000:]}")
    return r.stdout


def scan(asm: str):
    """{kernel symbol: [offending instruction lines]}"""
    hits, kernel = {}, None
    for ln in asm.splitlines():
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if BAD.match(ln):
            hits.setdefault(kernel, []).append(ln.strip())
    return hits


def scan_sources(extra=()):
    srcs = sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        asms = list(ex.map(lambda s: isa_of(s, extra), srcs))
    return {s.name: scan(a) for s, a in zip(srcs, asms)}


def main():
    total = 0
    for name, hits in scan_sources().items():
        n = sum(len(v) for v in hits.values())
        total += n
        print(f"{name}: {n} vulnerable packed-fp32 instruction(s) in {len(hits)} kernel(s)")
        for k, v in sorted(hits.items(), key=lambda kv: -len(kv[1]))[:40]:
            print(f"    {len(v):4d}  {k}    e.g. {v[0]}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
