#!/usr/bin/env python
"""Compile every kernel file for gfx950 with --save-temps and list the kernels in which a global load is directly followed by
`s_waitcnt vmcnt(0)` (within three instructions, no other load in between) -- the signature of `inside ? load : 0` and of clamped loads
whose only use sits under an `if`: one exposed memory round trip per load, and a drain of every prefetch in flight.
usage: python tools/exp/isa_scan.py [min_count]        (no GPU needed)"""
import glob
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
CSRC = REPO / "cfdbench_amd" / "csrc"


def main():
    floor = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with tempfile.TemporaryDirectory() as td:
        def comp(src):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize",
                            f"-I{CSRC}", f"-I{REPO / 'include'}", "-c", str(src), "-o", f"{td}/{src.stem}.o", "--save-temps"], cwd=td,
                           capture_output=True)
        with ThreadPoolExecutor(8) as ex:
            list(ex.map(comp, sorted(CSRC.glob("*.hip"))))
        for f in sorted(glob.glob(f"{td}/*-hip-amdgcn*.s")):
            lines = open(f).read().split("\n")
            kern, stats = None, {}
            for i, ln in enumerate(lines):
                m = re.match(r"^(_Z\w+):", ln)
                if m:
                    kern = m.group(1)
                    stats[kern] = [0, 0]
                if kern is None:
                    continue
                if "global_load" in ln or "buffer_load" in ln:
                    stats[kern][0] += 1
                    nxt = " ".join(lines[i + 1:i + 4])
                    if "s_waitcnt vmcnt(0)" in nxt and "global_load" not in nxt:
                        stats[kern][1] += 1
            for k, (nl, nw) in stats.items():
                if nw >= floor:
                    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120]
                    print(f"{Path(f).name.split('-')[0]:10s} loads={nl:4d} load+wait0={nw:4d}  {name}")


if __name__ == "__main__":
    main()
