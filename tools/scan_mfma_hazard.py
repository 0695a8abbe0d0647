#!/usr/bin/env python
"""Dev tool: scan the gfx950 ISA of every kernel for the pattern that made k_head_fwd nondeterministic under GPU time-slicing:
an MFMA whose SrcC is the destination of one of the `window` preceding MFMAs (so it cannot start before that one retires),
whose own destination differs from SrcC, followed closely by a load (ds_read / global_load / buffer_load / scratch_load) that
overwrites a SrcC register.  hipcc protects that write-after-read with a fixed number of wait states counted from ISSUE, but
the dependent MFMA reads SrcC only when its predecessor is done."""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CSRC = REPO / "cfdbench_amd" / "csrc"


def regs(tok):
    m = re.match(r"[va]\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"[va](\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm, window=2, reach=16):
    hits = []
    kernel = None
    lines = asm.splitlines()
    mfmas = []  # (line index, dst, srcC)
    for i, ln in enumerate(lines):
        if ln and not ln.startswith(("\t", " ", ".", ";")) and ln.endswith(":") or (":" in ln and ln.startswith("_Z")):
            kernel = ln.split(":")[0]
            mfmas = []
        t = ln.strip()
        if t.startswith("v_mfma"):
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            dst, srcc = regs(ops[0]), regs(ops[3].split()[0])
            dep = any(srcc & d for _, d, _ in mfmas[-window:])
            if dep and dst != srcc:
                for k in range(i + 1, min(i + 1 + reach, len(lines))):
                    u = lines[k].strip()
                    if u.startswith(("ds_read", "ds_load", "global_load", "buffer_load", "scratch_load", "flat_load")):
                        wr = regs(u.split(None, 1)[1].split(",")[0].strip())
                        if wr & srcc:
                            hits.append((kernel, i + 1, t, k + 1, u))
                            break
                    if u.startswith("v_mfma") and k > i + 6:
                        break
            mfmas.append((i, dst, srcc))
    return hits


def main():
    total = 0
    for src in sorted(CSRC.glob("*.hip")):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", f"-I{CSRC}",
                            f"-I{REPO / 'include'}", "-x", "hip", "-S", "--cuda-device-only", str(src), "-o", "-"],
                           capture_output=True, text=True)
        hits = scan(r.stdout)
        total += len(hits)
        print(f"{src.name}: {len(hits)} suspicious site(s)")
        for k, a, t, b, u in hits[:40]:
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
            print(f"   {dem[:60]}: line {a}: {t}   ->   line {b}: {u}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
