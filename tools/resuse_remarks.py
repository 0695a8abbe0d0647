#!/usr/bin/env python
"""Table of hipcc's -Rpass-analysis=kernel-resource-usage remarks: kernel, VGPRs, AGPRs, spilled VGPRs, scratch, LDS bytes, waves per SIMD.

    hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> remarks.txt; python tools/resuse_remarks.py remarks.txt [filter]
"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        return n


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    cur = None
    for ln in txt.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        name = demangle(r["name"])
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        if flt and flt not in name:
            continue
        print(f"{name[:90]:90s} v{r.get('VGPRs','?'):>4s} a{r.get('AGPRs','?'):>4s} spill{r.get('VGPR Spill','?'):>4s} scr{r.get('ScratchSize [bytes/lane]','?'):>5s} "
              f"lds{r.get('LDS Size [bytes/block]','?'):>7s} occ{r.get('Occupancy [waves/SIMD]','?'):>2s}")


if __name__ == "__main__":
    main()
