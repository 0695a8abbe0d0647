"""Dev helper: per-kernel register / LDS / scratch usage of a HIP source compiled for gfx950."""
import re
import subprocess
import sys

for f in sys.argv[1:]:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-I/root/repo/cfdbench_amd/csrc", "-I/root/repo/include", "-x", "hip", "-c", f, "-o", "/tmp/x.o",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?):\s*(.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            dem = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = {"k": dem.split("(")[0].replace("void ", "")}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for c in rows:
        print(f"{c['k'][:46]:46s} V={c.get('VGPRs'):>4} A={c.get('AGPRs'):>4} S={c.get('TotalSGPRs'):>4} "
              f"scr={c.get('ScratchSize [bytes/lane]'):>5} occ={c.get('Occupancy [waves/SIMD]'):>2} "
              f"spill={c.get('VGPRs Spill'):>4} lds={c.get('LDS Size [bytes/block]')}")
