"""Build cfdbench_amd/_C/libcfdbench_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

In-tree on purpose: the built .so is git-ignored but travels to the GPU box with the repo snapshot.
    python -m cfdbench_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "_C"
LIB = OUT / "libcfdbench_amd.so"
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: cannot build the cfdbench_amd HIP extension")


def sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")))


def _digest() -> str:
    h = hashlib.sha1()
    for f in sources() + sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "cfdbench_amd.h"]:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> Path:
    OUT.mkdir(exist_ok=True)
    stamp = OUT / "stamp"
    dig = _digest()
    if LIB.exists() and stamp.exists() and stamp.read_text() == dig and not force:
        return LIB
    cc = hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", f"-I{CSRC}",
             f"-I{PKG.parent / 'include'}", "-Wno-unused-result", *extra_flags]

    def compile_one(src: Path) -> str:
        obj = OUT / (src.name + ".o")
        cmd = [cc, *flags, "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return str(obj)

    with ThreadPoolExecutor(max_workers=min(8, len(sources()))) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
