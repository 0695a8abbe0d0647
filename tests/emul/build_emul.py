"""Build the product HIP sources against the CPU SIMT emulator (tests/emul/include) into
tests/emul/_build/libcfd_emul.so.  TEST INFRASTRUCTURE ONLY -- never used by the product path."""
from __future__ import annotations

import hashlib
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
CSRC = REPO / "cfdbench_amd" / "csrc"
OUT = HERE / "_build"
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    return sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp"))) + [HERE / "fiber_switch.cpp"]


def build(force: bool = False) -> Path:
    OUT.mkdir(exist_ok=True)
    lib = OUT / "libcfd_emul.so"
    h = hashlib.sha1()
    for f in sources() + sorted(CSRC.glob("*.h")) + sorted((HERE / "include").rglob("*.h")) + [REPO / "include" / "cfdbench_amd.h"]:
        h.update(f.read_bytes())
    stamp = OUT / "stamp"
    if lib.exists() and stamp.exists() and stamp.read_text() == h.hexdigest() and not force:
        return lib
    hdr = hashlib.sha1()
    for f in sorted(CSRC.glob("*.h")) + sorted((HERE / "include").rglob("*.h")) + [REPO / "include" / "cfdbench_amd.h"]:
        hdr.update(f.read_bytes())

    def compile_one(src: Path) -> str:
        obj = OUT / (src.name + ".o")
        ostamp = OUT / (src.name + ".stamp")  # per-object stamp: one edited kernel file recompiles alone
        odig = hashlib.sha1(hdr.digest() + src.read_bytes()).hexdigest()
        if obj.exists() and ostamp.exists() and ostamp.read_text() == odig and not force:
            return str(obj)
        cmd = [CLANG, "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-pthread", "-Wno-unused-value",
               f"-I{HERE / 'include'}", f"-I{CSRC}", f"-I{REPO / 'include'}", "-DCFD_CONV6_GRID=2", "-DCFD_CONVT6_NT4_MIN_WGS=2", "-DCFD_CONV1_NT4_MIN_WGS=2", "-c", str(src), "-o",
               str(obj)]
        subprocess.run(cmd, check=True)
        ostamp.write_text(odig)
        return str(obj)

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:  # one translation unit per core (was serial: 4 minutes)
        objs = list(ex.map(compile_one, sources()))
    subprocess.run([CLANG, "-shared", "-pthread", "-o", str(lib)] + objs, check=True)
    stamp.write_text(h.hexdigest())
    return lib


if __name__ == "__main__":
    print(build(force=True))
