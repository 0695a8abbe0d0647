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


# -fno-slp-vectorize: hipcc's SLP vectoriser pairs neighbouring scalar fp32 operations into v_pk_*_f32 with operand selects, among
# them the forms `op_sel:[0,1,...]` (low result from src0.lo and src1.HI) that return a wrong low half in lanes 48-63 while
# certain other kernels are resident on the GPU (tools/exp/pkfma_cotenancy.hip).  The packed math of the
# hot loops is written explicitly (cfd_f2), so nothing is lost; lint_object() below refuses a build that still contains one.
DEVICE_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=fast", "-fno-slp-vectorize"]
_LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")
_BAD_PK = __import__("re").compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[0,1")


def lint_object(obj: Path) -> list:
    """[(kernel symbol, instruction)] for every packed-fp32 instruction of the gfx950 code object inside ``obj`` whose LOW result
    takes src0.lo with src1.HI (see DEVICE_FLAGS).  Empty list = clean.  Objects without device code (the .cpp files) are clean."""
    objdump = _LLVM_BIN / "llvm-objdump"
    if not objdump.exists():
        raise RuntimeError(f"{objdump} not found: cannot lint the device code")
    import tempfile
    hits = []
    with tempfile.TemporaryDirectory() as td:
        tmp = Path(td) / obj.name
        shutil.copyfile(obj, tmp)
        subprocess.run([str(objdump), "--offloading", str(tmp)], capture_output=True, text=True, cwd=td)
        for co in Path(td).glob(obj.name + ".*amdgcn*"):
            r = subprocess.run([str(objdump), "-d", str(co)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"llvm-objdump -d failed on {co.name}: {r.stderr[-500:]}")
            kernel = None
            for ln in r.stdout.splitlines():
                if ln.endswith(">:") and "<" in ln:
                    kernel = ln[ln.index("<") + 1:-2]
                elif _BAD_PK.match(ln):
                    hits.append((kernel, ln.split("//")[0].strip()))
    return hits


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
    flags = [*DEVICE_FLAGS, "-fPIC", f"-I{CSRC}", f"-I{PKG.parent / 'include'}", "-Wno-unused-result", *extra_flags]

    hdr = hashlib.sha1()
    for f in sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "cfdbench_amd.h"]:
        hdr.update(f.name.encode())
        hdr.update(f.read_bytes())
    hdr.update(" ".join(flags).encode())

    def compile_one(src: Path) -> str:
        obj = OUT / (src.name + ".o")
        # per-object stamp (source + every header + flags): an edit of one kernel file recompiles that file only
        ostamp = OUT / (src.name + ".stamp")
        odig = hashlib.sha1(hdr.digest() + src.read_bytes()).hexdigest()
        if obj.exists() and ostamp.exists() and ostamp.read_text() == odig and not force:
            return str(obj)
        cmd = [cc, *flags, "-x", "hip", "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        if src.suffix == ".hip":
            bad = lint_object(obj)
            if bad:
                obj.unlink()
                lines = "\n".join(f"  {k}: {i}" for k, i in bad[:20])
                raise RuntimeError(f"{src.name}: {len(bad)} packed-fp32 instruction(s) of the vulnerable form op_sel:[0,1,..] "
                                   f"(DESIGN.md section 8):\n{lines}")
        ostamp.write_text(odig)
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
