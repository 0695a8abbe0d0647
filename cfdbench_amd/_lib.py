"""Loader of the HIP extension (cfdbench_amd/_C/libcfdbench_amd.so).

There is NO fallback: if the library is missing or lacks a symbol the import of any compute entry point
raises, so a GPU run can never silently route around the hand-written kernels.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path
from typing import Dict, Tuple

from ._capi import CApi, CfdError

# CFDBENCH_AMD_LIB: explicit path of another build of the same library (kernel experiments: tools/build_variant.sh); there is
# still no fallback -- a path that does not exist fails exactly like a missing default build
_LIB_PATH = Path(os.environ.get("CFDBENCH_AMD_LIB") or Path(__file__).resolve().parent / "_C" / "libcfdbench_amd.so")
_lock = threading.Lock()
_api: CApi | None = None
_plans: Dict[Tuple[int, int, int, int, int], int] = {}


def lib_path() -> Path:
    return _LIB_PATH


def api() -> CApi:
    """The bound C ABI.  torch is imported first so that libamdhip64 resolves to the copy torch already
    loaded (one HIP runtime per process: streams and allocations are shared with PyTorch)."""
    global _api
    if _api is None:
        with _lock:
            if _api is None:
                import torch  # noqa: F401  (must precede the dlopen below)
                if not _LIB_PATH.exists():
                    raise CfdError(
                        f"HIP extension not built: {_LIB_PATH} is missing. Run `python -m cfdbench_amd.build` "
                        "(or __graft_entry__.build()). cfdbench_amd has no CPU/PyTorch fallback by design.")
                _api = CApi(ctypes.CDLL(str(_LIB_PATH), mode=ctypes.RTLD_GLOBAL), require_all=True)
                _check_single_hip_runtime()
    return _api


def _check_single_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    real = {os.path.realpath(p) for p in libs}
    if len(real) > 1:
        raise CfdError(f"two HIP runtimes are loaded in this process ({sorted(real)}); the extension must share "
                       "PyTorch's libamdhip64 -- import torch before cfdbench_amd and do not preload /opt/rocm's copy")


def plan(H: int, W: int, m1: int, m2: int, device_index: int) -> int:
    """Process-wide cache of operator tables keyed by (grid, modes, device).  Creating a plan allocates device
    memory, so it must not happen inside stream capture: call this once before capturing."""
    key = (H, W, m1, m2, device_index)
    p = _plans.get(key)
    if p is None:
        with _lock:
            p = _plans.get(key)
            if p is None:
                import torch
                with torch.cuda.device(device_index):
                    p = api().plan_create(H, W, m1, m2)
                _plans[key] = p
    return p
