"""Array backends for the shared kernel checks: the same C-ABI calls run either on the CPU SIMT emulator
(NumPy buffers) or on the GPU (torch CUDA buffers through the product library)."""
from __future__ import annotations

import ctypes

import numpy as np


class NumpyBackend:
    """libcfd_emul.so: the product kernel sources compiled against tests/emul (host threads)."""
    name = "emul"

    def __init__(self):
        from cfdbench_amd._capi import CApi
        from tests.emul.build_emul import build
        self.api = CApi(ctypes.CDLL(str(build())))
        self.stream = None

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype)

    def ptr(self, a):
        return None if a is None else a.ctypes.data

    def host(self, a):
        return np.array(a)

    def bytes(self, n):
        return np.zeros(max(int(n), 1), np.uint8)

    def sync(self):
        pass


class TorchBackend:
    """The shipped library on a real GPU."""
    name = "gpu"

    def __init__(self):
        import torch
        from cfdbench_amd import _lib
        self.torch = torch
        self.api = _lib.api()
        self.stream = torch.cuda.current_stream().cuda_stream

    def dev(self, a):
        a = np.ascontiguousarray(a)
        return self.torch.from_numpy(a).cuda()

    def zeros(self, shape, dtype=np.float32):
        td = {np.float32: self.torch.float32, np.complex64: self.torch.complex64, np.uint8: self.torch.uint8}[dtype]
        return self.torch.zeros(shape, dtype=td, device="cuda")

    def ptr(self, a):
        return None if a is None else a.data_ptr()

    def host(self, a):
        return a.detach().cpu().numpy()

    def bytes(self, n):
        return self.torch.zeros(max(int(n), 1), dtype=self.torch.uint8, device="cuda")

    def sync(self):
        self.torch.cuda.synchronize()
