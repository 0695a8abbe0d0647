"""GPU box: phase stamps of k_ffn_stack_fwd (variant build: tools/build_variant.sh fsdiag ffn.hip -DCFD_FSDIAG, CFDBENCH_AMD_LIB pointing at
it): the last workgroup's wave 0, per layer: top | barrier1 | commit | barrier2 | fetch issue | gemm (+ epilogue up to the next top)."""
import ctypes, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from cfdbench_amd import _lib
from cfdbench_amd import functional as F_

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4290
L, wdt = 8, 100
dims = [2] + [wdt] * L
torch.manual_seed(0)
ws = [torch.randn(dims[l + 1], dims[l], device="cuda") / dims[l] ** 0.5 for l in range(L)]
bs = [torch.zeros(dims[l + 1], device="cuda") for l in range(L)]
x = torch.randn(R, 2, device="cuda")
xb = torch.randn(512, wdt, device="cuda")
for w in ws:
    w.requires_grad_(True)
for _ in range(3):
    yb, yt = F_.ffn_stacks([(xb, ws[1:], bs[1:], "relu", False), (x, ws, bs, "relu", False)])
    (yb.sum() + yt.sum()).backward()
torch.cuda.synchronize()
lib = ctypes.CDLL(str(_lib._LIB_PATH))
buf = (ctypes.c_ulonglong * 128)()
lib.cfd_dbg_fs_read(buf, 128)
ts = np.array(buf[:], dtype=np.int64).reshape(16, 8)
names = ["barrier1", "commit", "barrier2", "fetch_issue", "gemm"]
for l in range(L):
    d = np.diff(ts[l, :6])
    nxt = ts[l + 1, 0] - ts[l, 5] if l + 1 < L else 0
    print(l, dict(zip(names, d.tolist())), "epilogue->next", int(nxt))
print("total cycles", int(ts[L - 1, 5] - ts[0, 0]))
names2 = ["barrier1", "dZ pass", "commit", "barrier2+fetch_issue", "gemm_t"]
print("backward chain (layers L-1 .. 0):")
for l in range(L):
    d = np.diff(ts[8 + l, :6])
    nxt = ts[8 + l + 1, 0] - ts[8 + l, 5] if l + 1 < L else 0
    print(l, dict(zip(names2, d.tolist())), "epilogue->next", int(nxt))
print("total cycles", int(ts[8 + L - 1, 5] - ts[8, 0]))
