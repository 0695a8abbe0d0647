"""GPU box: phase timestamps of k_conv6 (variant build: tools/build_variant.sh c6diag conv6.hip -DCFD_C6DIAG, CFDBENCH_AMD_LIB
pointing at it): s_memtime of workgroup 0 / wave 0 at the phase boundaries into a __device__ array, read back through
cfd_dbg_c6_read.  usage: c6_diag.py B Ci Co HW ks fwd|dgrad"""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from cfdbench_amd import _lib  # noqa: E402
from cfdbench_amd import functional as F_  # noqa: E402

B, ci, co, hw, ks = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 12, 12, 64, 3))]
x = torch.randn(B, ci, hw, hw, device="cuda")
w = torch.randn(co, ci, ks, ks, device="cuda") * 0.1
b = torch.zeros(co, device="cuda")
mode = sys.argv[6] if len(sys.argv) > 6 else "fwd"
lib = ctypes.CDLL(str(_lib._LIB_PATH))
if mode == "fwd":
    for _ in range(3):
        y = F_.Conv2dReplicateFn.apply(x, w, b)
else:  # input gradient only: the last k_conv6 launch is the transposed pass
    x.requires_grad_(True)
    y = F_.Conv2dReplicateFn.apply(x, w, b)
    torch.cuda.synchronize()
    lib.cfd_dbg_c6_clear()
    y.backward(torch.randn_like(y))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.cfd_dbg_c6_read(buf, 128)
ts = np.array(buf[:], dtype=np.int64).reshape(16, 8)
names = ["top", "barrier1", "commit", "barrier2", "issue", "store", "mfma"]
print(f"{mode} {B=} {ci}->{co} {hw}x{hw} k{ks}: s_memtime ticks (100 MHz: 10 ns = ~21 cycles each) per phase, workgroup 0 wave 0:")
for it in range(16):
    if ts[it, 0] == 0:
        break
    d = np.diff(ts[it, :7])
    nxt = ts[it + 1, 0] - ts[it, 6] if it + 1 < 16 and ts[it + 1, 0] else 0
    print(it, dict(zip(names[1:], d.tolist())), "tail->next", int(nxt))
