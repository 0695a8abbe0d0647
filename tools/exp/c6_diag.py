"""GPU box: phase timestamps of k_conv6 / k_conv6_wgrad (variant build with -DCFD_C6DIAG, CFDBENCH_AMD_LIB pointing at it).
The C6_TS(slot) stamps were taken out of csrc/conv6.hip after the measurements (profiles/r03w_conv6_phase_timestamps.txt); they
live in the history: `git show 4bb9d6d:cfdbench_amd/csrc/conv6.hip` has them (s_memtime of workgroup 0 / wave 0 into a __device__
array read back through cfd_dbg_c6_read)."""
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
if mode == "fwd":
    for _ in range(3):
        y = F_.Conv2dReplicateFn.apply(x, w, b)
else:  # weight gradient only (the input needs no gradient): the last conv6 kernel to run is k_conv6_wgrad
    w.requires_grad_(True)
    y = F_.Conv2dReplicateFn.apply(x, w, b)
    y.backward(torch.randn_like(y))
torch.cuda.synchronize()
lib = ctypes.CDLL(str(_lib._LIB_PATH))
buf = (ctypes.c_ulonglong * 128)()
lib.cfd_dbg_c6_read(buf, 128)
ts = np.array(buf[:], dtype=np.int64).reshape(16, 8)
names = ["top", "barrier1", "commit", "barrier2", "issue", "store", "mfma"] if mode == "fwd" else ["top", "barrier1", "commit", "barrier2", "issue", "mfma", "-"]
print("s_memtime ticks (100 MHz: 10 ns each) per phase, workgroup 0 wave 0:")
for it in range(16):
    if ts[it, 0] == 0:
        break
    last = 6 if mode == "fwd" else 5
    d = np.diff(ts[it, :last + 1])
    nxt = ts[it + 1, 0] - ts[it, last] if it + 1 < 16 and ts[it + 1, 0] else 0
    print(it, dict(zip(names[1:], d.tolist())), "tail->next", int(nxt))
