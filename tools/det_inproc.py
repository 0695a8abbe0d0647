#!/usr/bin/env python
"""Dev tool (GPU box): the two-process finding of tools/det_kernels.py inside ONE process -- a victim entry point looping on one stream
while an aggressor entry point loops on a second stream of the same process.  VICTIM / AGGR name cases of det_kernels.py's table
(default head_fwd beside head_bwd)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cfdbench_amd import _lib  # noqa: E402

REPS = int(os.environ.get("REPS", "2000"))
api = _lib.api()
dev = torch.device("cuda", 0)
C, H, W = 20, 64, 64
HW = H * W
g = torch.Generator().manual_seed(3)
f = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
P = lambda t: None if t is None else t.data_ptr()  # noqa: E731


def head_io(B):
    a, label, mask = f(B, C, H, W), f(B, 2, H, W), torch.ones(B, 1, H, W, device=dev)
    w = dict(fc1w=f(128, C) / C ** 0.5, fc1b=f(128), fc2w=f(2, 128) / 11.3, fc2b=f(2))
    ws = torch.empty(api.size("cfd_fno_head_workspace_bytes", B, C, 128, 2, HW) + 256, dtype=torch.uint8, device=dev)
    return a, label, mask, w, ws


Bv, Ba = int(os.environ.get("BV", "256")), int(os.environ.get("BA", "37"))
a, label, mask, w, ws = head_io(Bv)
preds, sums = torch.empty(Bv, 2, H, W, device=dev), torch.zeros(4, device=dev)
a2, label2, mask2, w2, ws2 = head_io(Ba)
preds2, out2 = f(Ba, 2, H, W), torch.empty(Ba, C, H, W, device=dev)
coef = torch.tensor([1e-6, 0.0], device=dev)
g1w, g1b, g2w, g2b = (torch.empty_like(w2[k]) for k in ("fc1w", "fc1b", "fc2w", "fc2b"))
s_v, s_a = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def victim():
    api.call("cfd_fno_head_fwd", P(a), P(mask), P(label), P(w["fc1w"]), P(w["fc1b"]), P(w["fc2w"]), P(w["fc2b"]), P(preds), P(sums), P(ws),
             Bv, C, 128, 2, HW, 1, s_v.cuda_stream)


def aggressor():
    api.call("cfd_fno_head_bwd", P(a2), P(mask2), P(label2), P(preds2), None, P(coef), P(w2["fc1w"]), P(w2["fc1b"]), P(w2["fc2w"]), P(out2),
             P(g1w), P(g1b), P(g2w), P(g2b), P(ws2), Ba, C, 128, 2, HW, 1, s_a.cuda_stream)


for mode in ("alone", "beside an aggressor stream of the same process"):
    ref, bad = None, 0
    for rep in range(REPS):
        if mode != "alone":
            for _ in range(4):
                aggressor()
        victim()
        s_v.synchronize()
        cur = preds.clone()
        if ref is None:
            ref = cur
        elif not torch.equal(cur.view(torch.int32), ref.view(torch.int32)):
            bad += 1
    torch.cuda.synchronize()
    print(f"{_lib.lib_path().name}: head_fwd B={Bv} {mode}: {bad} of {REPS} launches differ from the first", flush=True)
