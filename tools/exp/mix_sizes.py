import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cfdbench_amd import _lib
api = _lib.api(); dev = torch.device("cuda", 0)
m1 = m2 = 12
plan = _lib.plan(64, 64, m1, m2, 0)
st = torch.cuda.current_stream().cuda_stream
for C in (32, 20):
    for B in (64, 256, 512):
        xh = torch.randn(B, C, 24, 12, 2, device=dev); z = torch.empty_like(xh)
        w1 = torch.randn(C, C, 12, 12, 2, device=dev); w2 = torch.randn(C, C, 12, 12, 2, device=dev)
        for nwv in ("0", "2", "4", "8"):
            os.environ["CFD_MIX_NWV"] = nwv
            f = lambda: api.call("cfd_spectral_mix", plan, xh.data_ptr(), w1.data_ptr(), w2.data_ptr(), z.data_ptr(), B, C, C, 0, st)
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): f()
            e1.record(); torch.cuda.synchronize()
            print(f"C={C} B={B} nwv={nwv}: {e0.elapsed_time(e1) * 10:.2f} us", flush=True)
