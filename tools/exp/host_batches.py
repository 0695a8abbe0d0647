"""How should a caller that owns HOST batches feed the fused train step?  (GPU box; `python tools/exp/host_batches.py [steps]`)

Times the headline step (Auto-FNO, B = 256, 64x64) fed from pinned host batches -- 21 MB per step over PCIe -- in these forms:
  base            no upload (the bench line)
  serial_memcpy   hipMemcpyAsync (torch copy_) on the step's stream, then the step
  ov2_memcpy      two device buffer sets, the next batch's hipMemcpyAsync on a copy stream behind events
  ov3_memcpy      three sets (the copy's event dependency is two steps old when it is enqueued)
  serial_kernel   an upload KERNEL on the step's stream: cfd_scale_copy_multi reading the pinned host pointers (mapped into the device's
                  address space), scale 1
  ov2_kernel      the upload kernel on a second stream, two buffer sets
Prints one JSON object."""
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from cfdbench_amd import _lib  # noqa: E402
from cfdbench_amd.engine import FnoTrainEngine  # noqa: E402
from cfdbench_amd.models.fno.fno2d import Fno2d  # noqa: E402
from cfdbench_amd.models.loss import loss_name_to_fn  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda", 0)
    api = _lib.api()
    B, C, L, H, W, p = 256, 20, 4, 64, 64, 5
    torch.manual_seed(0)
    model = Fno2d(2, 2, p, loss_name_to_fn("nmse"), L, 12, 12, C).to(dev)
    eng = FnoTrainEngine(model, lr=1e-3, loss_name="nmse")
    g = torch.Generator(device="cpu").manual_seed(1234)
    inputs = torch.randn(B, 2, H, W, generator=g)
    label = inputs + 0.1 * torch.randn(B, 2, H, W, generator=g)
    cp = torch.randn(B, p, generator=g)
    mask = torch.ones(B, 1, H, W)
    host = (inputs, label, cp, mask)
    nset = 3
    hb = [tuple(t.clone().pin_memory() for t in host) for _ in range(nset)]
    db = [tuple(torch.empty_like(t, device=dev) for t in host) for _ in range(nset)]
    nbytes = sum(t.numel() * 4 for t in host)
    col = lambda vals, ty: (ty * len(vals))(*vals)  # noqa: E731
    tables = [(len(host), col([h.data_ptr() for h in hb[k]], ctypes.c_void_p), col([d.data_ptr() for d in db[k]], ctypes.c_void_p),
               col([h.numel() for h in hb[k]], ctypes.c_size_t)) for k in range(nset)]

    def up_memcpy(k):
        for d, h in zip(db[k], hb[k]):
            d.copy_(h, non_blocking=True)

    def up_kernel(k):
        n, sp, dp, nn = tables[k]
        api.call("cfd_scale_copy_multi", n, sp, dp, nn, 1.0, torch.cuda.current_stream().cuda_stream)

    def timed(fn, warm=8):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    out = {"h2d_MB_per_step": round(nbytes / 1e6, 2), "steps": steps}
    for d, h in zip(db[0], hb[0]):
        d.copy_(h)
    out["base_ms"] = round(min(timed(lambda: eng.train_step(*db[0])) for _ in range(2)), 4)
    out["upload_only_memcpy_ms"] = round(timed(lambda: up_memcpy(0)), 4)
    print(json.dumps(out), flush=True)  # (partial lines: the kernel forms read host memory from the device -- should they fault, the rest is kept)

    turn = [0]

    def run_overlap(up, sets, side):
        cur = torch.cuda.current_stream()
        ready, free = [torch.cuda.Event() for _ in range(sets)], [torch.cuda.Event() for _ in range(sets)]
        for e in free:
            e.record(cur)

        def upload(k):
            side.wait_event(free[k])
            with torch.cuda.stream(side):
                up(k)
            ready[k].record(side)
        turn[0] = 0
        torch.cuda.synchronize()
        upload(0)

        def overlapped():
            k = turn[0] % sets
            turn[0] += 1
            upload((k + 1) % sets)
            cur.wait_event(ready[k])
            eng.train_step(*db[k])
            free[k].record(cur)
        dt = round(min(timed(overlapped) for _ in range(2)), 4)
        torch.cuda.synchronize()
        return dt

    for name, up in (("memcpy", up_memcpy), ("kernel", up_kernel)):
        if name == "kernel":
            print(json.dumps(out), flush=True)
            for d in db[0]:
                d.zero_()
            torch.cuda.synchronize()
            up_kernel(0)
            torch.cuda.synchronize()
            out["upload_kernel_exact"] = all(bool(torch.equal(d.cpu(), h)) for d, h in zip(db[0], hb[0]))
            out["upload_only_kernel_ms"] = round(timed(lambda: up_kernel(0)), 4)

        def serial():
            k = turn[0] % 2
            turn[0] += 1
            up(k)
            eng.train_step(*db[k])
        out[f"serial_{name}_ms"] = round(min(timed(serial) for _ in range(2)), 4)
        # streams map onto a few hardware queues: a copy stream that shares the step's queue cannot overlap it.  Six fresh streams of
        # the default priority one after the other, then a high-priority one (its own queue class)
        out[f"ov2_{name}_ms_by_stream"] = [run_overlap(up, 2, torch.cuda.Stream()) for _ in range(40 if name == "memcpy" else 3)]
        out[f"ov2_{name}_high_priority_ms"] = run_overlap(up, 2, torch.cuda.Stream(priority=-1))
        if name == "memcpy":
            out["ov3_memcpy_high_priority_ms"] = run_overlap(up, 3, torch.cuda.Stream(priority=-1))
    out["priority_range"] = list(torch.cuda.Stream.priority_range())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
