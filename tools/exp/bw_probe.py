"""GPU box: what a plain streaming kernel sustains (torch ops): read-only (sum), write-only (fill), copy, at 84 MB (one
activation of the bench step) and 1 GiB."""
import torch
dev = torch.device("cuda", 0)
for mb in (84, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    def t(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    for name, fn, bytes_ in (("read (sum)", lambda: x.sum(), 4 * n), ("write (fill)", lambda: y.fill_(1.0), 4 * n),
                             ("copy", lambda: y.copy_(x), 8 * n), ("read abs-max", lambda: x.abs().max(), 0)):
        if bytes_ == 0: continue
        s = t(fn)
        print(f"{mb:5d} MB {name:14s} {s * 1e6:8.1f} us  {bytes_ / s / 1e12:6.2f} TB/s", flush=True)
