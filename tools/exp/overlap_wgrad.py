#!/usr/bin/env python
"""Dev experiment (GPU box): one FnoBlock's backward (dft -> adjoint mix + spectral weight gradient -> 1x1 weight gradient ->
fused input gradient) with the 1x1 weight gradient on a second stream, concurrent with the two mode-domain launches,
against the single-stream order.  Public C-ABI calls only (every reduction a kernel of its own in both variants)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from cfdbench_amd import _lib  # noqa: E402


def main():
    api = _lib.api()
    dev = torch.device("cuda", 0)
    B, C, H, W, m1, m2 = 256, 20, 64, 64, 12, 12
    plan = _lib.plan(H, W, m1, m2, 0)
    f = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    nl = 4  # distinct buffers per "layer" so that nothing is served from the Infinity Cache by accident
    g, a, gn = [f(B, C, H, W) for _ in range(nl)], [f(B, C, H, W) for _ in range(nl)], [f(B, C, H, W) for _ in range(nl)]
    xh, gh, z = f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2)
    w1, w2 = f(C, C, m1, m2, 2) / (C * C), f(C, C, m1, m2, 2) / (C * C)
    gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
    w0 = f(C, C) / C
    gw0, gb0 = torch.empty_like(w0), torch.empty(C, device=dev)
    ws1 = torch.empty(api.size("cfd_spectral_wgrad_workspace_bytes", plan, B, C, C) + 256, dtype=torch.uint8, device=dev)
    ws2 = torch.empty(api.size("cfd_chan_wgrad_workspace_bytes", B, C, C, H * W) + 256, dtype=torch.uint8, device=dev)
    P = lambda t: t.data_ptr()  # noqa: E731
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def layer(i, two_streams):
        st1 = s1.cuda_stream
        if two_streams:
            ev0 = torch.cuda.Event()
            ev0.record(s1)
            s2.wait_event(ev0)
            api.call("cfd_chan_wgrad", P(g[i]), P(a[i]), P(gw0), P(gb0), P(ws2), B, C, C, H * W, 1, s2.cuda_stream)
        api.call("cfd_spectral_dft", plan, P(g[i]), P(gh), B * C, 0, st1)
        api.call("cfd_spectral_mix_adj_wgrad", plan, P(xh), P(gh), P(w1), P(w2), P(z), P(gw1), P(gw2), P(ws1), B, C, C, st1)
        if two_streams:
            ev1 = torch.cuda.Event()
            ev1.record(s2)
            s1.wait_event(ev1)
        else:
            api.call("cfd_chan_wgrad", P(g[i]), P(a[i]), P(gw0), P(gb0), P(ws2), B, C, C, H * W, 1, st1)
        api.call("cfd_fno_block_bwd_input", plan, P(g[i]), P(z), P(w0), P(a[i]), P(gn[i]), B, C, C, st1)

    for two in (False, True, False, True):
        with torch.cuda.stream(s1):
            for _ in range(2):
                for i in range(nl):
                    layer(i, two)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(s1)
            for _ in range(reps):
                for i in range(nl):
                    layer(i, two)
            e1.record(s1)
            torch.cuda.synchronize()
            print(f"{'two streams' if two else 'one stream '}: {e0.elapsed_time(e1) / (reps * nl) * 1e3:7.2f} us per block backward", flush=True)


if __name__ == "__main__":
    main()
