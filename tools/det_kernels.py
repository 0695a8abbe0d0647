#!/usr/bin/env python
"""Dev tool (GPU box): every FNO-path C-ABI entry point repeated on fixed inputs, outputs compared bit for bit with the first
run.  Start two instances at once to check the kernels under GPU time-slicing (REPS=300 python tools/det_kernels.py &)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from cfdbench_amd import _lib  # noqa: E402

REPS = int(os.environ.get("REPS", "100"))

def build_cases(B, C=20, dev=None, stream=None):
    """{name: (launch function, [output tensors])} for every FNO-path C-ABI entry point on fixed seeded inputs (batch B, C channels,
    64 x 64, modes 12).  ``stream``: raw HIP stream the launches go to (default: torch's current stream)."""
    api = _lib.api()
    dev = dev or torch.device("cuda", 0)
    H, W, m1, m2 = 64, 64, 12, 12
    HW = H * W
    plan = _lib.plan(H, W, m1, m2, 0)
    st = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    f = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    a, a2, gg, out = f(B, C, H, W), f(B, C, H, W), f(B, C, H, W), f(B, C, H, W)
    xh, gh, z = f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2), f(B, C, 2 * m1, m2, 2)
    w1, w2 = f(C, C, m1, m2, 2) / (C * C), f(C, C, m1, m2, 2) / (C * C)
    gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
    w0, b0 = f(C, C) / C, f(C)
    gw0, gb0 = torch.empty_like(w0), torch.empty_like(b0)
    inputs, label, mask, cp = f(B, 2, H, W), f(B, 2, H, W), torch.ones(B, 1, H, W, device=dev), f(B, 5)
    fc0w, fc0b = f(C, 10), f(C)
    gfc0w, gfc0b = torch.empty_like(fc0w), torch.empty_like(fc0b)
    fc1w, fc1b, fc2w, fc2b = f(128, C) / C ** 0.5, f(128), f(2, 128) / 11.3, f(2)
    g1w, g1b, g2w, g2b = (torch.empty_like(t) for t in (fc1w, fc1b, fc2w, fc2b))
    preds, sums, coef = f(B, 2, H, W), torch.zeros(4, device=dev), torch.tensor([1e-6, 0.0], device=dev)
    xo, zo = torch.empty_like(xh), torch.empty_like(z)
    ws_n = max(api.size("cfd_spectral_wgrad_workspace_bytes", plan, B, C, C), api.size("cfd_chan_wgrad_workspace_bytes", B, C, C, HW),
               api.size("cfd_fno_head_workspace_bytes", B, C, 128, 2, HW), api.size("cfd_fno_stem_bwd_workspace_bytes", plan, B, 2, 5, C),
               api.size("cfd_spectral_conv2d_bwd_workspace_bytes", plan, B, C, C))
    ws = torch.empty(ws_n + 256, dtype=torch.uint8, device=dev)
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    cases = {
        "stem_fwd": (lambda: api.call("cfd_fno_stem_fwd", plan, P(inputs), P(mask), P(cp), P(fc0w), P(fc0b), P(out), B, 2, 5, C, st), [out]),
        "dft": (lambda: api.call("cfd_spectral_dft", plan, P(a), P(xo), B * C, 0, st), [xo]),
        "dft_act": (lambda: api.call("cfd_spectral_dft", plan, P(a), P(xo), B * C, 1, st), [xo]),
        "mix": (lambda: api.call("cfd_spectral_mix", plan, P(xh), P(w1), P(w2), P(zo), B, C, C, 0, st), [zo]),
        "mix_adj_wgrad": (lambda: api.call("cfd_spectral_mix_adj_wgrad", plan, P(xh), P(gh), P(w1), P(w2), P(zo), P(gw1), P(gw2), P(ws), B, C, C, st), [zo, gw1, gw2]),
        "idft": (lambda: api.call("cfd_spectral_idft", plan, P(z), None, None, P(out), B * C, 0, st), [out]),
        "block_fwd": (lambda: api.call("cfd_fno_block_fwd", plan, P(a), P(z), P(w0), P(b0), P(out), B, C, C, 0, st), [out]),
        "block_fwd_act": (lambda: api.call("cfd_fno_block_fwd", plan, P(a), P(z), P(w0), P(b0), P(out), B, C, C, 1, st), [out]),
        "block_bwd": (lambda: api.call("cfd_fno_block_bwd_input", plan, P(gg), P(z), P(w0), None, P(out), B, C, C, st), [out]),
        "block_bwd_dgelu": (lambda: api.call("cfd_fno_block_bwd_input", plan, P(gg), P(z), P(w0), P(a2), P(out), B, C, C, st), [out]),
        "chan_wgrad_act": (lambda: api.call("cfd_chan_wgrad", P(gg), P(a), P(gw0), P(gb0), P(ws), B, C, C, HW, 1, st), [gw0, gb0]),
        "stem_bwd": (lambda: api.call("cfd_fno_stem_bwd", plan, P(gg), P(inputs), P(mask), P(cp), P(gfc0w), P(gfc0b), P(ws), B, 2, 5, C, st), [gfc0w, gfc0b]),
        "head_fwd": (lambda: api.call("cfd_fno_head_fwd", P(a), P(mask), P(label), P(fc1w), P(fc1b), P(fc2w), P(fc2b), P(preds), P(sums), P(ws), B, C, 128, 2, HW, 1, st), [preds, sums]),
        "head_bwd": (lambda: api.call("cfd_fno_head_bwd", P(a), P(mask), P(label), P(preds), None, P(coef), P(fc1w), P(fc1b), P(fc2w), P(out), P(g1w), P(g1b), P(g2w), P(g2b), P(ws), B, C, 128, 2, HW, 1, st), [out, g1w, g1b, g2w, g2b]),
        "chanmix_act": (lambda: api.call("cfd_chanmix", P(a), P(w0), P(b0), P(out), B, C, C, HW, 1, 0, st), [out]),
        "chanmix_t": (lambda: api.call("cfd_chanmix", P(gg), P(w0), None, P(out), B, C, C, HW, 0, 1, st), [out]),
        "head_train": (lambda: (api.call("cfd_label_energy_coef", P(label), P(mask), P(sums), P(coef), P(ws), B, 2, HW, 1, 1.0, st),
                                api.call("cfd_fno_head_train", P(a), P(mask), P(label), P(coef), P(fc1w), P(fc1b), P(fc2w), P(fc2b), P(preds), P(sums),
                                         P(out), P(g1w), P(g1b), P(g2w), P(g2b), P(ws), B, C, 128, 2, HW, 1, st)), [preds, sums, out, g1w, g1b, g2w, g2b]),
    }
    return cases


def main():
    for B in (int(v) for v in os.environ.get("BATCHES", "4,37").split(",")):
        C, H, W = int(os.environ.get("CH", "20")), 64, 64
        HW = H * W
        cases = build_cases(B, C)
        only = [v for v in os.environ.get("ONLY", "").split(",") if v]
        for name, (fn, outs) in cases.items():
            if only and name not in only:
                continue
            ref, bad = None, []
            if os.environ.get("MARK"):
                Path(os.environ["MARK"]).write_text(name)
            for rep in range(REPS):
                for o in outs:
                    o.fill_(float("nan")) if o.is_floating_point() else None
                fn()
                torch.cuda.synchronize()
                cur = [o.clone() for o in outs]
                if ref is None:
                    ref = cur
                elif not all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(cur, ref)):
                    bad.append(rep)
                    if os.environ.get("DUMP_DIR") and len(bad) <= int(os.environ.get("DUMP_MAX", "6")):
                        torch.save(dict(name=name, B=B, C=C, rep=rep, cur=[t.cpu() for t in cur], ref=[t.cpu() for t in ref]),
                                   Path(os.environ["DUMP_DIR"]) / f"{name}_B{B}_rep{rep}.pt")
                    if len(bad) <= 3:
                        for oi, (x, y) in enumerate(zip(cur, ref)):
                            d = (x.view(torch.int32) != y.view(torch.int32)).flatten().nonzero().flatten()
                            if d.numel():
                                xf, yf = x.flatten()[d], y.flatten()[d]
                                print(f"    rep {rep} output {oi}: {d.numel()} of {x.numel()} differ, first idx {d[:12].tolist()}, "
                                      f"max abs {float((xf - yf).abs().max()):.3e}, got {xf[:4].tolist()} ref {yf[:4].tolist()}", flush=True)
                                if x.dim() == 4 and x.shape[-2:] == (H, W):  # which 64-pixel tiles / lanes (pixel 4 n + j of a tile)
                                    pix = d % HW
                                    img = d // HW
                                    tiles = sorted({(int(i), int(t)) for i, t in zip(img.tolist(), (pix // 64).tolist())})
                                    print(f"      (plane, tile): {tiles[:8]}{' ...' if len(tiles) > 8 else ''}; lanes n = "
                                          f"{sorted(set(((pix % 64) // 4).tolist()))}; phases j = {sorted(set((pix % 4).tolist()))}", flush=True)
            if os.environ.get("DBG_FETCH"):  # diagnostic build (-DCFD_HDIAG=2048): in-kernel records of zero LDS reads
                import ctypes
                lib = ctypes.CDLL(str(_lib.lib_path()))
                buf = (ctypes.c_uint32 * (1 + 64 * 24))()
                rc = lib.cfd_debug_fetch(buf)
                print(f"    cfd_debug_fetch rc={rc}: {buf[0]} zero-read events", flush=True)
                for k in range(min(int(buf[0]), 64)):
                    o = buf[1 + 24 * k: 1 + 24 * (k + 1)]
                    hw, gpr, lds = o[15], o[16], o[17]
                    if o[0] == 4:
                        import struct
                        fl = lambda u: struct.unpack("f", struct.pack("I", u))[0]  # noqa: E731
                        print(f"      PACKED!=SCALAR block={o[1]} wave={o[2] >> 6} lane={o[2] & 63} (q={(o[2] & 63) >> 4} n={o[2] & 15}) tile={o[3]} j={o[4]} "
                              f"x packed {fl(o[7]):.6f} scalar {fl(o[8]):.6f} | y packed {fl(o[9]):.6f} scalar {fl(o[10]):.6f} | "
                              f"xcc{(hw >> 16) & 15}.se{(hw >> 13) & 7}.cu{(hw >> 8) & 15}.simd{(hw >> 4) & 3}.w{hw & 15} vgpr_base={gpr & 0x1ff} "
                              f"vgpr_size={(gpr >> 12) & 0xff} lds_base={lds & 0xff} lds_size={(lds >> 12) & 0x1ff}", flush=True)
                        continue
                    if o[0] == 3:
                        f = lambda h: f"xcc{(h >> 16) & 15}.se{(h >> 13) & 7}.cu{(h >> 8) & 15}.simd{(h >> 4) & 3}.w{h & 15}"  # noqa: E731
                        print(f"      GAP block={o[1]} wave={o[2] >> 6} tile={o[3]} (plane b={o[3] // 64} t={o[3] % 64}) j={o[4]} hw {f(o[5])} -> {f(o[6])} "
                              f"{'MOVED ' if o[5] != o[6] else ''}dt={o[7] / 100:.1f} us vgpr_base={gpr & 0x1ff} lds_base={lds & 0xff}", flush=True)
                        continue
                    print(f"      kind={'b1' if o[0] == 1 else 'w2'} block={o[1]} thread={o[2]} (wave {o[2] >> 6} lane {o[2] & 63} q={(o[2] & 63) >> 4}) tile={o[3]} j={o[4]} "
                          f"mt={o[5]} r={o[6]} read={[hex(v) for v in o[7:11]]} reread={[hex(v) for v in o[11:15]]} HW_ID={hw:#010x} "
                          f"(wave_id {hw & 15} simd {(hw >> 4) & 3} cu {(hw >> 8) & 15} sh {(hw >> 12) & 1} se {(hw >> 13) & 7} xcc {(hw >> 16) & 15}) "
                          f"GPR_ALLOC={gpr:#x} (vgpr base {gpr & 0x1ff}) LDS_ALLOC={lds:#x} (base {(lds & 0xff)} size {(lds >> 12) & 0x1ff}) "
                          f"lds_addr={o[18]:#x} t={o[19]}", flush=True)
            print(f"B={B:3d} {name:16s}", "ok" if not bad else f"NONDETERMINISTIC in reps {bad[:10]} ({len(bad)} of {REPS})", flush=True)


if __name__ == "__main__":
    main()
