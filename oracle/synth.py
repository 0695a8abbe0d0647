"""Deterministic synthetic weights / batches shared by the golden generator, the tests and bench.py.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  NumPy's PCG64 stream is bit-stable across
machines, so a fixture only needs to store the *outputs* the reference produced for these inputs.
Shapes and distributions follow SURVEY.md section 8(d) ("Synthetic inputs").
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np


def fno_param_shapes(C: int, L: int, m1: int, m2: int, p: int, in_chan: int = 2, out_chan: int = 2,
                     head: int = 128) -> Dict[str, Tuple[Tuple[int, ...], bool]]:
    """Reference state_dict layout of Fno2d (src/models/fno/fno2d.py:150-176); value = (shape, is_complex)."""
    cin0 = in_chan + 1 + 2 + p
    shapes: Dict[str, Tuple[Tuple[int, ...], bool]] = {}
    shapes["fc0.weight"] = ((C, cin0, 1, 1), False)
    shapes["fc0.bias"] = ((C,), False)
    for l in range(L):
        shapes[f"blocks.{l}.conv0.weights1"] = ((C, C, m1, m2), True)
        shapes[f"blocks.{l}.conv0.weights2"] = ((C, C, m1, m2), True)
        shapes[f"blocks.{l}.w0.weight"] = ((C, C, 1, 1), False)
        shapes[f"blocks.{l}.w0.bias"] = ((C,), False)
    shapes["fc1.weight"] = ((head, C, 1, 1), False)
    shapes["fc1.bias"] = ((head,), False)
    shapes["fc2.weight"] = ((out_chan, head, 1, 1), False)
    shapes["fc2.bias"] = ((out_chan,), False)
    return shapes


def make_fno_params(seed: int, C: int = 20, L: int = 4, m1: int = 12, m2: int = 12, p: int = 5,
                    dtype=np.float32, spectral_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Weights with the reference's init *distributions* (conv: U(+-1/sqrt(fan_in)); spectral:
    scale*U[0,1) re/im, scale = 1/(Cin*Cout), fno2d.py:31-51) drawn from a NumPy stream.
    ``spectral_gain`` > 1 makes the spectral branch numerically visible next to the 1x1 branch."""
    rng = np.random.default_rng(seed)
    cd = np.complex64 if dtype == np.float32 else np.complex128
    out: Dict[str, np.ndarray] = {}
    for name, (shape, is_c) in fno_param_shapes(C, L, m1, m2, p).items():
        if is_c:
            scale = spectral_gain / (shape[0] * shape[1])
            re = rng.random(shape)
            im = rng.random(shape)
            out[name] = (scale * (re + 1j * im)).astype(cd)
        else:
            fan_in = shape[1] if len(shape) == 4 else None
            if fan_in is None:  # bias: fan_in of the matching weight
                wname = name.replace("bias", "weight")
                fan_in = out[wname].shape[1]
            bound = 1.0 / np.sqrt(fan_in)
            out[name] = rng.uniform(-bound, bound, size=shape).astype(dtype)
    return out


def make_fno_propagator_params(seed: int, C: int = 20, L: int = 4, m1: int = 12, m2: int = 12, p: int = 5, eps: float = 0.05,
                               spectral_gain: float = 1.0, signal: float = 1.0, decay: float = 0.0,
                               dtype=np.float32, route: str = "w0") -> Dict[str, np.ndarray]:
    """Weights that make Fno2d a NEAR-IDENTITY map x_{t+1} = x_t + O(eps) -- a stand-in for a trained one-step propagator
    in rollout studies (random-init weights collapse every input to one fixed point within five steps, so a 200-step
    rollout of them exercises nothing).  Construction: gelu(x) - gelu(-x) = x exactly, so the pair of channels (+s u, -s u)
    survives every GELU of the network when each following linear layer takes the difference of the pair:
      fc0: ch0 = s u, ch1 = -s u, ch2 = s v, ch3 = -s v;  w0 of block 0 (no GELU before it): identity on those channels;
      w0 of blocks >= 1: ch0' = h0 - h1, ch1' = h1 - h0, ... ;  fc1: hidden 0..3 likewise;  fc2: u' = (1 - decay) (g0 - g1) / s
    (``decay`` balances the growth the random perturbation adds, so that 200-step rollouts stay O(1)).
    Every other weight (incl. all biases and the spectral weights, which couple the pixels) is the usual random init
    scaled by ``eps``.  Needs C >= 4.
    ``route`` = "spectral" (round 3): the blocks' identity goes through SpectralConv2d instead of the 1x1 conv -- the same pair
    arithmetic written into weights1 / weights2 at EVERY kept mode, i.e. identity o the low-pass projector onto the kept modes,
    which is the identity on the band-limited fields of ``make_smooth_batch`` (and removes the out-of-band harmonics GELU adds,
    so the pair difference still returns the field exactly).  A 200-step rollout of this network sends all its energy through
    DFT -> mode mixing -> inverse DFT in every layer: the fixture for the rounding of the transforms' fixed twiddle operands."""
    assert C >= 4 and route in ("w0", "spectral")
    base = make_fno_params(seed, C, L, m1, m2, p, dtype=dtype, spectral_gain=spectral_gain)
    out = {k: (v * eps).astype(v.dtype) for k, v in base.items()}
    s = signal
    sgn = np.array([1.0, -1.0])
    for c in range(4):  # fc0 input features: [u, v, mask, grid_x, grid_y, props...]
        out["fc0.weight"][c, :, 0, 0] *= 1.0
        out["fc0.weight"][c, c // 2, 0, 0] += s * sgn[c % 2]
    for l in range(L):
        w = out[f"blocks.{l}.w0.weight"]
        for c in range(4):
            pair = 2 * (c // 2)
            if route == "spectral":  # weights{1,2}[in, out, k, l] (fno2d.py:54-57: einsum "bixy,ioxy->boxy")
                for key in (f"blocks.{l}.conv0.weights1", f"blocks.{l}.conv0.weights2"):
                    if l == 0:
                        out[key][c, c] += 1.0
                    else:
                        out[key][pair, c] += sgn[c % 2]
                        out[key][pair + 1, c] -= sgn[c % 2]
            elif l == 0:
                w[c, c, 0, 0] += 1.0
            else:
                w[c, pair, 0, 0] += sgn[c % 2]
                w[c, pair + 1, 0, 0] -= sgn[c % 2]
    for c in range(4):
        pair = 2 * (c // 2)
        if L == 0:
            out["fc1.weight"][c, c, 0, 0] += 1.0
        else:
            out["fc1.weight"][c, pair, 0, 0] += sgn[c % 2]
            out["fc1.weight"][c, pair + 1, 0, 0] -= sgn[c % 2]
    for o in range(2):
        out["fc2.weight"][o, 2 * o, 0, 0] += (1.0 - decay) / s
        out["fc2.weight"][o, 2 * o + 1, 0, 0] -= (1.0 - decay) / s
    return out


def make_state_dict(shapes, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic stand-in for a module's initial ``state_dict`` given its ordered (name, shape) list -- so that a golden
    fixture of a LARGE model (U-Net dim 12: 1.1 M parameters) stores a seed instead of the weights.  Rules by name / rank:
    num_batches_tracked -> 0 (int64); running_var -> 1 + U[0, 0.5); running_mean -> 0.1 N; 1-D ``weight`` (BatchNorm
    gamma) -> 1 + 0.2 N; 1-D ``bias`` -> 0.1 N; 0-D / (1,) -> 0.03; everything else -> U(+-1/sqrt(fan_in)) with fan_in =
    prod(shape[1:])."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in shapes:
        shape = tuple(int(v) for v in shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        elif name.endswith("running_var"):
            out[name] = (1.0 + 0.5 * rng.random(shape)).astype(np.float32)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 0 or int(np.prod(shape)) == 1:
            out[name] = np.full(shape, 0.03, np.float32)
        elif len(shape) == 1 and name.endswith("weight"):
            out[name] = (1.0 + 0.2 * rng.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 1:
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            bound = 1.0 / np.sqrt(float(np.prod(shape[1:])))
            out[name] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
    return out


def summarize(a: np.ndarray, idx_seed: int, n: int = 64) -> Dict[str, np.ndarray]:
    """Compact fingerprint of a large tensor for a golden fixture: its l2 norm, its sum and ``n`` sampled entries."""
    flat = np.ascontiguousarray(a).reshape(-1)
    rng = np.random.default_rng(idx_seed)
    idx = rng.integers(0, flat.size, size=min(n, flat.size))
    return dict(norm=np.sqrt(np.sum(np.abs(flat.astype(np.complex128)) ** 2)), sum=flat.sum(dtype=np.complex128), idx=idx,
                vals=flat[idx])


def make_rollout_case(pseed: int, bseed: int, B: int, C: int, L: int, H: int, W: int, p: int, eps: float, gain: float,
                      decay: float, route: str = "w0"):
    """Weights and start frame of the long-rollout fixtures / studies: the near-identity propagator above on a band-limited
    field with the tube / dam style border mask (zero top / bottom rows and left column)."""
    params = make_fno_propagator_params(pseed, C, L, 12, 12, p, eps=eps, spectral_gain=gain, decay=decay, route=route)
    batch = make_smooth_batch(bseed, B, H, W, p)
    batch["mask"][:, :, 0, :] = 0
    batch["mask"][:, :, -1, :] = 0
    batch["mask"][:, :, :, 0] = 0
    batch["inputs"] = batch["inputs"] * batch["mask"]
    return params, batch


def make_batch(seed: int, B: int, H: int = 64, W: int = 64, p: int = 5, border_mask: bool = False,
               dtype=np.float32) -> Dict[str, np.ndarray]:
    """inputs=randn(B,2,H,W); label=inputs+0.1 randn; case_params=randn(B,p); mask=ones (cavity) or
    zero top/bottom/left border (tube/dam style) -- SURVEY.md 8(d)."""
    rng = np.random.default_rng(seed)
    inputs = rng.standard_normal((B, 2, H, W))
    label = inputs + 0.1 * rng.standard_normal((B, 2, H, W))
    case_params = rng.standard_normal((B, p))
    mask = np.ones((B, 1, H, W))
    if border_mask:
        mask[:, :, 0, :] = 0
        mask[:, :, -1, :] = 0
        mask[:, :, :, 0] = 0
    return dict(inputs=inputs.astype(dtype), label=label.astype(dtype),
                case_params=case_params.astype(dtype), mask=mask.astype(dtype))


def make_smooth_batch(seed: int, B: int, H: int = 64, W: int = 64, p: int = 5, kmax: int = 4,
                      dtype=np.float32) -> Dict[str, np.ndarray]:
    """Band-limited fields (sum of |k|<=kmax sin/cos with 1/|k|^2 amplitudes) for rollout studies."""
    rng = np.random.default_rng(seed)
    xs = np.arange(H)[:, None] / H
    ys = np.arange(W)[None, :] / W
    f = np.zeros((B, 2, H, W))
    for kx in range(-kmax, kmax + 1):
        for ky in range(0, kmax + 1):
            kk = max(1.0, float(kx * kx + ky * ky))
            a = rng.standard_normal((B, 2, 1, 1)) / kk
            b = rng.standard_normal((B, 2, 1, 1)) / kk
            ph = 2 * np.pi * (kx * xs + ky * ys)
            f += a * np.cos(ph) + b * np.sin(ph)
    out = make_batch(seed + 1, B, H, W, p, dtype=dtype)
    out["inputs"] = f.astype(dtype)
    out["label"] = (f + 0.01 * rng.standard_normal(f.shape)).astype(dtype)
    return out


def write_cavity_tree(root, seed: int = 0, h: int = 12, w: int = 12):
    """A small deterministic CFDBench-layout cavity data set under ``root``/cavity/{prop,bc,geo}/case<NNNN>/ with
    ``u.npy``, ``v.npy`` (float64, (T,h,w)) and ``case.json``: smooth fields relaxing exponentially towards a steady
    state, fast enough in some cases to trigger the loaders' steady-state cut-off.  Test infrastructure (golden
    generation + loader parity tests)."""
    import json
    from pathlib import Path
    rng = np.random.default_rng(seed)
    root = Path(root) / "cavity"
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    for subset, ids in (("prop", [0, 1, 2, 3, 10, 11, 12]), ("bc", [0, 2, 5, 7, 8, 9]), ("geo", [1, 3, 4, 20])):
        for cid in ids:
            d = root / subset / f"case{cid:04d}"
            d.mkdir(parents=True, exist_ok=True)
            T = int(rng.integers(6, 14))
            tau = float(rng.uniform(0.3, 3.0))
            vel, dens, visc = float(rng.uniform(1, 50)), float(rng.uniform(0.5, 10)), float(rng.uniform(1e-4, 1e-2))
            hh, ww = (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0))) if subset == "geo" else (1.0, 1.0)
            pat_u = np.sin(np.pi * xx) * np.sin(np.pi * yy) ** 2 * (1 + 0.3 * rng.standard_normal())
            pat_v = np.cos(np.pi * xx) * np.sin(2 * np.pi * yy) * 0.5
            ramp = (1.0 - np.exp(-(np.arange(T) + 1.0) / tau))[:, None, None]
            u = 0.2 * vel * ramp * pat_u[None]
            v = 0.2 * vel * ramp * pat_v[None]
            np.save(d / "u.npy", u)
            np.save(d / "v.npy", v)
            with open(d / "case.json", "w", encoding="utf8") as f:
                json.dump(dict(vel_top=vel, density=dens, viscosity=visc, height=hh, width=ww), f)
    return root.parent


def write_flow_tree(root, problem: str, seed: int = 0, h: int = 10, w: int = 12):
    """Like write_cavity_tree for the tube and dam problems (their case.json keys: tube ``vel_in``; dam ``velocity``,
    ``barrier_width``, ``barrier_height``, ``dx``, ``dy``): ``root``/<problem>/{prop,bc,geo}/case<NNNN>/."""
    import json
    from pathlib import Path
    if problem == "cavity":
        return write_cavity_tree(root, seed, h, w)
    if problem == "cylinder":
        return write_cylinder_tree(root, seed)
    rng = np.random.default_rng(seed + (17 if problem == "tube" else 29))
    base = Path(root) / problem
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    for subset, ids in (("prop", [0, 1, 2, 5, 6, 7, 8, 30]), ("bc", [0, 1, 3, 4, 9]), ("geo", [2, 3, 11, 12])):
        for cid in ids:
            d = base / subset / f"case{cid:04d}"
            d.mkdir(parents=True, exist_ok=True)
            T = int(rng.integers(5, 12))
            tau = float(rng.uniform(0.3, 2.5))
            vel, dens, visc = float(rng.uniform(0.5, 5)), float(rng.uniform(0.5, 10)), float(rng.uniform(1e-4, 1e-2))
            hh, ww = (float(rng.uniform(0.2, 1.0)), float(rng.uniform(1.0, 8.0))) if subset == "geo" else (0.4, 5.0)
            ramp = (1.0 - np.exp(-(np.arange(T) + 1.0) / tau))[:, None, None]
            u = vel * ramp * (4 * yy * (1 - yy))[None] * (1 + 0.1 * np.sin(2 * np.pi * xx))[None]
            v = 0.1 * vel * ramp * (np.sin(np.pi * yy) * np.cos(2 * np.pi * xx))[None]
            np.save(d / "u.npy", u)
            np.save(d / "v.npy", v)
            if problem == "tube":
                cp = dict(vel_in=vel, density=dens, viscosity=visc, height=hh, width=ww)
            else:
                cp = dict(velocity=vel, density=dens, viscosity=visc, height=hh, width=ww,
                          barrier_width=float(rng.uniform(0.1, 0.4)), barrier_height=float(rng.uniform(0.05, 0.3)),
                          dx=ww / w, dy=hh / h, extra_key_the_loader_drops=1.0)
            with open(d / "case.json", "w", encoding="utf8") as f:
                json.dump(cp, f)
    return Path(root)


def write_cylinder_tree(root, seed: int = 0, h: int = 12, w: int = 14):
    """Synthetic cylinder-problem tree (case.json: vel_in, density, viscosity, radius, x_min, x_max, y_min, y_max and --
    in some cases -- an explicit center_x / center_y): ``root``/cylinder/{prop,bc,geo}/case<NNNN>/."""
    import json
    from pathlib import Path
    rng = np.random.default_rng(seed + 41)
    base = Path(root) / "cylinder"
    yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    for subset, ids in (("prop", [0, 1, 2, 3, 4, 9, 21]), ("bc", [0, 5, 6, 7, 8]), ("geo", [1, 2, 3, 10])):
        for cid in ids:
            d = base / subset / f"case{cid:04d}"
            d.mkdir(parents=True, exist_ok=True)
            T = int(rng.integers(8, 16))
            tau = float(rng.uniform(0.5, 3.0))
            vel, dens, visc = float(rng.uniform(0.05, 0.15)), float(rng.uniform(900, 1100)), float(rng.uniform(5e-3, 1.5e-2))
            ramp = (1.0 - np.exp(-(np.arange(T) + 1.0) / tau))[:, None, None]
            u = 10 * vel * ramp * (1 + 0.3 * np.sin(2 * np.pi * xx) * np.cos(np.pi * yy))[None]
            v = 3 * vel * ramp * (np.sin(2 * np.pi * yy) * np.sin(np.pi * xx))[None]
            np.save(d / "u.npy", u)
            np.save(d / "v.npy", v)
            cp = dict(vel_in=vel, density=dens, viscosity=visc, radius=float(rng.uniform(0.05, 0.12)),
                      x_min=-float(rng.uniform(0.3, 0.6)), x_max=float(rng.uniform(0.8, 1.4)),
                      y_min=-float(rng.uniform(0.3, 0.5)), y_max=float(rng.uniform(0.3, 0.5)))
            if cid % 2 == 1:
                cp["center_x"], cp["center_y"] = float(rng.uniform(-0.05, 0.05)), float(rng.uniform(-0.05, 0.05))
            with open(d / "case.json", "w", encoding="utf8") as f:
                json.dump(cp, f)
    return Path(root)
