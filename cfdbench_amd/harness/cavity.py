"""Native loaders for CFDBench's lid-driven-cavity problem (the data format feeding the hot path):
the on-disk layout ``<data_dir>/cavity/{prop,bc,geo}/case<NNNN>/{u.npy, v.npy, case.json}``, preprocessing, split and
item conventions of ``src/dataset/cavity.py`` -- ``CavityFlowAutoDataset`` (:220-355), ``CavityFlowDataset`` (:34-217),
``get_cavity_auto_datasets`` / ``get_cavity_datasets`` (:355-441) -- so the harness runs ``--data cavity_*`` without the
reference's package.

Differences that are cost, not behaviour: frames are stacked with NumPy and the steady-state cut-off
(``stable_state_diff``, cavity.py:307-316) is found with one vectorised pass per case instead of a Python loop over
frames; ``device=`` keeps the stacked frames resident on the GPU so a training step does no host-to-device copy of
fields (the reference's collate does four per step, train_auto.py:53-58).  The other three problems (tube, dam,
cylinder; ~1500 lines of boundary padding) still go through the reference's package (harness/data.py)."""
from __future__ import annotations

import json
import random
from bisect import bisect_right
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor
from torch.utils.data import Dataset

DATA_DELTA_TIME = 0.1  # seconds between two stored frames (cavity.py:65,240)


def normalize_physics_props(case_params: Dict[str, float]) -> None:
    """In place, src/dataset/utils.py:8-21."""
    case_params["density"] = (case_params["density"] - 5) / 4
    case_params["viscosity"] = (case_params["viscosity"] - 0.00238) / 0.005


def normalize_bc(case_params: Dict[str, float], key: str) -> None:
    """In place, src/dataset/utils.py:24-28."""
    case_params[key] = case_params[key] / 50 - 0.5


def load_case_data(case_dir: Path) -> Tuple[np.ndarray, Dict[str, float]]:
    """(T, 3, h, w) float array [u, v, mask = 1] and the case.json dict (cavity.py:15-37)."""
    with open(Path(case_dir) / "case.json", "r", encoding="utf8") as f:
        case_params = json.load(f)
    u = np.load(Path(case_dir) / "u.npy")
    v = np.load(Path(case_dir) / "v.npy")
    return np.stack([u, v, np.ones_like(u)], axis=1), case_params


def _prepared_params(case_params: Dict[str, float], norm_props: bool, norm_bc: bool) -> Dict[str, float]:
    if norm_props:
        normalize_physics_props(case_params)
    if norm_bc:
        normalize_bc(case_params, "vel_top")
    return case_params


class CavityFlowAutoDataset(Dataset):
    """Items ``(input (3,h,w), label (3,h,w), case_params {key: 0-dim float32 tensor})``: frame t -> frame
    t + delta_time/0.1 of every case up to (excluding) the first pair whose mean speed change drops below
    ``stable_state_diff`` (cavity.py:262-355).  Attributes of the reference class are kept: ``all_features`` (list of
    (T,3,h,w) arrays, read by test_multistep.py), ``case_params`` (list of dicts), ``inputs``, ``labels``, ``case_ids``."""

    data_delta_time = DATA_DELTA_TIME

    def __init__(self, case_dirs: List[Path], norm_props: bool, norm_bc: bool, delta_time: float = 0.1,
                 stable_state_diff: float = 0.001, device: Optional[str] = None):
        self.case_dirs = case_dirs
        self.norm_props = norm_props
        self.norm_bc = norm_bc
        self.delta_time = delta_time
        self.stable_state_diff = stable_state_diff
        self.time_step_size = int(self.delta_time / self.data_delta_time)
        self.case_params: List[dict] = []
        self.all_features: List[np.ndarray] = []
        ins, outs, ids = [], [], []
        step = self.time_step_size
        for case_id, case_dir in enumerate(case_dirs):
            feats, params = load_case_data(case_dir)
            self.all_features.append(feats)
            self.case_params.append(_prepared_params(params, norm_props, norm_bc))
            f32 = feats.astype(np.float32)
            inp, out = f32[:-step], f32[step:]
            # steady state: the reference compares float32 speed fields frame by frame and stops at the first hit
            diff = np.abs(np.sqrt(inp[:, 0] ** 2 + inp[:, 1] ** 2) - np.sqrt(out[:, 0] ** 2 + out[:, 1] ** 2)) \
                .reshape(len(inp), -1).mean(axis=1, dtype=np.float32)
            hit = np.nonzero(diff < np.float32(stable_state_diff))[0]
            n = int(hit[0]) if len(hit) else len(inp)
            if np.isnan(inp[:n]).any() or np.isnan(out[:n]).any():
                raise AssertionError(f"NaN in {case_dir}")
            ins.append(inp[:n])
            outs.append(out[:n])
            ids += [case_id] * n
        self.inputs = torch.from_numpy(np.concatenate(ins))
        self.labels = torch.from_numpy(np.concatenate(outs))
        self.case_ids = np.array(ids)
        if device is not None:
            self.inputs, self.labels = self.inputs.to(device), self.labels.to(device)

    def __getitem__(self, idx: int):
        case_params = {k: torch.tensor(v, dtype=torch.float32) for k, v in self.case_params[self.case_ids[idx]].items()}
        return self.inputs[idx], self.labels[idx], case_params

    def __len__(self) -> int:
        return len(self.inputs)


class CavityFlowDataset(Dataset):
    """Non-autoregressive view (cavity.py:34-217): items ``(case_params (5,), t (1,), frame (3,h,w))`` for whole frames,
    or ``(case_params, (t,x,y), value)`` point samples when ``sample_point_by_point``."""

    data_delta_time = DATA_DELTA_TIME
    case_params_keys = ["vel_top", "density", "viscosity", "height", "width"]

    def __init__(self, case_dirs: List[Path], norm_props: bool, norm_bc: bool, sample_point_by_point: bool = False,
                 stable_state_diff: float = 0.001):
        self.case_dirs = case_dirs
        self.norm_props = norm_props
        self.norm_bc = norm_bc
        self.sample_point_by_point = sample_point_by_point
        self.stable_state_diff = stable_state_diff
        self.case_params: List[Tensor] = []
        self.num_features = 0
        self.num_frames: List[int] = []
        self.features: List[Tensor] = []
        self.all_features: List[np.ndarray] = []
        for case_dir in case_dirs:
            feats, params = load_case_data(case_dir)
            params = _prepared_params(params, norm_props, norm_bc)
            T, c, h, w = feats.shape
            self.num_features += T * h * w
            self.all_features.append(feats)
            self.case_params.append(torch.tensor([params[k] for k in self.case_params_keys], dtype=torch.float32))
            self.features.append(torch.tensor(feats, dtype=torch.float32))
            self.num_frames.append(T)
        self.case_ids = torch.arange(len(case_dirs))
        self.num_frames_before = list(np.cumsum(self.num_frames).tolist())

    def idx_to_case_id_and_frame_idx(self, idx: int) -> Tuple[int, int]:
        case_id = bisect_right(self.num_frames_before, idx)
        return case_id, idx if case_id == 0 else idx - self.num_frames_before[case_id - 1]

    def __getitem__(self, idx: int):
        if self.sample_point_by_point:
            h, w = self.features[0].shape[2:]
            case_id, t = self.idx_to_case_id_and_frame_idx(idx // (h * w))
            pix = idx % (h * w)
            y, x = pix // w, pix % w
            return self.case_params[case_id], torch.tensor([t, x, y]).float(), self.features[case_id][t, :, y, x].squeeze().float()
        case_id, frame_idx = self.idx_to_case_id_and_frame_idx(idx)
        return self.case_params[case_id], torch.tensor([frame_idx]).float(), self.features[case_id][frame_idx]

    def __len__(self) -> int:
        return self.num_features if self.sample_point_by_point else self.num_frames_before[-1]


def _split_case_dirs(data_dir: Path, case_name: str, seed: int):
    """Subsets named in ``case_name`` in the fixed order prop, bc, geo; case<N> sorted by N; ``random.seed(seed)``
    shuffle; 80 / 10 / 10 split with ``round`` (cavity.py:366-382,403-421)."""
    case_dirs: List[Path] = []
    for name in ["prop", "bc", "geo"]:
        if name in case_name:
            case_dirs += sorted((Path(data_dir) / name).glob("case*"), key=lambda x: int(x.name[4:]))
    assert case_dirs != [], f"no cavity cases under {data_dir} for subset {case_name!r}"
    random.seed(seed)
    random.shuffle(case_dirs)
    n = len(case_dirs)
    n_train, n_dev = round(n * 0.8), round(n * 0.1)
    return case_dirs[:n_train], case_dirs[n_train:n_train + n_dev], case_dirs[n_train + n_dev:]


def get_cavity_auto_datasets(data_dir: Path, case_name: str, norm_props: bool, norm_bc: bool, delta_time: float = 0.1,
                             stable_state_diff: float = 0.001, seed: int = 0, device: Optional[str] = None):
    tr, dv, te = _split_case_dirs(data_dir, case_name, seed)
    kw = dict(delta_time=delta_time, stable_state_diff=stable_state_diff, norm_props=norm_props, norm_bc=norm_bc,
              device=device)
    return CavityFlowAutoDataset(tr, **kw), CavityFlowAutoDataset(dv, **kw), CavityFlowAutoDataset(te, **kw)


def get_cavity_datasets(data_dir: Path, case_name: str, norm_props: bool, norm_bc: bool, seed: int = 0):
    tr, dv, te = _split_case_dirs(data_dir, case_name, seed)
    kw = dict(norm_props=norm_props, norm_bc=norm_bc)
    return CavityFlowDataset(tr, **kw), CavityFlowDataset(dv, **kw), CavityFlowDataset(te, **kw)
