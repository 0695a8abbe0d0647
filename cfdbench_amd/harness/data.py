"""Dataset glue.  The on-disk format and preprocessing of CFDBench (src/dataset/*.py, ~2500 lines of NumPy) feed the hot
path but are not part of it ("next" row 1): ``get_auto_dataset`` routes every benchmark problem
(cavity, tube, dam, cylinder) to the native loaders of harness/flow_data.py.
``SyntheticAutoDataset`` produces items of exactly the reference's shape -- ``(inputs (3,h,w), label (3,h,w),
case_params dict)`` with the mask as the last channel (src/dataset/base.py, cavity.py:333-347) -- for smoke runs, tests
and benchmarks without data on disk."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticAutoDataset(Dataset):
    """Smooth random velocity fields advected by a fixed linear map: learnable, deterministic, no files.

    all_features: list over cases of (T, 3, h, w) float32 arrays [u, v, mask]; case_params: list of dicts in the
    reference's key order (vel_top, density, viscosity, height, width) -- the attributes test_multistep.py:195-212 reads."""

    def __init__(self, n_cases: int = 4, n_frames: int = 6, height: int = 64, width: int = 64, seed: int = 0,
                 border_mask: bool = False):
        rng = np.random.default_rng(seed)
        yy, xx = np.meshgrid(np.linspace(0, 1, height), np.linspace(0, 1, width), indexing="ij")
        self.all_features: List[np.ndarray] = []
        self.case_params: List[Dict[str, float]] = []
        mask = np.ones((height, width), np.float32)
        if border_mask:
            mask[0, :] = mask[-1, :] = mask[:, 0] = 0.0
        for _ in range(n_cases):
            vel, dens, visc = rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5), rng.uniform(0.5, 1.5)
            amp = rng.standard_normal((2, 3, 3)) / (1.0 + np.arange(3)[None, :, None] + np.arange(3)[None, None, :])
            frames = []
            for t in range(n_frames):
                ph = 0.15 * t * vel
                u = sum(amp[0, k, l] * np.sin(2 * np.pi * (k * xx + l * yy) + ph) for k in range(3) for l in range(3))
                v = sum(amp[1, k, l] * np.cos(2 * np.pi * (k * xx - l * yy) + ph * dens) for k in range(3) for l in range(3))
                frames.append(np.stack([u * mask, v * mask, mask]).astype(np.float32))
            self.all_features.append(np.stack(frames))
            self.case_params.append(dict(vel_top=float(vel), density=float(dens), viscosity=float(visc),
                                         height=1.0, width=1.0))
        self.index: List[Tuple[int, int]] = [(c, t) for c in range(n_cases) for t in range(n_frames - 1)]
        # the stacked view the reference's autoregressive datasets expose (cavity.py:345-355): frame t -> frame t + 1
        self.inputs = torch.from_numpy(np.concatenate([f[:-1] for f in self.all_features]))
        self.labels = torch.from_numpy(np.concatenate([f[1:] for f in self.all_features]))
        self.case_ids: List[int] = [c for c, _ in self.index]

    def __len__(self) -> int:
        return len(self.index)

    def __getitem__(self, i: int):
        c, t = self.index[i]
        f = self.all_features[c]
        return torch.from_numpy(f[t]), torch.from_numpy(f[t + 1]), self.case_params[c]


def get_auto_dataset(data_dir: Path, data_name: str, delta_time: float, norm_props: bool, norm_bc: bool,
                     load_splits: Optional[List[str]] = None):
    """(train, dev, test) CfdAutoDatasets (src/dataset/__init__.py:64) through the native loaders
    (harness/flow_data.py); ``load_splits`` is honoured for the cylinder problem only, as in the reference."""
    problem = data_name.split("_")[0]
    if problem in ("cavity", "tube", "dam", "cylinder"):
        from .flow_data import get_flow_auto_datasets
        assert delta_time > 0
        kw = dict(load_splits=tuple(load_splits)) if (problem == "cylinder" and load_splits is not None) else {}
        return get_flow_auto_datasets(problem, Path(data_dir) / problem, data_name[len(problem) + 1:],
                                      norm_props=norm_props, norm_bc=norm_bc, delta_time=delta_time, **kw)
    raise ValueError(f'Invalid data name "{data_name}"')  # src/dataset/__init__.py:125


class DeviceBatchLoader:
    """On-device batch assembly: the replacement for ``DataLoader(data, collate_fn=collate_fn)`` when the
    frames of an autoregressive dataset fit in HBM (the whole interpolated CFDBench corpus is 13.4 GB).

    The reference builds a batch item by item on the host -- ``__getitem__`` makes a dict of 0-dim tensors per sample,
    ``collate_fn`` stacks the frames, rebuilds the case-parameter table from Python floats and copies four tensors to the
    GPU (src/train_auto.py:33-58) -- which tops out at a few thousand frames/s, two orders of magnitude below the
    training step.  Here the stacked frames ``data.inputs`` / ``data.labels`` (N, 3, h, w) and the per-frame parameter
    table live on the device once; a batch is four ``index_select`` gathers with the epoch's permutation, and the
    yielded dict is exactly ``collate_fn``'s: ``inputs`` = channels [:-1], ``mask`` = the last channel, ``label`` =
    channels [:-1] of the next frame, ``case_params`` = every case.json key except rotated / dx / dy, in key order.

    ``data`` needs the attributes of the reference's datasets: ``inputs``, ``labels``, ``case_ids``, ``case_params``.
    ``indices`` restricts the loader to a subset (data-parallel shard).  The permutation is drawn from the host generator
    (``torch.randperm``), so seeding / ``--resume`` behave as with a DataLoader."""

    def __init__(self, data, batch_size: int, shuffle: bool = True, drop_last: bool = False, device: str = "cuda",
                 indices=None, generator=None):
        self.batch_size, self.shuffle, self.drop_last, self.generator = int(batch_size), shuffle, drop_last, generator
        # resident planes, split the way a batch is consumed: fields (N, c, h, w), mask (N, 1, h, w), next frame's fields (N, c, h, w).
        # (Round 6: one gather per batch tensor and no `.contiguous()` of channel slices -- three copy launches less per batch; the labels'
        # mask channel, which no consumer reads, is not kept.)
        self.inputs = data.inputs[:, :-1].contiguous().to(device)
        self.mask = data.inputs[:, -1:].contiguous().to(device)
        self.labels = data.labels[:, :-1].contiguous().to(device)
        self._bound = None
        keys = [k for k in data.case_params[0].keys() if k not in ("rotated", "dx", "dy")]
        table = torch.tensor([[float(cp[k]) for k in keys] for cp in data.case_params], dtype=torch.float32)
        self.case_params = table[torch.as_tensor(data.case_ids, dtype=torch.long)].to(device)
        self.set_indices(indices)

    def set_indices(self, indices=None) -> None:
        """Restrict the loader to ``indices`` (None = every frame) WITHOUT touching the resident tensors: a data-parallel run
        re-partitions the frames every epoch, and only this index list changes -- the corpus is uploaded once."""
        n = len(self.inputs)
        self.indices = torch.arange(n) if indices is None else torch.as_tensor(indices, dtype=torch.long)

    def __len__(self) -> int:
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.indices
        if self.shuffle:
            order = order[torch.randperm(len(order), generator=self.generator)]
        order = order.to(self.inputs.device)
        for k in range(len(self)):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            src = dict(inputs=self.inputs, label=self.labels, mask=self.mask, case_params=self.case_params)
            out = self._bound
            if out is not None and len(idx) == out["inputs"].shape[0]:
                # gathered STRAIGHT into the buffers a captured training step reads (graph.GraphedTrainStep.static): the step then finds
                # its own tensors in the batch and copies nothing (four device-to-device copies per step before)
                for name, t in src.items():
                    torch.index_select(t, 0, idx, out=out[name])
                yield dict(out)
            else:
                yield {name: t.index_select(0, idx) for name, t in src.items()}

    def bind(self, buffers) -> None:
        """Gather every full-size batch into ``buffers`` (a dict with inputs / label / mask / case_params tensors of one batch's shapes and
        this loader's device, e.g. ``GraphedTrainStep.static``) instead of fresh tensors; ``None`` unbinds.  The yielded dict then holds
        those very tensors -- valid until the next batch is drawn."""
        if buffers is not None:
            for name, ref in dict(inputs=self.inputs, label=self.labels, mask=self.mask, case_params=self.case_params).items():
                b = buffers.get(name)
                if b is None or b.shape[1:] != ref.shape[1:] or b.dtype != ref.dtype or b.device != ref.device or not b.is_contiguous():
                    raise ValueError(f"DeviceBatchLoader.bind: buffer '{name}' does not match the resident tensor")
        self._bound = buffers


def overlapping_copy_stream(device=None, tries: int = 8, work_passes: int = 40):
    """A stream whose host-to-device copies run WHILE the kernels of the current stream execute -- for callers that own pinned HOST
    batches and upload batch i + 1 during step i (bench.py's ``train_host_batches`` leg; the resident ``DeviceBatchLoader`` needs none).

    Not every stream can: HIP maps its streams round-robin onto a few hardware queues (four here), and a stream that shares the
    current stream's queue executes behind it -- on MI355X every fourth ``torch.cuda.Stream()`` serialises with the default stream
    (measured, tools/exp/host_batches.py: step 1.29 ms with the upload hidden on 33 of 40 fresh streams, 1.69 ms = step + copy on the
    other 7).  So candidates are probed: ~2 ms of kernels on the current stream, a small pinned copy on the candidate behind the same
    start event, and the candidate is taken when its copy finished in under half the kernels' time.  Returns ``(stream, overlaps)``;
    ``overlaps`` is False when no candidate passed (the last one is returned)."""
    cur = torch.cuda.current_stream(device)
    dev = cur.device
    x = torch.empty(32 << 20, dtype=torch.float32, device=dev)
    h = torch.zeros(1024, dtype=torch.float32).pin_memory()
    d = torch.empty(1024, dtype=torch.float32, device=dev)
    d.copy_(h)  # (first touch of the freshly pinned page)
    cand = None
    for _ in range(tries):
        cand = torch.cuda.Stream(dev)
        start, k_end, c_end = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize(dev)
        start.record(cur)
        for _ in range(work_passes):
            x.mul_(1.0)
        k_end.record(cur)
        cand.wait_event(start)
        with torch.cuda.stream(cand):
            d.copy_(h, non_blocking=True)
        c_end.record(cand)
        torch.cuda.synchronize(dev)
        if start.elapsed_time(c_end) < 0.5 * start.elapsed_time(k_end):
            return cand, True
    return cand, False
