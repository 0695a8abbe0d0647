"""Cavity-problem names of the native loaders (drop-in for ``dataset.cavity``); the implementation shared with the tube
and dam problems lives in harness/flow_data.py."""
from __future__ import annotations

from pathlib import Path
from typing import List, Optional

from .flow_data import (DATA_DELTA_TIME, PROBLEMS, FlowAutoDataset, FlowDataset, get_flow_auto_datasets,  # noqa: F401
                        get_flow_datasets, load_cavity_case as load_case_data, normalize_bc, normalize_physics_props)


class CavityFlowAutoDataset(FlowAutoDataset):
    def __init__(self, case_dirs: List[Path], norm_props: bool, norm_bc: bool, delta_time: float = 0.1,
                 stable_state_diff: float = 0.001, device: Optional[str] = None):
        super().__init__(PROBLEMS["cavity"], case_dirs, norm_props, norm_bc, delta_time, stable_state_diff, device)


class CavityFlowDataset(FlowDataset):
    def __init__(self, case_dirs: List[Path], norm_props: bool, norm_bc: bool, sample_point_by_point: bool = False,
                 stable_state_diff: float = 0.001):
        super().__init__(PROBLEMS["cavity"], case_dirs, norm_props, norm_bc, sample_point_by_point, stable_state_diff)


def get_cavity_auto_datasets(data_dir: Path, case_name: str, norm_props: bool, norm_bc: bool, delta_time: float = 0.1,
                             stable_state_diff: float = 0.001, seed: int = 0, device: Optional[str] = None):
    return get_flow_auto_datasets("cavity", data_dir, case_name, norm_props, norm_bc, delta_time, stable_state_diff, seed,
                                  device)


def get_cavity_datasets(data_dir: Path, case_name: str, norm_props: bool, norm_bc: bool, seed: int = 0):
    return get_flow_datasets("cavity", data_dir, case_name, norm_props, norm_bc, seed)
