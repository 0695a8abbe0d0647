"""Multi-step rollout inference with the reference's metric conventions (src/test_multistep.py:73-239):
per step and per case, u channel only, ``preds*mask`` vs ``label*mask`` with prediction k (= frame k+1) compared
against label frame k, averaged over cases -> multistep_metrics.json.

    python -m cfdbench_amd.harness.test_multistep --model fno --data dam_prop_bc_geo --infer_steps 200

Cases are independent, so ``infer`` rolls ALL cases out as one batch (one launch sequence per step instead of one per
case and step) -- for models whose forward is per-sample in the mode they are in.  The reference never calls
``model.eval()`` here (test_multistep.py:180-239) and rolls out one case at a time, so a U-Net's BatchNorm normalises
with the statistics of that ONE case and a ResNet's dropout stays active; batching such a model in training mode would
pool the statistics over the cases and change the metrics.  ``infer`` therefore falls back to the reference's per-case
loop (``infer_case``) whenever the model holds BatchNorm / Dropout layers in training mode.  Metrics are reduced on the
device and fetched once at the end instead of three ``.item()`` syncs per case and step.

Multi-GPU: cases shard over the ranks of a ``torch.distributed`` launch (one process per GPU) -- rank r rolls
out cases r, r + world, ... with NO collective on the data path; the per-step sums of the per-case metrics (a (steps, 3)
fp64 tensor + the case count) are all-reduced ONCE at the end and rank 0 writes multistep_metrics.json:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m cfdbench_amd.harness.test_multistep --model fno --data dam_prop_bc_geo --infer_steps 200
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List

import numpy as np
import torch
from torch import Tensor

from .args import Args, is_args_valid
from .autoregressive import init_model
from .common import dump_json, get_output_dir, load_best_ckpt


def get_metrics(preds: Tensor, labels: Tensor) -> Dict[str, float]:  # test_multistep.py:73-82
    """Same numbers as the reference's helper with ONE device-to-host transfer instead of three (``infer`` below does not call
    it at all: it reduces every step and case on the device and transfers once per rollout)."""
    assert preds.shape == labels.shape, f"{preds.shape}, {labels.shape}"
    d = (preds - labels).detach()
    mse, l2, mae = torch.stack([(d * d).mean(), (labels.detach() ** 2).mean(), d.abs().mean()]).cpu().tolist()
    return dict(mse=mse, nmse=mse / l2, mae=mae)


def case_params_to_tensor(case_params_dict: dict) -> Tensor:  # test_multistep.py:85-92
    keys = [x for x in case_params_dict.keys() if x not in ["rotated", "dx", "dy"]]
    return torch.tensor([case_params_dict[k] for k in keys], dtype=torch.float32)


def infer_case(model, case_features: Tensor, case_params: Tensor, infer_steps: int) -> List[Tensor]:
    """One case: start from frame 0 (channels [:-1]) with mask = last channel (test_multistep.py:102-132)."""
    with torch.no_grad():
        return model.generate_many(inputs=case_features[0, :-1], case_params=case_params, mask=case_features[0, -1],
                                   steps=infer_steps)


def batch_dependent(model) -> bool:
    """True when a batched forward differs from per-sample forwards: BatchNorm with batch statistics or active dropout."""
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training:
            return True
        if isinstance(m, torch.nn.Dropout) and m.training and m.p > 0:
            return True
    return False


def rollout_frames(model, start: Tensor, cps: Tensor, mask: Tensor, infer_steps: int, dtype: str = "fp32"):
    """``infer_steps`` predicted frames for a batch of cases.  An Fno2d rolls out from ONE HIP graph per horizon
    (cfdbench_amd.rollout.FnoRollout: fp32 = bitwise ``generate_many``; "bf16" = bf16 activation storage); other models go through
    their own ``generate_many``."""
    from ..models.fno.fno2d import Fno2d
    from ..rollout import ACT_DTYPES, FnoRollout
    if isinstance(model, Fno2d) and start.is_cuda:
        if dtype not in ACT_DTYPES:
            raise ValueError(f"--dtype {dtype}: one of {sorted(ACT_DTYPES)}")
        ro = getattr(model, "_multistep_rollout", None)
        if ro is None or ro.dtype != dtype:
            ro = FnoRollout(model, dtype=dtype)
            object.__setattr__(model, "_multistep_rollout", ro)  # graph cache lives with the model (not a sub-module / parameter)
        return ro.generate_frames(start, cps, mask, infer_steps)[1:]
    if dtype not in ("fp32", "f32", "float32"):
        raise ValueError(f"--dtype {dtype} is built for the FNO rollout only")
    with torch.no_grad():
        return model.generate_many(inputs=start, case_params=cps, mask=mask, steps=infer_steps)


def case_metric_sums(model, all_features: List[Tensor], all_case_params: List[Tensor], infer_steps: int,
                     dtype: str = "fp32") -> np.ndarray:
    """(infer_steps, 3) fp64: for every step the SUM over the given cases of the per-case (mse, nmse, mae) -- the additive form
    of test_multistep.py:160-174, so shards combine with one all-reduce.  all_features[c]: (>=infer_steps, c+1, h, w) on the
    device.  The cases run as one batch unless the model is batch-dependent in its current mode (see the module docstring),
    in which case they run one by one exactly as in the reference."""
    n_cases = len(all_features)
    if n_cases == 0:
        return np.zeros((infer_steps, 3), dtype=np.float64)
    start = torch.stack([f[0, :-1] for f in all_features])          # (n, c, h, w)
    mask = torch.stack([f[0, -1] for f in all_features])            # (n, h, w)
    cps = torch.stack(list(all_case_params))                        # (n, p)
    if batch_dependent(model):
        per_case = [infer_case(model, f, cp, infer_steps) for f, cp in zip(all_features, all_case_params)]
        n_frames = len(per_case[0])
        preds = [torch.cat([pc[k] for pc in per_case], dim=0) for k in range(n_frames)]
    else:
        preds = rollout_frames(model, start, cps, mask, infer_steps, dtype)
    sums = torch.empty(infer_steps, n_cases, 3, device=start.device)
    for step in range(infer_steps):
        lab = torch.stack([f[step, 0] * f[step, -1] for f in all_features])   # u channel of label frame `step`, masked
        msk = torch.stack([f[step, -1] for f in all_features])
        p = preds[step][:, 0] * msk
        d = p - lab
        n = float(d[0].numel())
        sums[step, :, 0] = (d * d).flatten(1).sum(1) / n
        sums[step, :, 1] = (lab * lab).flatten(1).sum(1) / n
        sums[step, :, 2] = d.abs().flatten(1).sum(1) / n
    s = sums.double().cpu().numpy()  # the ONE device-to-host transfer of the rollout
    return np.stack([s[:, :, 0].sum(1), (s[:, :, 0] / s[:, :, 1]).sum(1), s[:, :, 2].sum(1)], axis=1)


def shard_cases(n_cases: int, rank: int, world: int) -> List[int]:
    """Cases owned by ``rank``: r, r + world, ... (round robin keeps long and short cases spread over the ranks)."""
    return list(range(rank, n_cases, world))


def infer(model, all_features: List[Tensor], all_case_params: List[Tensor], infer_steps: int,
          dtype: str = "fp32", n_total_cases: int | None = None) -> List[Dict[str, float]]:
    """test_multistep.py:135-177: per step, the mean over ALL test cases of the per-case metrics.  In a multi-process launch
    ``all_features`` holds this rank's shard (``shard_cases``) and ``n_total_cases`` the global count; the shards' sums meet in
    one all-reduce (the only collective of the rollout)."""
    import torch.distributed as dist
    local = case_metric_sums(model, all_features, all_case_params, infer_steps, dtype)
    n = len(all_features) if n_total_cases is None else int(n_total_cases)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = all_features[0].device if len(all_features) else (torch.device("cuda", torch.cuda.current_device())
                                                                if torch.cuda.is_available() else torch.device("cpu"))
        if dist.get_backend() == "gloo":
            dev = torch.device("cpu")
        t = torch.cat([torch.from_numpy(local).flatten(), torch.tensor([float(len(all_features))], dtype=torch.float64)]).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu().numpy()
        assert int(round(t[-1])) == n, f"the ranks hold {int(round(t[-1]))} cases, expected {n}"
        local = t[:-1].reshape(infer_steps, 3)
    return [dict(mse=float(local[k, 0] / n), nmse=float(local[k, 1] / n), mae=float(local[k, 2] / n)) for k in range(infer_steps)]


def prepare_cases(test_data, infer_steps: int, device="cuda", only: List[int] | None = None):
    """Pad every case to >= infer_steps frames by repeating the last frame (steady state), move to the device
    (test_multistep.py:201-218).  ``only``: indices of the cases to prepare (this rank's shard)."""
    feats, cps = [], []
    for ci, (case_features, case_params) in enumerate(zip(test_data.all_features, test_data.case_params)):
        if only is not None and ci not in only:
            continue
        case_features = np.asarray(case_features)
        if case_features.shape[0] < infer_steps:
            pad = np.repeat(case_features[-1:], infer_steps - case_features.shape[0], axis=0)
            case_features = np.concatenate([case_features, pad], axis=0)
        feats.append(torch.as_tensor(case_features, dtype=torch.float32).to(device))
        cps.append(case_params_to_tensor(case_params).to(device))
    return feats, cps


def main(argv=None):
    from .data import get_auto_dataset
    from .dist_util import init_distributed
    args = Args().parse_args(argv)
    is_args_valid(args)
    rank, world = init_distributed()  # one process per GPU; a plain start stays single-process
    if rank == 0:
        print(args)
    _, _, test_data = get_auto_dataset(data_dir=Path(args.data_dir), data_name=args.data_name, delta_time=args.delta_time,
                                       norm_props=bool(args.norm_props), norm_bc=bool(args.norm_bc), load_splits=["test"])
    n_cases = len(test_data.all_features)
    mine = shard_cases(n_cases, rank, world)
    feats, cps = prepare_cases(test_data, args.infer_steps, only=set(mine))
    model = init_model(args).cuda()
    output_dir = get_output_dir(args, is_auto=True)
    load_best_ckpt(model, output_dir)  # every rank reads rank 0's checkpoint tree: identical replicas, no broadcast needed
    all_metrics = infer(model, feats, cps, args.infer_steps, dtype=args.dtype, n_total_cases=n_cases)
    if rank == 0:
        dump_json(all_metrics, output_dir / "multistep_metrics.json")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
