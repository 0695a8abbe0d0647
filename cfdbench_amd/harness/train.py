"""Non-autoregressive train / eval / test loop (DeepONet, FfnModel) with the reference's behaviour and artefacts
(src/train.py:26-350).

    python -m cfdbench_amd.harness.train --model deeponet --data cavity_prop_bc_geo --loss_name nmse

Dataset items are ``(case_params (p,), t (1,), frame (c,h,w))`` (src/dataset/cavity.py:198-205).  Training queries
``num_label_samples`` random points per frame inside the model (deeponet.py:187-197); evaluation reconstructs whole
frames with ``generate_one`` and scores the u channel.  Files written: ckpt-{ep}/{model.pt, dev_loss.json,
train_loss.json, scores.json}, train_losses.json, test/{preds.pt, scores.json}, args.json -- the names of src/train.py.
Under torch.distributed each rank trains on its shard of the frames; gradients are summed over ranks in one flat
all-reduce; rank 0 alone evaluates and writes files."""
from __future__ import annotations

import time
from pathlib import Path
from shutil import copyfile
from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch.optim import Adam
from torch.utils.data import DataLoader, Dataset, Subset

from ..engine import sync_gradients
from ..models.base_model import CfdModel
from ..models.deeponet import DeepONet
from ..models.ffn import FfnModel
from ..models.loss import loss_name_to_fn
from .args import Args, is_args_valid
from .common import dump_json, get_output_dir, load_best_ckpt, plot_loss, plot_predictions
from .schedule import EarlyStopping, LrSchedule
from .dist_util import (average_buffers, broadcast_model_state, check_resume_state, init_distributed, rank_world,
                        shard_indices)


def collate_fn(batch: list, device: Optional[str] = "cuda"):
    """list of (case_params (p,), t (1,), frame (c,h,w)) -> kwargs of the model's forward (src/train.py:26-35)."""
    case_params, t, label = zip(*batch)
    out = dict(case_params=torch.stack(case_params), t=torch.stack(t), label=torch.stack(label))
    if device is not None:
        out = {k: v.to(device, non_blocking=True).contiguous() for k, v in out.items()}
    return out


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def evaluate(model: CfdModel, data, output_dir: Path, batch_size: int = 64, plot_interval: int = 1,
             measure_time: bool = False) -> Dict[str, Any]:
    """src/train.py:38-113: whole-frame reconstruction with ``generate_one``; scores of the u channel; predictions
    repeated to three channels for the plotting / preds.pt layout of the reference."""
    loader = DataLoader(data, batch_size=batch_size, shuffle=False, collate_fn=collate_fn)
    scores: Dict[str, List[float]] = {name: [] for name in model.loss_fn.get_score_names()}
    all_preds = []
    start_time = time.time()
    model.eval()
    with torch.no_grad():
        for step, batch in enumerate(loader):
            label = batch["label"]
            height, width = label.shape[-2:]
            preds = model.generate_one(case_params=batch["case_params"], t=batch["t"], height=height, width=width)
            loss = model.loss_fn(labels=label[:, :1], preds=preds)
            for key in scores:
                scores[key].append(loss[key].detach().reshape(()))  # stays on the device: one transfer below
            preds = preds.repeat(1, 3, 1, 1)
            all_preds.append(preds.detach())
            if plot_interval > 0 and step % plot_interval == 0 and not measure_time:
                plot_predictions(inp=None, label=label[0][0], pred=preds[0][0], out_dir=Path(output_dir) / "images",
                                 step=step)
    for key in scores:
        scores[key] = torch.stack(scores[key]).cpu().tolist() if scores[key] else []
    all_preds = [p_.cpu() for p_ in all_preds]  # same list-of-batches layout as the reference's preds.pt (src/train.py:143)
    if measure_time:
        print(f"Time per step: {1000 * (time.time() - start_time) / max(len(loader), 1):.3f} ms")
    avg_scores = {key: float(np.mean(vals)) for key, vals in scores.items()}
    if "nmse" in scores:
        plot_loss(scores["nmse"], Path(output_dir) / "loss.png")
    return dict(scores=dict(mean=avg_scores, all=scores), preds=all_preds)


def test(model: CfdModel, data, output_dir: Path, plot_interval: int = 10, batch_size: int = 1,
         measure_time: bool = False):
    """src/train.py:116-145."""
    assert plot_interval > 0
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    result = evaluate(model, data, output_dir, batch_size=batch_size, plot_interval=plot_interval,
                      measure_time=measure_time)
    torch.save(result["preds"], output_dir / "preds.pt")
    dump_json(result["scores"], output_dir / "scores.json")
    return result


def train(model: CfdModel, train_data, dev_data, output_dir: Path, num_epochs: int = 400, lr: float = 1e-3,
          lr_step_size: int = 1, lr_gamma: float = 0.9, batch_size: int = 64, log_interval: int = 50,
          eval_interval: int = 2, measure_time: bool = False, plot_interval: int = 1, resume: bool = False,
          lr_scheduler_kind: str = "step", lr_scheduler_factor: float = 0.5, lr_scheduler_patience: int = 5,
          early_stopping_patience: int = 0, early_stopping_delta: float = 1e-5, gradient_accumulation_steps: int = 1,
          graph: bool = False):
    """src/train.py:148-253: fwd -> ``loss["nmse"].backward()`` -> Adam -> zero_grad; StepLR per epoch.
    ``graph``: the step replayed from HIP graphs (harness/train_auto.py:train has the semantics; no accumulation; several ranks: graph.py).
    ``resume``: continue from ``train_state.pt`` (see harness/train_auto.py:train).  ``lr_scheduler_kind`` /
    ``early_stopping_patience`` / ``gradient_accumulation_steps``: the same options, with the same semantics, as
    harness/train_auto.py:train (harness/schedule.py); the defaults are the reference's loop."""
    rank, world = _rank_world()
    output_dir = Path(output_dir)
    if world > 1:
        broadcast_model_state(model)

    def make_loader(ep: int):  # equal shards, re-partitioned every epoch: every rank runs the same number of steps
        data = Subset(train_data, shard_indices(len(train_data), rank, world, batch_size, epoch=ep)) if world > 1 else train_data
        return DataLoader(data, batch_size=batch_size, collate_fn=collate_fn, shuffle=True, drop_last=world > 1)

    loader = make_loader(0)
    if rank == 0:
        output_dir.mkdir(exist_ok=True, parents=True)
    accum = max(1, int(gradient_accumulation_steps))
    if graph:
        if accum > 1:
            raise NotImplementedError("--graph 1 needs gradient_accumulation_steps == 1")
        # fused: ONE multi-tensor kernel per step (the capturable foreach form with a device-resident rate takes ~2.7 ms for the U-Net's
        # 136 tensors, tools/exp/adam_fused_ab.py); same formula, results within rounding of the foreach form
        from ..optim import Adam as MultiTensorAdam
        if MultiTensorAdam.supports(list(model.parameters())):  # one launch per 80 tensors (cfd_adam_multi; torch's fused form: 3 x 31 us)
            optimizer = MultiTensorAdam(model.parameters(), lr=torch.tensor(float(lr), device="cuda"))
        else:
            optimizer = Adam(model.parameters(), lr=torch.tensor(float(lr), device="cuda"), capturable=True,
                             fused=not any(p_.is_complex() for p_ in model.parameters()))  # (torch's fused form takes real parameters only)
    else:
        optimizer = Adam(model.parameters(), lr=lr)
    graphed = None
    schedule = LrSchedule(lr_scheduler_kind, lr, num_epochs, optimizer, lr_step_size=lr_step_size, lr_gamma=lr_gamma,
                          factor=lr_scheduler_factor, patience=lr_scheduler_patience)
    stopper = EarlyStopping(early_stopping_patience, early_stopping_delta)
    start_time = time.time()
    global_step = 0
    all_train_losses: List[float] = []
    start_ep = 0
    state_path = output_dir / "train_state.pt"
    if resume and state_path.exists():
        state = torch.load(state_path, map_location="cpu", weights_only=False)
        check_resume_state(state, fused=False, world=world)
        model.load_state_dict(torch.load(output_dir / state["ckpt"] / "model.pt", map_location="cpu"))
        optimizer.load_state_dict(state["optimizer"])
        if graph:  # the rate of a capturable Adam is a device tensor
            for g_ in optimizer.param_groups:
                g_["lr"] = torch.tensor(float(g_["lr"]), device="cuda")
        if "sched" in state["scheduler"]:
            schedule.load_state_dict(state["scheduler"])
        else:  # format 1 (round 2): the bare StepLR state
            schedule.sched.load_state_dict(state["scheduler"])
        if state.get("early_stopping") is not None:
            stopper.load_state_dict(state["early_stopping"])
        start_ep, global_step, all_train_losses = state["ep"] + 1, state["global_step"], list(state["train_losses"])
        torch.set_rng_state(state["rng"])
        if rank == 0:
            print(f"resuming after epoch {state['ep']} (step {global_step}) from {state_path}")
    for ep in range(start_ep, num_epochs):
        ep_start_time = time.time()
        ep_train_losses: List[float] = []
        model.train()
        if world > 1 and ep > 0:
            loader = make_loader(ep)
        n_steps = len(loader)
        n_full = (n_steps // accum) * accum
        for step, batch in enumerate(loader):
            if graph:
                from ..graph import GraphedTrainStep
                if graphed is None and step + 1 < n_steps:  # capture on a full-size batch
                    graphed = GraphedTrainStep(model, optimizer, batch, "nmse", restore_state=True)
                if graphed is not None and graphed.matches(batch):
                    loss = graphed(**batch)["nmse"]
                elif graphed is not None:  # a batch of another shape: the same step (and gradient exchange), eagerly
                    loss = graphed.eager_step(batch)["nmse"]
                else:
                    if world > 1:
                        raise NotImplementedError("--graph 1 with several ranks needs at least two batches per epoch and rank")
                    optimizer.zero_grad(set_to_none=False)
                    loss = model(**batch)["loss"]["nmse"]
                    loss.backward()
                    optimizer.step()
                ep_train_losses.append(loss.detach().clone())  # fetched once per epoch
                global_step += 1
                continue
            loss = model(**batch)["loss"]["nmse"]
            group = accum if step < n_full else n_steps - n_full  # the epoch's trailing group may be shorter
            (loss / group if group > 1 else loss).backward()
            if (step + 1) % accum == 0 or step + 1 == n_steps:
                if world > 1:
                    sync_gradients(list(model.parameters()))
                optimizer.step()
                optimizer.zero_grad()
            ep_train_losses.append(loss.item())  # src/train.py:203
            global_step += 1
            if global_step % log_interval == 0 and not measure_time and rank == 0:
                avg_loss = sum(ep_train_losses) / (len(ep_train_losses) + 1e-5)
                print(dict(ep=ep, step=step, loss=f"{avg_loss:.3e}", lr=f"{schedule.lr:.3e}",
                           time=round(time.time() - start_time)))
        if graph:
            ep_train_losses = torch.stack(ep_train_losses).tolist() if ep_train_losses else []
        if measure_time:
            print("Time usage:", time.time() - ep_start_time)
            return all_train_losses + ep_train_losses
        schedule.epoch_end()
        if world > 1 and (ep + 1) % eval_interval == 0:
            average_buffers(model)
        stop = False
        if (ep + 1) % eval_interval == 0 and rank == 0:
            ckpt_dir = output_dir / f"ckpt-{ep}"
            ckpt_dir.mkdir(exist_ok=True, parents=True)
            dev_scores = evaluate(model, dev_data, ckpt_dir, plot_interval=plot_interval)["scores"]
            dump_json(dev_scores, ckpt_dir / "dev_loss.json")
            dump_json(ep_train_losses, ckpt_dir / "train_loss.json")
            ckpt_path = ckpt_dir / "model.pt"
            if ckpt_path.exists():
                copyfile(ckpt_path, ckpt_dir / "backup_model.pt")
            torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, ckpt_path)
            dev_loss = float(np.mean(dev_scores["mean"]["nmse"]))
            dump_json(dict(ep=ep, train_loss=float(np.mean(ep_train_losses)), dev_loss=dev_loss,
                           time=time.time() - ep_start_time), ckpt_dir / "scores.json")
            schedule.validation(dev_loss)
            stop = stopper.update(dev_loss)
            tmp = output_dir / "train_state.pt.tmp"
            torch.save(dict(format=2, ep=ep, global_step=global_step, train_losses=all_train_losses + ep_train_losses,
                            ckpt=ckpt_dir.name, optimizer=optimizer.state_dict(), scheduler=schedule.state_dict(),
                            early_stopping=stopper.state_dict(), rng=torch.get_rng_state(), world=world), tmp)
            tmp.replace(state_path)
        if world > 1:  # every rank follows rank 0's validation-driven decisions (plateau rate, early stop)
            flags = torch.tensor([schedule.lr, float(stop)], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
            dist.broadcast(flags, 0)
            for g_ in optimizer.param_groups:
                g_["lr"] = float(flags[0])
            stop = bool(flags[1] > 0.5)
            dist.barrier()
        all_train_losses += ep_train_losses
        if stop:
            if rank == 0:
                print(f"early stopping after epoch {ep}: no improvement of {early_stopping_delta} in {early_stopping_patience} evaluations")
            break
    if rank == 0:
        dump_json(all_train_losses, output_dir / "train_losses.json")
        plot_loss(all_train_losses, output_dir / "train_losses.png")
    return all_train_losses


def init_model(args: Args) -> CfdModel:
    """Instantiate a non-autoregressive model (src/train.py:256-291): branch input = the case parameters
    (8 for the cylinder problem, else 5), trunk / query input = (t, x, y)."""
    loss_fn = loss_name_to_fn(args.loss_name)
    query_coord_dim = 3
    n_case_params = 8 if "cylinder" in args.data_name else 5
    if args.model == "deeponet":
        return DeepONet(branch_dim=n_case_params, trunk_dim=query_coord_dim, loss_fn=loss_fn, width=args.deeponet_width,
                        trunk_depth=args.trunk_depth, branch_depth=args.branch_depth, act_name=args.act_fn,
                        act_norm=bool(args.act_scale_invariant), act_on_output=bool(args.act_on_output))
    if args.model == "ffn":
        widths = [n_case_params + query_coord_dim] + [args.ffn_width] * args.ffn_depth + [1]
        return FfnModel(widths=widths, loss_fn=loss_fn)
    raise ValueError(f"Invalid model name: {args.model}")


class SyntheticDataset(Dataset):
    """Items of the reference's non-autoregressive shape, built from the frames of a ``SyntheticAutoDataset``:
    (case_params (5,), t (1,), frame (3,h,w)) -- for smoke runs and tests without data on disk."""

    def __init__(self, n_cases: int = 3, n_frames: int = 4, height: int = 16, width: int = 16, seed: int = 0):
        from .data import SyntheticAutoDataset
        src = SyntheticAutoDataset(n_cases, n_frames, height, width, seed)
        self.items = []
        for feats, cp in zip(src.all_features, src.case_params):
            cpt = torch.tensor(list(cp.values()), dtype=torch.float32)
            for t in range(feats.shape[0]):
                self.items.append((cpt, torch.tensor([float(t)]), torch.from_numpy(feats[t])))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def get_dataset(data_dir: Path, data_name: str, norm_props: bool, norm_bc: bool):
    """(train, dev, test) CfdDatasets (src/dataset/__init__.py:12) through the native loaders (harness/flow_data.py)."""
    problem = data_name.split("_")[0]
    if problem in ("cavity", "tube", "dam", "cylinder"):
        from .flow_data import get_flow_datasets
        return get_flow_datasets(problem, Path(data_dir) / problem, data_name[len(problem) + 1:], norm_props=norm_props,
                                 norm_bc=norm_bc)
    raise ValueError(f"Invalid data name {data_name}!")  # src/dataset/__init__.py:60


def main(argv=None):
    args = Args().parse_args(argv)
    is_args_valid(args)
    if args.dtype != "fp32":
        raise NotImplementedError("--dtype bf16 is an inference option (test_multistep); training stores fp32")
    rank, world = init_distributed()  # one process per GPU under torch.distributed.run; (0, 1) otherwise
    output_dir = get_output_dir(args)
    if rank == 0:
        print(args)
        output_dir.mkdir(exist_ok=True, parents=True)
        args.save(str(output_dir / "args.json"))
    train_data, dev_data, test_data = get_dataset(Path(args.data_dir), args.data_name, bool(args.norm_props),
                                                  bool(args.norm_bc))
    model = init_model(args).cuda()
    if rank == 0:
        print(f"Model has {sum(p.numel() for p in model.parameters())} parameters")
    if "train" in args.mode:
        if rank == 0:
            args.save(str(output_dir / "train_args.json"))
        train(model, train_data, dev_data, output_dir, batch_size=args.batch_size, lr=args.lr,
              lr_step_size=args.lr_step_size, lr_gamma=args.lr_gamma, num_epochs=args.num_epochs,
              eval_interval=args.eval_interval, log_interval=args.log_interval, plot_interval=args.plot_interval,
              resume=bool(args.resume), lr_scheduler_kind=args.lr_scheduler, lr_scheduler_factor=args.lr_scheduler_factor,
              lr_scheduler_patience=args.lr_scheduler_patience,
              early_stopping_patience=args.early_stopping_patience if args.early_stop else 0,
              early_stopping_delta=args.early_stopping_delta, gradient_accumulation_steps=args.gradient_accumulation_steps,
              graph=bool(args.graph))
    if "test" in args.mode and rank == 0:
        args.save(str(output_dir / "test_args.json"))
        load_best_ckpt(model, output_dir)
        test(model, data=test_data, output_dir=output_dir / "test", batch_size=1, plot_interval=10)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
