"""Autoregressive train / eval / test loop with the reference's behaviour and artefacts (src/train_auto.py:33-381).

    python -m cfdbench_amd.harness.train_auto --model fno --data cavity_prop_bc_geo --loss_name nmse --fno_hidden_dim 20

Two training paths:
  * default -- the reference's own sequence ``model(**batch)`` -> ``loss["nmse"].backward()`` -> ``Adam.step()``
    (train_auto.py:231-257) on the drop-in ``Fno2d`` (one autograd node over the HIP kernels);
  * ``--fused 1`` -- ``FnoTrainEngine``: forward + nMSE + backward + flat Adam as C-ABI calls without autograd and
    without the per-step ``.item()`` host sync (scores are read every ``log_interval`` steps only).
Both use Adam(lr) + StepLR(lr_step_size, gamma) per epoch and write the same files: ckpt-{ep}/{model.pt,
dev_scores.json, train_loss.json, scores.json}, train_losses.json, test/{preds.pt, scores.json}, args.json.
Under torch.distributed (one process per GPU) each rank trains on its shard of the frames, gradients are summed over
ranks (engine path), and rank 0 alone writes files.
"""
from __future__ import annotations

import time
from copy import deepcopy
from pathlib import Path
from shutil import copyfile
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor
from torch.optim import Adam, lr_scheduler
from torch.utils.data import DataLoader, Subset

from ..engine import FnoTrainEngine, sync_gradients
from ..models.base_model import AutoCfdModel
from ..models.fno.fno2d import Fno2d
from .args import Args, is_args_valid
from .autoregressive import init_model
from .common import dump_json, get_output_dir, load_best_ckpt, plot, plot_loss, plot_predictions
from .schedule import EarlyStopping, LrSchedule
from .dist_util import (average_buffers, broadcast_model_state, check_resume_state, init_distributed, rank_world,
                        shard_indices)


def collate_fn(batch: list, device: Optional[str] = "cuda"):
    """list of (input (3,h,w), label (3,h,w), case_params dict) -> kwargs of the model's forward (train_auto.py:33-58):
    the last channel is the mask; case_params = all case.json keys except rotated/dx/dy, in key order."""
    inputs, labels, case_params = zip(*batch)
    inputs = torch.stack(inputs)
    labels = torch.stack(labels)
    labels = labels[:, :-1]
    mask = inputs[:, -1:]
    inputs = inputs[:, :-1]
    keys = [x for x in case_params[0].keys() if x not in ["rotated", "dx", "dy"]]
    case_params_t = torch.tensor([[cp[k] for k in keys] for cp in case_params], dtype=torch.float32)
    out = dict(inputs=inputs, label=labels, mask=mask, case_params=case_params_t)
    if device is not None:
        out = {k: v.to(device, non_blocking=True).contiguous() for k, v in out.items()}
    return out


_rank_world = rank_world


def evaluate(model: AutoCfdModel, data, output_dir: Path, batch_size: int = 2, plot_interval: int = 1,
             measure_time: bool = False, device_loader: bool = False, sharded: bool = False):
    """Single-step evaluation (train_auto.py:61-148): identity baseline + model scores per batch, predictions.

    ``sharded`` (data-parallel runs; every rank must call): batch k of the evaluation order is evaluated by rank k % world -- the
    SAME batches as the single-process loop, so the per-batch scores and the predictions are identical -- and rank 0 gathers
    them back into order (one ``gather_object`` of the ranks' score tables and prediction blocks).  Rank 0 returns the result,
    the other ranks None.  The reference evaluates on one process; with the training step at > 200 k frames/s the evaluation
    every ``eval_interval`` epochs was the serial part of a multi-GPU run."""
    rank, world = _rank_world() if sharded else (0, 1)
    n_batches = (len(data) + batch_size - 1) // batch_size
    mine = list(range(rank, n_batches, world))
    own = None
    if world > 1:  # the frames of this rank's batches, batch after batch (a short last batch is the last of its owner)
        own = [i for k in mine for i in range(k * batch_size, min((k + 1) * batch_size, len(data)))]
    if device_loader:  # same batches, gathered on the device (harness/data.py)
        from .data import DeviceBatchLoader
        loader = DeviceBatchLoader(data, batch_size, shuffle=False, indices=own)
    else:
        dev = "cuda" if torch.cuda.is_available() else None
        loader = DataLoader(data if own is None else Subset(data, own), batch_size=batch_size, shuffle=False,
                            collate_fn=lambda b: collate_fn(b, device=dev))
    scores = {name: [] for name in model.loss_fn.get_score_names()}
    input_scores = deepcopy(scores)
    all_preds: List[Tensor] = []
    start_time = time.time()
    model.eval()
    # the reference moves every score and every prediction to the host per batch (train_auto.py:96,
    # :110-113), i.e. two device synchronisations per batch of 16.  Here the per-batch scores and predictions stay on the
    # device and cross PCIe once at the end; the returned lists / tensors are the same.
    with torch.inference_mode():
        for step, batch in enumerate(loader):
            inputs, labels = batch["inputs"], batch["label"]
            input_loss = model.loss_fn(labels=labels[:, :1], preds=inputs[:, :1])  # train_auto.py:93
            for key in input_scores:
                input_scores[key].append(input_loss[key].detach().reshape(()))
            outputs = model(**batch)
            loss, preds = outputs["loss"], outputs["preds"]
            height, width = labels.shape[2:]
            preds = preds.view(-1, 1, height, width)  # train_auto.py:106
            for key in scores:
                scores[key].append(loss[key].detach().reshape(()))
            all_preds.append(preds.detach())
            gstep = mine[step] if world > 1 else step  # position in the evaluation order
            # every rank plots the batches it owns (the file names carry the GLOBAL step, all ranks write into the same images/
            # directory): the output tree of an N-rank run is the single-process one
            if plot_interval > 0 and gstep % plot_interval == 0 and not measure_time:
                plot_predictions(inp=inputs[0][0], label=labels[0][0], pred=preds[0][0], out_dir=Path(output_dir) / "images",
                                 step=gstep)
    for table in (scores, input_scores):
        for key in table:
            table[key] = torch.stack(table[key]).cpu().tolist() if table[key] else []
    # all predictions of this rank to the host with ONE synchronisation: every batch is copied asynchronously into its slice of one
    # pinned host buffer (a device-side torch.cat first would hold the split's predictions twice on the device: ADVICE r4)
    sizes = [int(p_.shape[0]) for p_ in all_preds]
    pred_blocks = []
    if all_preds:
        first = all_preds[0]
        shape = (sum(sizes),) + tuple(first.shape[1:])
        nbytes = first.element_size() * int(torch.tensor(shape).prod())
        # pinned only while the split is small (<= 256 MB): a multi-GB hipHostMalloc is slow, can fail where pageable memory would
        # not, and stays pinned in the caching host allocator afterwards; pageable copies are synchronous but correct
        pin = first.is_cuda and nbytes <= (256 << 20)
        try:
            host = torch.empty(shape, dtype=first.dtype, pin_memory=pin)
        except RuntimeError:
            host, pin = torch.empty(shape, dtype=first.dtype), False
        off = 0
        for p_, n_ in zip(all_preds, sizes):
            host[off:off + n_].copy_(p_, non_blocking=pin)
            off += n_
        if first.is_cuda:
            torch.cuda.current_stream().synchronize()
        # (clones under a process group: gather_object pickles every view together with the WHOLE storage it points into)
        pred_blocks = [b_.clone() for b_ in host.split(sizes)] if world > 1 else list(host.split(sizes))
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((mine, scores, input_scores, pred_blocks), gathered, dst=0)
        if rank != 0:
            return None
        order = {}
        for r_mine, r_scores, r_in, r_preds in gathered:
            for pos, k in enumerate(r_mine):
                order[k] = ({key: r_scores[key][pos] for key in r_scores}, {key: r_in[key][pos] for key in r_in}, r_preds[pos])
        assert sorted(order) == list(range(n_batches)), "the ranks' evaluation shards do not tile the batches"
        scores = {key: [order[k][0][key] for k in range(n_batches)] for key in scores}
        input_scores = {key: [order[k][1][key] for k in range(n_batches)] for key in input_scores}
        pred_blocks = [order[k][2] for k in range(n_batches)]
    all_preds = [torch.cat(pred_blocks, dim=0)] if pred_blocks else []
    if measure_time:
        print(f"Time (ms) per step: {1000 * (time.time() - start_time) / max(len(loader), 1):.3f}")
    avg_scores = {}
    for key in scores:
        avg_scores[key] = float(np.mean(scores[key]))
        avg_scores[f"input_{key}"] = float(np.mean(input_scores[key]))
    if "nmse" in scores:
        plot_loss(scores["nmse"], Path(output_dir) / "loss.png")
    return dict(preds=torch.cat(all_preds, dim=0), scores=dict(mean=avg_scores, all=scores))


def test(model: AutoCfdModel, data, output_dir: Path, infer_steps: int = 200, plot_interval: int = 10,
         batch_size: int = 1, measure_time: bool = False):
    """train_auto.py:151-178: evaluate on the test split, write preds.pt and scores.json."""
    assert infer_steps > 0 and plot_interval > 0
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True, parents=True)
    result = evaluate(model, data, output_dir=output_dir, batch_size=batch_size, plot_interval=plot_interval,
                      measure_time=measure_time)
    torch.save(result["preds"], output_dir / "preds.pt")
    dump_json(result["scores"], output_dir / "scores.json")
    return result


def train(model: AutoCfdModel, train_data, dev_data, output_dir: Path, num_epochs: int = 400, lr: float = 1e-3,
          lr_step_size: int = 1, lr_gamma: float = 0.9, batch_size: int = 2, eval_batch_size: int = 2,
          log_interval: int = 10, eval_interval: int = 2, measure_time: bool = False, fused: bool = False,
          plot_interval: int = 1, resume: bool = False, device_loader: bool = False, lr_scheduler_kind: str = "step",
          lr_scheduler_factor: float = 0.5, lr_scheduler_patience: int = 5, early_stopping_patience: int = 0,
          early_stopping_delta: float = 1e-5, gradient_accumulation_steps: int = 1, act_dtype: str = "fp32", graph: bool = False):
    """train_auto.py:181-313.  ``fused`` selects FnoTrainEngine (needs an Fno2d and the nmse loss).

    ``graph`` (autograd path; with several ranks the gradient exchange sits between two graphs, cfdbench_amd/graph.py): the step ``model(**batch) -> loss["nmse"].backward() -> Adam.step()`` is captured once
    as a HIP graph (cfdbench_amd/graph.py) at the first full batch and replayed for every batch of that shape; a short last batch
    runs eagerly on the same (capturable) optimiser.  Per-step losses stay on the device and are fetched once per epoch.  The
    kernels are the same; what goes away is the per-operator host work of the eager loop (U-Net dim 12, B = 128: 5.2 -> 4.0 ms per
    step; Auto-DeepONet B = 512: 1.8 -> 0.8 ms).  Adam runs in its capturable form (step count and rate on the device, fp32), so a
    graph run follows an eager run to rounding, not to the bit.

    ``resume`` (the reference saves weights only, train_auto.py:301, and cannot continue a run): every
    checkpoint epoch also writes ``train_state.pt`` (optimiser moments / step, LR schedule, epoch, loss history, host RNG
    state); with ``resume`` a run that finds it reloads the weights of that checkpoint and continues with the NEXT epoch,
    reproducing the uninterrupted run step for step (same shuffles, same Adam state).

    ``lr_scheduler_kind`` / ``early_stopping_patience`` / ``gradient_accumulation_steps``: the training options of this fork's
    other trainers (harness/schedule.py; src/args.py:53-56,77-80,323) -- the defaults are the reference's benchmark loop."""
    rank, world = _rank_world()
    output_dir = Path(output_dir)
    if world > 1:
        broadcast_model_state(model)  # identical replicas (weights, BatchNorm statistics) before the first step

    def make_loader(ep: int):
        """The epoch's loader.  With several ranks the frames are re-partitioned EVERY epoch (``DistributedSampler.set_epoch``
        semantics: shard_indices(..., epoch=ep) -- an equal part of that epoch's permutation per rank), so a
        rank does not see the same 1/world of the data for the whole run."""
        shard = shard_indices(len(train_data), rank, world, batch_size, epoch=ep) if world > 1 else None
        if device_loader:  # frames resident in HBM, batches gathered on the device (harness/data.py)
            from .data import DeviceBatchLoader
            if resident:  # the split is uploaded ONCE; an epoch's re-partition only swaps the index list
                resident[0].set_indices(shard)
            else:
                resident.append(DeviceBatchLoader(train_data, batch_size, shuffle=True, drop_last=world > 1, indices=shard))
            return resident[0]
        data = Subset(train_data, shard) if world > 1 else train_data
        return DataLoader(data, batch_size=batch_size, shuffle=True, collate_fn=collate_fn, drop_last=world > 1)

    resident: List = []
    train_loader = make_loader(0)
    if rank == 0:
        output_dir.mkdir(exist_ok=True, parents=True)
    engine = None
    accum = max(1, int(gradient_accumulation_steps))
    if fused:
        if not isinstance(model, Fno2d):
            raise NotImplementedError("--fused 1 needs the fno model")
        if accum > 1:
            raise NotImplementedError("--gradient_accumulation_steps > 1 needs the autograd path (--fused 0): the engine's "
                                      "backward pass overwrites the flat gradient")
        engine = FnoTrainEngine(model, lr=lr, loss_name="nmse", act_dtype=act_dtype)
        optimizer = None
    elif graph:
        if accum > 1 or getattr(model, "graph_unsafe", False):
            raise NotImplementedError("--graph 1 needs gradient_accumulation_steps == 1 and a model without per-step host state")
        # world > 1 (round 5): graph.GraphedTrainStep replays forward + backward + gradient pack, exchanges the flat gradient over the
        # process group and replays the optimizer from a second graph
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
    train_losses: List[float] = []
    start_ep = 0
    state_path = output_dir / "train_state.pt"
    if resume and state_path.exists():
        state = torch.load(state_path, map_location="cpu", weights_only=False)
        check_resume_state(state, fused=engine is not None, world=world)
        model.load_state_dict(torch.load(output_dir / state["ckpt"] / "model.pt", map_location="cpu"))
        if hasattr(model, "load_extra_train_state") and state.get("model_extra") is not None:
            model.load_extra_train_state(state["model_extra"])  # e.g. ResNet's dropout step counter
        if engine is not None:
            engine.load_state_dict(state["optimizer"])
        else:
            optimizer.load_state_dict(state["optimizer"])
            if graph:  # the rate of a capturable Adam is a device tensor (load_state_dict brought the saved copy)
                for g_ in optimizer.param_groups:
                    g_["lr"] = torch.tensor(float(g_["lr"]), device="cuda")
        if state.get("scheduler") is None:
            # train_state.pt of a round-2 fused run (format 1: the engine carried the rate, no scheduler state was written):
            # replay the epoch-end steps of the schedule, which for step / cosine depends on the epoch count alone
            if schedule.kind == "plateau":
                raise RuntimeError("train_state.pt holds no scheduler state (written before format 2): a plateau schedule cannot be rebuilt")
            for _ in range(state["ep"] + 1):
                schedule.epoch_end()
        else:
            schedule.load_state_dict(state["scheduler"])
        if state.get("early_stopping") is not None:
            stopper.load_state_dict(state["early_stopping"])
        start_ep, global_step, train_losses = state["ep"] + 1, state["global_step"], list(state["train_losses"])
        torch.set_rng_state(state["rng"])  # the DataLoader draws its shuffles from the host generator
        if rank == 0:
            print(f"resuming after epoch {state['ep']} (step {global_step}) from {state_path}")
    for ep in range(start_ep, num_epochs):
        ep_start_time = time.time()
        cur_lr = schedule.lr
        ep_train_losses: List = []
        model.train()
        if world > 1 and ep > 0:
            train_loader = make_loader(ep)
        n_steps = len(train_loader)
        n_full = (n_steps // accum) * accum  # micro-batches in complete accumulation groups; the rest form one short group
        for step, batch in enumerate(train_loader):
            if engine is not None:
                engine.lr = cur_lr
                sums = engine.train_step(batch["inputs"], batch["label"], batch["case_params"], batch["mask"])
                # nmse = sum sq err / sum sq label; kept on the device, fetched once per epoch (no per-step host sync)
                ep_train_losses.append((sums[0] / sums[2]).clone())
                loss_mse = loss_nmse = None
            elif graph:
                from ..graph import GraphedTrainStep
                if graphed is None and step + 1 < n_steps:  # capture on a full-size batch (never on a short last one)
                    graphed = GraphedTrainStep(model, optimizer, batch, "nmse", restore_state=True)
                    if hasattr(train_loader, "bind"):  # the resident loader gathers the next batches straight into the step's buffers
                        train_loader.bind(graphed.static)
                if graphed is not None and graphed.matches(batch):
                    loss = graphed(**batch)
                    preds = graphed.preds
                elif graphed is not None:  # a batch of another shape: the same step (and exchange), eagerly
                    loss = graphed.eager_step(batch)
                    preds = graphed._eager_preds
                else:  # a one-batch epoch: nothing to capture on
                    if world > 1:
                        raise NotImplementedError("--graph 1 with several ranks needs at least two batches per epoch and rank")
                    optimizer.zero_grad(set_to_none=False)
                    outputs = model(**batch)
                    loss, preds = outputs["loss"], outputs["preds"]
                    loss["nmse"].backward()
                    optimizer.step()
                if step == 0 and not measure_time and rank == 0 and plot_interval > 0:
                    plot(batch["inputs"][0][0], batch["label"][0][0], preds[0][0].detach(), Path("example.png"))
                ep_train_losses.append(loss["nmse"].detach().clone())  # static tensor of the graph: cloned, fetched per epoch
                loss_mse, loss_nmse = loss["mse"], loss["nmse"]
            else:
                outputs = model(**batch)
                if step == 0 and not measure_time and rank == 0 and plot_interval > 0:
                    plot(batch["inputs"][0][0], batch["label"][0][0], outputs["preds"][0][0].detach(), Path("example.png"))
                loss = outputs["loss"]
                # mean over the micro-batches of THIS group: the trailing group of an epoch may be shorter than `accum`
                group = accum if step < n_full else n_steps - n_full
                (loss["nmse"] / group if group > 1 else loss["nmse"]).backward()  # train_auto.py:255
                if (step + 1) % accum == 0 or step + 1 == n_steps:
                    if world > 1:
                        sync_gradients(list(model.parameters()))  # one flat all-reduce, DDP semantics
                    optimizer.step()
                    optimizer.zero_grad()
                ep_train_losses.append(loss["nmse"].item())  # train_auto.py:260
                loss_mse, loss_nmse = loss["mse"], loss["nmse"]
            global_step += 1
            if global_step % log_interval == 0 and rank == 0:
                if engine is not None:
                    sc = engine.scores()
                    mse_v, nmse_v = sc["mse"], sc["nmse"]
                else:
                    mse_v, nmse_v = loss_mse.item(), loss_nmse.item()
                print(dict(ep=ep, step=step, mse=f"{mse_v:.3e}", nmse=f"{nmse_v:.3e}", lr=f"{cur_lr:.3e}",
                           time=round(time.time() - start_time)))
        if engine is not None or graph:
            ep_train_losses = torch.stack(ep_train_losses).tolist() if ep_train_losses else []
        schedule.epoch_end()
        if measure_time:
            print("Time usage:", time.time() - ep_start_time)
            return
        train_losses += ep_train_losses
        if world > 1 and (ep + 1) % eval_interval == 0:
            average_buffers(model)  # BatchNorm running statistics of all shards go into the checkpoint
        result = None
        if (ep + 1) % eval_interval == 0:  # every rank evaluates its share of the dev batches; rank 0 gets the assembled result
            ckpt_dir = output_dir / f"ckpt-{ep}"
            if rank == 0:
                ckpt_dir.mkdir(exist_ok=True, parents=True)
            result = evaluate(model, dev_data, ckpt_dir, batch_size=eval_batch_size, plot_interval=plot_interval,
                              device_loader=device_loader, sharded=world > 1)
        if (ep + 1) % eval_interval == 0 and rank == 0:
            dev_scores = result["scores"]
            dump_json(dev_scores, ckpt_dir / "dev_scores.json")
            dump_json(ep_train_losses, ckpt_dir / "train_loss.json")
            ckpt_path = ckpt_dir / "model.pt"
            if ckpt_path.exists():
                copyfile(ckpt_path, ckpt_dir / "backup_model.pt")
            # bare state_dict with the reference's key names (train_auto.py:301); cloned because under the fused engine
            # the parameters are float/complex views of ONE flat buffer, which torch.save refuses to serialise as is
            torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, ckpt_path)
            dump_json(dict(ep=ep, train_loss=float(np.mean(ep_train_losses)), dev_loss=float(np.mean(dev_scores["all"]["nmse"])),
                           time=time.time() - ep_start_time), ckpt_dir / "scores.json")
            # full training state next to the reference's artefacts (written last, atomically: a crash mid-checkpoint
            # leaves the previous state in place)
            dev_loss = float(np.mean(dev_scores["all"]["nmse"]))
            schedule.validation(dev_loss)
            stop = stopper.update(dev_loss)
            opt_state = engine.state_dict() if engine is not None else optimizer.state_dict()
            tmp = output_dir / "train_state.pt.tmp"
            torch.save(dict(format=2, ep=ep, global_step=global_step, train_losses=train_losses, ckpt=ckpt_dir.name,
                            optimizer=opt_state, scheduler=schedule.state_dict(), early_stopping=stopper.state_dict(),
                            rng=torch.get_rng_state(), fused=engine is not None, world=world,
                            model_extra=model.extra_train_state() if hasattr(model, "extra_train_state") else None), tmp)
            tmp.replace(state_path)
        else:
            stop = False
        if world > 1:  # every rank follows rank 0's validation-driven decisions (plateau rate, early stop)
            flags = torch.tensor([schedule.lr, float(stop)], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
            dist.broadcast(flags, 0)
            for g_ in schedule.optimizer.param_groups:
                g_["lr"] = float(flags[0])
            stop = bool(flags[1] > 0.5)
            dist.barrier()
        if stop:
            if rank == 0:
                print(f"early stopping after epoch {ep}: no improvement of {early_stopping_delta} in {early_stopping_patience} evaluations")
            break
    if rank == 0:
        dump_json(train_losses, output_dir / "train_losses.json")
        plot_loss(train_losses, output_dir / "train_losses.png")
    return train_losses


def main(argv=None):
    from .data import get_auto_dataset
    args = Args().parse_args(argv)
    is_args_valid(args)
    if args.dtype != "fp32" and not args.fused:
        raise NotImplementedError("--dtype bf16 training is the fused FNO engine's option (--fused 1): the autograd path stores fp32")
    rank, world = init_distributed()  # one process per GPU under torch.distributed.run; (0, 1) otherwise
    output_dir = get_output_dir(args, is_auto=True)
    if rank == 0:
        print("#" * 80)
        print(args)
        print("#" * 80)
        output_dir.mkdir(exist_ok=True, parents=True)
        args.save(str(output_dir / "args.json"))
    train_data, dev_data, test_data = get_auto_dataset(
        data_dir=Path(args.data_dir), data_name=args.data_name, delta_time=args.delta_time,
        norm_props=bool(args.norm_props), norm_bc=bool(args.norm_bc))
    model = init_model(args).cuda()  # the reference forgets .cuda() for FNO
    if rank == 0:
        print(f"Model has {sum(p.numel() for p in model.parameters())} parameters")
    if "train" in args.mode:
        if rank == 0:
            args.save(str(output_dir / "train_args.json"))
        train(model, train_data=train_data, dev_data=dev_data, output_dir=output_dir, lr=args.lr,
              lr_step_size=args.lr_step_size, lr_gamma=args.lr_gamma, num_epochs=args.num_epochs,
              batch_size=args.batch_size, eval_batch_size=args.eval_batch_size, eval_interval=args.eval_interval,
              log_interval=args.log_interval, fused=bool(args.fused), plot_interval=args.plot_interval,
              resume=bool(args.resume), device_loader=bool(args.device_loader), lr_scheduler_kind=args.lr_scheduler,
              lr_scheduler_factor=args.lr_scheduler_factor, lr_scheduler_patience=args.lr_scheduler_patience,
              early_stopping_patience=args.early_stopping_patience if args.early_stop else 0,
              early_stopping_delta=args.early_stopping_delta, gradient_accumulation_steps=args.gradient_accumulation_steps,
              act_dtype=args.dtype, graph=bool(args.graph))
    if "test" in args.mode and rank == 0:  # the test split is small: rank 0 evaluates it alone
        args.save(str(output_dir / "test_args.json"))
        load_best_ckpt(model, output_dir)
        test(model, test_data, output_dir / "test", batch_size=1, infer_steps=20, plot_interval=10)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
