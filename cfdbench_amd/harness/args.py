"""Command-line configuration with the reference's flag names and defaults (src/args.py:5-217).

The reference declares flags as annotated class attributes of a ``tap.Tap`` subclass; ``tap`` is not a dependency here, so
the same attribute table drives ``argparse``.  Differences, all documented in SURVEY.md section 0.1:
  * ``lr_step_size`` / ``lr_gamma`` exist (train_auto.py:357 reads ``args.lr_step_size`` but the reference's Args lacks it);
  * ``--data`` is an explicit alias of ``--data_name`` (README.md:179 uses it; argparse prefix matching would be ambiguous);
  * additions: ``infer_steps`` (test_multistep.py:198 hard-codes 20), ``fused`` (FnoTrainEngine instead of autograd+Adam),
    ``resume`` (continue from ``train_state.pt``: optimiser moments, schedule, epoch, RNG -- the reference saves weights only),
    ``device_loader`` (frames resident in HBM, batches gathered on the device instead of DataLoader + collate_fn),
    ``lr_scheduler`` (step = the reference's StepLR | plateau = ReduceLROnPlateau(lr_scheduler_factor, lr_scheduler_patience) |
    cosine), ``early_stop`` (1: stop after early_stopping_patience evaluations without early_stopping_delta improvement),
    ``gradient_accumulation_steps`` (args.py:323; micro-batches per optimiser step),
    ``graph`` (1: the autograd train step -- forward, loss, backward, Adam -- is captured once as a HIP graph and replayed per batch;
    single process, no gradient accumulation, models without per-step host state, i.e. not the ResNet's dropout),
    ``dtype`` ("bf16": the FNO's activations between kernels are stored as bf16 -- test_multistep: BASELINE configs[4];
    train_auto --fused 1: bf16-storage training with fp32 master weights, gradients and optimiser).
Flags of models that are not built yet are carried so existing command lines and args.json files keep working.
"""
from __future__ import annotations

import argparse
import json
from typing import Any, Dict, List, Optional

_FLAGS: Dict[str, Any] = dict(
    # 1. general (args.py:22-29)
    mode="train", seed=0, output_dir="result",
    # 2. training (args.py:37-84)
    lr=1e-4, weight_decay=1e-5, num_epochs=100, batch_size=8, eval_batch_size=16,
    lr_scheduler_factor=0.5, lr_scheduler_patience=5, loss_name="mse", log_interval=50, eval_interval=2,
    save_checkpoint_every_n_epochs=20, save_images_every_n_epochs=20, early_stopping_patience=20,
    early_stopping_delta=1e-5,
    # 3. dataset (args.py:88-112)
    data_name="cylinder_geo", data_dir="../data", num_rows=64, num_cols=64, delta_time=0.1, norm_props=1, norm_bc=1,
    # 4. model selection (args.py:119-136)
    model="pixel_diffusion", in_chan=2, out_chan=2,
    # 5. model hyper-parameters (args.py:143-217)
    ffn_depth=8, ffn_width=100, autoffn_depth=8, autoffn_width=200, deeponet_width=100, branch_depth=8, trunk_depth=8,
    act_fn="relu", act_scale_invariant=1, act_on_output=0, autoedeeponet_width=100, autoedeeponet_depth=8,
    autoedeeponet_act_fn="relu", fno_depth=4, fno_hidden_dim=32, fno_modes_x=12, fno_modes_y=12, unet_dim=12,
    unet_insert_case_params_at="input", resnet_depth=4, resnet_hidden_chan=16, resnet_kernel_size=7, resnet_padding=3,
    # missing in the reference's Args but read by its trainers (train_auto.py:357, :188-189)
    lr_step_size=20, lr_gamma=0.9,
    # additions of this harness
    infer_steps=20, fused=0, plot_interval=1, resume=0, device_loader=0, dtype="fp32", graph=0,
    lr_scheduler="step", early_stop=0, gradient_accumulation_steps=1,
)


class Args:
    """Attribute bag with ``parse_args`` / ``save`` / ``as_dict`` like the reference's Tap class."""

    def __init__(self, **overrides):
        for k, v in _FLAGS.items():
            setattr(self, k, v)
        for k, v in overrides.items():
            if k not in _FLAGS:
                raise AttributeError(f"unknown flag {k!r}")
            setattr(self, k, v)

    def parse_args(self, argv: Optional[List[str]] = None) -> "Args":
        ap = argparse.ArgumentParser(allow_abbrev=False)
        for k, v in _FLAGS.items():
            names = [f"--{k}"] + (["--data"] if k == "data_name" else [])
            ap.add_argument(*names, dest=k, type=type(v), default=getattr(self, k))
        ns = ap.parse_args(argv)
        for k in _FLAGS:
            setattr(self, k, getattr(ns, k))
        return self

    def as_dict(self) -> Dict[str, Any]:
        return {k: getattr(self, k) for k in _FLAGS}

    def save(self, path: str) -> None:
        with open(path, "w") as f:
            json.dump(self.as_dict(), f, indent=4, sort_keys=True)

    def __repr__(self) -> str:
        return "Args(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in _FLAGS) + ")"


def is_args_valid(args: Args) -> None:
    """Validate argument values (src/args.py:372-378): the reference's two asserts (the model name is left to
    ``init_model``, like there), then this harness's own additions -- a flag combination that is not implemented is an
    error, never silently ignored.  Called by every ``main`` right after parsing."""
    assert any(key in args.data_name for key in ["poiseuille", "cavity", "karman", "tube", "dam", "cylinder"])
    assert args.batch_size > 0
    assert args.eval_batch_size > 0 and args.gradient_accumulation_steps >= 1
    assert args.lr_scheduler in ("step", "plateau", "cosine"), args.lr_scheduler
    assert args.dtype in ("fp32", "bf16"), args.dtype
    if args.dtype == "bf16":  # bf16 activation STORAGE: FNO inference (test_multistep) and the fused FNO trainer
        assert args.model == "fno", "--dtype bf16 is built for the FNO (test_multistep, train_auto --fused 1)"
    if args.fused:
        assert args.model == "fno", "--fused 1 is the FnoTrainEngine path"
        assert args.gradient_accumulation_steps == 1, "--fused 1 runs one fused optimiser step per batch"
    if args.graph:
        assert not args.fused, "--graph 1 captures the autograd step; the fused FNO engine has its own launch path"
        assert args.gradient_accumulation_steps == 1, "--graph 1 captures one optimiser step per batch"
    assert args.unet_insert_case_params_at in ("input", "hidden")
