"""Learning-rate schedules and early stopping for the two training loops.

The benchmark loops of the reference use ``StepLR(step_size=lr_step_size, gamma=0.9)`` per epoch (src/train_auto.py:213-214);
this fork's other trainers use ``ReduceLROnPlateau(factor=lr_scheduler_factor, patience=lr_scheduler_patience)`` on the
validation loss (src/train_vae.py:135-140) with early stopping on ``early_stopping_patience`` / ``early_stopping_delta``
(src/train_vae.py:231-245), and cosine decay in the GenCast trainer.  ``--lr_scheduler step|plateau|cosine`` selects one of
them for ``train_auto`` / ``train``; the default reproduces the reference's benchmark loop.

The schedule is always a torch scheduler: the autograd path attaches it to its Adam, the fused engine (whose learning rate
is a kernel argument) drives it through a one-parameter shadow optimizer and reads the rate back each epoch, so both paths
follow torch's semantics to the bit."""
from __future__ import annotations

from typing import Optional

import torch
from torch.optim import lr_scheduler

SCHEDULES = ("step", "plateau", "cosine")


class LrSchedule:
    def __init__(self, kind: str, lr: float, num_epochs: int, optimizer: Optional[torch.optim.Optimizer] = None,
                 lr_step_size: int = 20, lr_gamma: float = 0.9, factor: float = 0.5, patience: int = 5):
        if kind not in SCHEDULES:
            raise ValueError(f"--lr_scheduler must be one of {SCHEDULES}, got {kind!r}")
        self.kind = kind
        self._shadow = optimizer is None
        if optimizer is None:  # fused engine: a shadow optimizer carries the schedule's state
            optimizer = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
        self.optimizer = optimizer
        if kind == "step":
            self.sched = lr_scheduler.StepLR(optimizer, step_size=lr_step_size, gamma=lr_gamma)
        elif kind == "cosine":
            self.sched = lr_scheduler.CosineAnnealingLR(optimizer, T_max=max(1, num_epochs))
        else:
            self.sched = lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=factor, patience=patience)

    @property
    def lr(self) -> float:
        return float(self.optimizer.param_groups[0]["lr"])

    def epoch_end(self) -> None:
        """After every training epoch (StepLR / cosine advance here, train_auto.py:280)."""
        if self._shadow:
            self.optimizer.step()  # no gradients: a no-op that keeps torch's "step order" bookkeeping quiet
        if self.kind != "plateau":
            self.sched.step()

    def validation(self, dev_loss: float) -> None:
        """After every evaluation (the plateau schedule watches the validation loss, train_vae.py:226)."""
        if self.kind == "plateau":
            self.sched.step(dev_loss)

    def state_dict(self) -> dict:
        return dict(kind=self.kind, sched=self.sched.state_dict(), lr=self.lr)

    def load_state_dict(self, state: dict) -> None:
        if state.get("kind", self.kind) != self.kind:
            raise RuntimeError(f"train_state.pt was written with --lr_scheduler {state['kind']}, this run uses {self.kind}")
        self.sched.load_state_dict(state["sched"])
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g["lr"]):  # capturable optimiser (train_auto --graph 1): the rate lives on the device
                g["lr"].fill_(float(state["lr"]))
            else:
                g["lr"] = state["lr"]


class EarlyStopping:
    """Stop when the validation loss has not improved by ``delta`` for ``patience`` evaluations (train_vae.py:231-245);
    ``patience`` <= 0 disables it (the benchmark loops of the reference never stop early)."""

    def __init__(self, patience: int = 0, delta: float = 1e-5):
        self.patience, self.delta = int(patience), float(delta)
        self.best = float("inf")
        self.bad = 0

    def update(self, dev_loss: float) -> bool:
        if self.patience <= 0:
            return False
        if dev_loss < self.best - self.delta:
            self.best, self.bad = dev_loss, 0
        else:
            self.bad += 1
        return self.bad >= self.patience

    def state_dict(self) -> dict:
        return dict(best=self.best, bad=self.bad)

    def load_state_dict(self, state: dict) -> None:
        self.best, self.bad = float(state["best"]), int(state["bad"])
