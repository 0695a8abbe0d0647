"""Plugin ABCs, same contract as the reference's src/models/base_model.py:7-81."""
from typing import List, Optional

from torch import Tensor, nn

from .loss import MseLoss


class CfdModel(nn.Module):
    """Non-autoregressive surrogate: (case params, t) -> field (base_model.py:7-38)."""

    def __init__(self, loss_fn: MseLoss):
        super().__init__()
        self.loss_fn = loss_fn

    def forward(self, x: Tensor, mask: Optional[Tensor] = None, label: Optional[Tensor] = None,
                case_params: Optional[dict] = None) -> dict:
        raise NotImplementedError

    def generate_one(self, case_params, t: Tensor, height: int, width: int, **kwargs) -> Tensor:
        raise NotImplementedError


class AutoCfdModel(nn.Module):
    """Autoregressive surrogate: frame t -> frame t+dt (base_model.py:41-81)."""

    def __init__(self, loss_fn: nn.Module):
        super().__init__()
        self.loss_fn = loss_fn

    def forward(self, inputs: Tensor, label: Optional[Tensor] = None, case_params: Optional[dict] = None,
                mask: Optional[Tensor] = None, **kwargs) -> dict:
        raise NotImplementedError

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Tensor, **kwargs) -> Tensor:
        raise NotImplementedError

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int, **kwargs) -> List[Tensor]:
        raise NotImplementedError
