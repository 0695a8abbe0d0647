"""``MseLoss`` / ``loss_name_to_fn`` with the reference's names and semantics (src/models/loss.py:8-50); the three
full-tensor reductions run as ONE fused HIP pass instead of F.mse_loss + F.l1_loss + mean, the whole loss as one autograd node
(functional.MseLossFn: cfd_mse_loss_fwd / cfd_mse_loss_bwd)."""
from typing import List

from torch import Tensor, nn

from ..functional import mse_loss_scores


class MseLoss(nn.Module):
    def __init__(self, normalize: bool, is_masked: bool = False):
        super().__init__()
        self.normalize = normalize
        self.is_masked = is_masked

    def get_score_names(self) -> List[str]:  # loss.py:14-20
        names = ["mse", "rmse", "mae"]
        if self.normalize:
            names += ["nmse"]
        return names

    def forward(self, preds: Tensor, labels: Tensor) -> dict:  # loss.py:22-37
        return mse_loss_scores(preds, labels, self.normalize)


def loss_name_to_fn(name: str, masked: bool = False) -> MseLoss:  # loss.py:40-50
    name = name.lower()
    if masked:
        raise NotImplementedError
    if name == "mse":
        return MseLoss(normalize=False)
    if name == "nmse":
        return MseLoss(normalize=True)
    raise NotImplementedError
