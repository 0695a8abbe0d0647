"""Non-autoregressive DeepONet with the reference's constructor, ``state_dict`` keys (``branch_net.layers.*``,
``fc_trunk_t.*``, ``fc_trunk_xy.*``, ``trunk_net.layers.*``, ``bias``) and return conventions
(src/models/deeponet.py:13-257): case parameters -> branch net, (t, x, y) -> trunk net, per-sample inner product."""
from itertools import product
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .. import functional as F_
from .act_fn import get_act_fn
from .base_model import CfdModel
from .ffn import Ffn
from .loss import MseLoss


class DeepONet(CfdModel):
    def __init__(self, branch_dim: int, trunk_dim: int, loss_fn: MseLoss, num_label_samples: int = 1000,
                 branch_depth: int = 4, trunk_depth: int = 3, width: int = 100, act_name: str = "relu",
                 act_norm: bool = False, act_on_output: bool = False):
        super().__init__(loss_fn)
        self.branch_dim = branch_dim
        self.trunk_dim = trunk_dim
        self.branch_depth = branch_depth
        self.trunk_depth = trunk_depth
        self.width = width
        self.act_name = act_name
        self.act_norm = act_norm
        self.act_on_output = act_on_output
        self.num_label_samples = num_label_samples
        self.branch_dims = [branch_dim] + [width] * branch_depth
        act_fn = get_act_fn(act_name, act_norm)
        self.branch_net = Ffn(self.branch_dims, act_fn=act_fn, act_on_output=self.act_on_output)
        self.fc_trunk_t = nn.Linear(1, width)
        self.fc_trunk_xy = nn.Linear(2, width)
        self.trunk_dims = [width] * trunk_depth
        self.trunk_net = Ffn(self.trunk_dims, act_fn=act_fn)
        self.bias = nn.Parameter(torch.zeros(1))

    def forward(self, case_params: Tensor, t: Tensor, label: Optional[Tensor] = None,
                query_idxs: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """case_params (b,p), t (b,1), label (b,c,h,w), query_idxs (k,2) -> preds (b,k) [+ loss]  (deeponet.py:153-223);
        without query_idxs, ``num_label_samples`` random lattice points are drawn (torch.randint, as the reference)."""
        if query_idxs is None:
            assert label is not None
            height, width = label.shape[-2:]
            query_idxs = torch.stack([torch.randint(0, height, (self.num_label_samples,), device=label.device),
                                      torch.randint(0, width, (self.num_label_samples,), device=label.device)], dim=-1)
        x_trunk_t = F_.linear_act(t, self.fc_trunk_t.weight, self.fc_trunk_t.bias, None)                     # (b, p)
        x_trunk_xy = F_.linear_act(query_idxs.float(), self.fc_trunk_xy.weight, self.fc_trunk_xy.bias, None)  # (k, p)
        x_trunk = F_.BcastAddFn.apply(x_trunk_t, x_trunk_xy)                                                  # (b, k, p)
        x_branch = self.branch_net(case_params)                                                                # (b, p)
        x_trunk = self.trunk_net(x_trunk)                                                                      # (b, k, p)
        preds = F_.RowDotFn.apply(x_branch, x_trunk, self.bias)                                               # (b, k)
        if label is not None:
            labels = label[:, 0][:, query_idxs[:, 0], query_idxs[:, 1]]
            assert preds.shape == labels.shape, f"{preds.shape}, {labels.shape}"
            return dict(preds=preds, loss=self.loss_fn(preds=preds, labels=labels))
        return dict(preds=preds)

    def generate_one(self, case_params: Tensor, t: Tensor, height: int, width: int) -> Tensor:
        """One frame at time t: (b, 1, h, w)  (deeponet.py:225-257)."""
        if len(case_params.shape) == 1:
            case_params = case_params.unsqueeze(0)
        if len(t.shape) == 0:
            t = t.unsqueeze(0).unsqueeze(0)
        elif len(t.shape) == 1:
            t = t.unsqueeze(0)
        query_idxs = torch.tensor(list(product(range(height), range(width))), device=case_params.device)
        return self.forward(case_params, t=t, query_idxs=query_idxs)["preds"].view(-1, 1, height, width)
