"""Auto-regressive enhanced DeepONet (two branch nets: previous field, case parameters) with the reference's
constructor, ``state_dict`` keys (``branch1.layers.*``, ``branch2.layers.*``, ``trunk_net.layers.*``, ``bias``) and
return conventions (src/models/auto_edeeponet.py:13-185).  The branch x trunk contraction with its bias and residual is
the same fused GEMM as AutoDeepONet's (``cfd_deeponet_inner_fwd/bwd``); the (b, k, p) broadcast product never exists."""
from typing import Dict, List, Optional

import torch
from torch import Tensor, nn

from .. import functional as F_
from .act_fn import get_act_fn
from .auto_deeponet import AutoDeepONet
from .base_model import AutoCfdModel
from .ffn import Ffn, run_ffns_together
from .loss import MseLoss


class AutoEDeepONet(AutoCfdModel):
    def __init__(self, dim_branch1: int, dim_branch2: int, trunk_dim: int, loss_fn: MseLoss,
                 num_label_samples: int = 1000, branch_depth: int = 4, trunk_depth: int = 4, width: int = 100,
                 act_name: str = "relu", act_norm: bool = False, act_on_output: bool = False):
        super().__init__(loss_fn)
        self.dim_branch1 = dim_branch1
        self.dim_branch2 = dim_branch2
        self.trunk_dim = trunk_dim
        self.branch_depth = branch_depth
        self.trunk_depth = trunk_depth
        self.width = width
        self.act_name = act_name
        self.act_norm = act_norm
        self.act_on_output = act_on_output
        self.num_label_samples = num_label_samples
        self.branch1_dims = [dim_branch1] + [width] * branch_depth
        self.branch2_dims = [dim_branch2] + [width] * branch_depth
        self.trunk_dims = [trunk_dim] + [width] * trunk_depth
        act_fn = get_act_fn(act_name, act_norm)
        self.branch1 = Ffn(self.branch1_dims, act_fn=act_fn, act_on_output=act_on_output)
        self.branch2 = Ffn(self.branch2_dims, act_fn=act_fn, act_on_output=act_on_output)
        self.trunk_net = Ffn(self.trunk_dims, act_fn=act_fn)
        self.bias = nn.Parameter(torch.zeros(1))
        self._lattice: Dict = {}
        self._trunk_cache = None

    _full_lattice = AutoDeepONet._full_lattice
    _trunk = AutoDeepONet._trunk
    _trunk_input = AutoDeepONet._trunk_input

    def forward(self, inputs: Tensor, case_params: Tensor, label: Optional[Tensor] = None,
                mask: Optional[Tensor] = None, query_idxs: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """inputs (b,c,h,w), case_params (b,p), query_idxs (k,2) -> preds (b,k) [+ loss]  (auto_edeeponet.py:66-131).
        Without a label the reference returns the (b, k) predictions un-reshaped (its ``.view`` result is dropped, :130)."""
        batch_size, num_chan, height, width = inputs.shape
        u = inputs[:, 0]
        full = query_idxs is None
        if full:
            query_idxs = self._full_lattice(height, width, inputs.device)
        if torch.is_grad_enabled() and inputs.is_cuda:
            # training: three independent nets, their Linear stacks in one launch per direction
            b1, b2, x_trunk = run_ffns_together([self.branch1, self.branch2, self.trunk_net],
                                                [u.reshape(batch_size, -1), case_params, self._trunk_input(query_idxs, full)])
            x_branch = b1 * b2                                                           # (:91-93)
        else:
            x_branch = self.branch1(u.reshape(batch_size, -1)) * self.branch2(case_params)  # (:91-93)
            x_trunk = self._trunk(query_idxs, full)                                        # (:103-105)
        qflat = None if full else (query_idxs[:, 0] * width + query_idxs[:, 1])
        preds = F_.DeepONetInnerFn.apply(x_branch, x_trunk, self.bias, u, qflat)         # (:107-112)
        if label is not None:
            label = label[:, 0]
            labels = label.reshape(batch_size, -1) if full else label[:, query_idxs[:, 0], query_idxs[:, 1]]
            assert preds.shape == labels.shape, f"{preds.shape}, {labels.shape}"
            return dict(preds=preds, loss=self.loss_fn(preds=preds, labels=labels))
        return dict(preds=preds)

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        batch_size, num_chan, height, width = inputs.shape
        preds = self.forward(inputs=inputs, case_params=case_params)["preds"]
        return preds.view(-1, 1, height, width)

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int) -> List[Tensor]:
        assert len(inputs.shape) == len(case_params.shape) + 2
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        assert inputs.shape[0] == case_params.shape[0]
        cur_frame = inputs
        preds = []
        for _ in range(steps):
            cur_frame = self.generate(inputs=cur_frame, case_params=case_params, mask=None)
            preds.append(cur_frame)
        return preds
