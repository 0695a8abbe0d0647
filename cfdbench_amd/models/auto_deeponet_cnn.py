"""Auto-regressive DeepONet with a CNN branch, with the reference's constructor, ``state_dict`` keys
(``branch_net.{in_conv,blocks.{0,3,6,9},out_conv}.*``, ``trunk_net.layers.*``, ``out_ffn.layers.*``, ``bias``) and
return conventions (src/models/auto_deeponet_cnn.py:13-238).

The CNN branch uses ZERO-padded 5x5 convolutions.  The conv kernels of this package implement replicate padding, and a
zero-padded convolution is exactly the interior of a replicate-padded one on the explicitly zero-padded image (every
5x5 window of an interior pixel lies inside the padded array, so the clamp never fires):
``zero_conv(x) = crop(replicate_conv(zero_pad(x, 2)), 2)`` -- 13 % more pixels at 64x64, no new kernel.  Max-pool, ReLU,
the Linear+activation stacks and the (b*k, 512) output FFN are the fused kernels of the other families."""
from itertools import product
from typing import List, Optional

import torch
from torch import Tensor, nn

from .. import functional as F_
from .act_fn import get_act_fn
from .auto_deeponet import AutoDeepONet
from .base_model import AutoCfdModel
from .ffn import Ffn
from .loss import MseLoss


def zero_padded_conv(x: Tensor, conv: nn.Conv2d) -> Tensor:
    """nn.Conv2d(k, padding=k//2) (zero padding): on the zero-padding form of the implicit-GEMM kernels where they take the shape
    (functional.Conv2dZeroPadFn, round 4), else through the replicate-padding kernel on a zero-padded input, cropped."""
    if F_.Conv2dZeroPadFn.supported(x, conv.weight):
        return F_.Conv2dZeroPadFn.apply(x, conv.weight, conv.bias)
    p = conv.kernel_size[0] // 2
    y = F_.Conv2dReplicateFn.apply(torch.nn.functional.pad(x, (p, p, p, p)), conv.weight, conv.bias)
    return y[:, :, p:-p, p:-p]


class CnnBranch(nn.Module):
    def __init__(self, in_chan: int, kernel_size: int, padding: int, depth: int = 4):
        super().__init__()
        self.in_chan = in_chan
        self.in_conv = nn.Conv2d(in_chan, 32, kernel_size=kernel_size, padding=padding)
        self.out_conv = nn.Conv2d(32, 32, kernel_size=kernel_size, padding=padding)
        blocks = []
        for _ in range(depth):
            blocks += [nn.Conv2d(32, 32, kernel_size, padding=padding), nn.MaxPool2d(2), nn.ReLU()]
        self.blocks = nn.Sequential(*blocks)  # parameter container; forward() below runs the HIP kernels

    def forward(self, x: Tensor) -> Tensor:
        x = zero_padded_conv(x, self.in_conv)
        mods = list(self.blocks)
        for i in range(0, len(mods), 3):  # Conv2d -> MaxPool2d(2) -> ReLU   (auto_deeponet_cnn.py:27-33)
            x = zero_padded_conv(x, mods[i])
            x = F_.MaxPool2Fn.apply(x.contiguous())
            x = F_.act(x, "relu")
        return zero_padded_conv(x, self.out_conv)


class AutoDeepONetCnn(AutoCfdModel):
    def __init__(self, in_chan: int, query_dim: int, loss_fn: MseLoss, height: int = 100, width: int = 100,
                 num_case_params: int = 5, trunk_depth: int = 4, act_name="relu", act_norm: bool = False,
                 act_on_output: bool = False):
        super().__init__(loss_fn)
        self.in_chan = in_chan
        self.query_dim = query_dim
        self.num_case_params = num_case_params
        self.trunk_depth = trunk_depth
        self.height = height
        self.width = width
        self.act_name = act_name
        self.act_norm = act_norm
        self.act_on_output = act_on_output
        act_fn = get_act_fn(act_name, act_norm)
        self.trunk_dims = [query_dim] + [100] * trunk_depth + [4 * 4 * 32]
        self.branch_net = CnnBranch(in_chan + 1 + num_case_params, kernel_size=5, padding=2)  # + 1: the mask channel
        self.trunk_net = Ffn(self.trunk_dims, act_fn=act_fn, act_on_output=False)
        self.out_ffn = Ffn([32 * 4 * 4] * 3 + [1], act_fn=act_fn, act_on_output=False)
        self.bias = nn.Parameter(torch.zeros(1))  # unused by forward, kept for the state_dict (auto_deeponet_cnn.py:103)
        self._lattice = {}

    _full_lattice = AutoDeepONet._full_lattice

    def forward(self, inputs: Tensor, case_params: Tensor, label: Optional[Tensor] = None,
                mask: Optional[Tensor] = None, query_idxs: Optional[Tensor] = None):
        """inputs (b,c,h,w), case_params (b,p), mask (b,h,w)|(b,1,h,w), query_idxs (k,2) -> preds (b,k) + loss if label,
        else (b,1,h,w)   (auto_deeponet_cnn.py:105-186)."""
        if mask is not None:
            if mask.dim() == 3:
                mask = mask.unsqueeze(1)
            inputs = torch.cat([inputs, mask], dim=1)
        batch_size, num_chan, height, width = inputs.shape
        residuals = inputs
        cp = case_params.unsqueeze(-1).unsqueeze(-1).expand(-1, -1, height, width)
        x_branch = self.branch_net(torch.cat([inputs, cp], dim=1).contiguous()).reshape(batch_size, -1)  # (b, 512)
        if query_idxs is None:
            query_idxs = self._full_lattice(height, width, inputs.device)
        x_trunk = self.trunk_net((query_idxs.float() - 50) / 100)                                        # (k, 512)
        preds = self.out_ffn(x_branch.unsqueeze(1) * x_trunk.unsqueeze(0)).squeeze(-1)                   # (b, k)
        preds = preds + residuals[:, 0, query_idxs[:, 0], query_idxs[:, 1]]
        if label is not None:
            labels = label[:, 0][:, query_idxs[:, 0], query_idxs[:, 1]]
            return dict(preds=preds, loss=self.loss_fn(labels=labels, preds=preds))
        return dict(preds=preds.view(-1, 1, height, width))

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Tensor) -> Tensor:
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        batch_size, num_chan, height, width = inputs.shape
        preds = self.forward(inputs, case_params=case_params, mask=mask)["preds"]
        return preds.view(-1, 1, height, width)

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int) -> List[Tensor]:
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        cur_frame = inputs
        p = inputs[:, -1:]  # the last input channel is carried along unchanged (auto_deeponet_cnn.py:227,235)
        preds = []
        for _ in range(steps):
            cur_frame = self.generate(cur_frame, case_params=case_params, mask=mask)
            preds.append(cur_frame)
            cur_frame = torch.cat([cur_frame, p], dim=1)
        return preds
