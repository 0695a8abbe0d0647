"""Auto-regressive FFN ("data-driven PINN") with the reference's constructor, ``state_dict`` keys (``ffn.layers.*``) and
return conventions (src/models/auto_ffn.py:13-178).

The reference materialises one row per (frame, query) sample, ``[flat field | case params | query]`` of width h*w+p+2
(auto_ffn.py:100-106: 17 GB at b=256, 64x64).  The first Linear is linear in that concatenation, so here it is two small
GEMMs -- ``U = [field | params] W_a^T + b`` (b rows) and ``V = query W_q^T`` (k rows) -- and row r of the first hidden
layer is ``U[r % b] + V[r % k]``: exactly the reference's pairing, including its quirk that ``repeat`` tiles the frames
while ``preds.view(b, -1)`` reads the rows frame-major (sample (i, j) of the output is frame (i*k + j) % b, query j).
The remaining layers run on the (b*k, width) matrix through the fused Linear+activation GEMMs."""
from typing import List, Optional

import torch
from torch import Tensor

from .. import functional as F_
from .act_fn import NormAct, get_act_fn
from .auto_deeponet import AutoDeepONet
from .base_model import AutoCfdModel
from .ffn import Ffn
from .loss import MseLoss


class AutoFfn(AutoCfdModel):
    def __init__(self, input_field_dim: int, num_case_params: int, query_dim: int, loss_fn: MseLoss,
                 num_label_samples: int = 1000, depth: int = 8, width: int = 100, act_norm: bool = False,
                 act_name="relu"):
        super().__init__(loss_fn)
        self.input_field_dim = input_field_dim
        self.num_case_params = num_case_params
        self.query_dim = query_dim
        self.depth = depth
        self.width = width
        self.act_name = act_name
        self.act_norm = act_norm
        self.num_label_samples = num_label_samples
        self.in_dim = input_field_dim + num_case_params + query_dim
        act_fn = get_act_fn(act_name, act_norm)
        self.widths = [self.in_dim] + [width] * depth + [1]
        self.ffn = Ffn(self.widths, act_fn=act_fn, act_on_output=False)
        self._lattice = {}

    _full_lattice = AutoDeepONet._full_lattice

    def forward(self, inputs: Tensor, case_params: Tensor, label: Optional[Tensor] = None,
                mask: Optional[Tensor] = None, query_idxs: Optional[Tensor] = None):
        batch_size, _num_chan, height, width = inputs.shape
        u = inputs[:, 0]
        flat = torch.cat([u.reshape(batch_size, -1), case_params], dim=1)           # (b, h*w + p)   (:84-89)
        if query_idxs is None:
            query_idxs = self._full_lattice(height, width, inputs.device)
        k = query_idxs.shape[0]
        lin0 = self.ffn.layers[0]
        na = flat.shape[1]
        # first Linear split over the concatenation [frame part | query part]
        U = F_.linear_act(flat, lin0.weight[:, :na], lin0.bias, None)                 # (b, width)
        V = F_.linear_act(query_idxs.float(), lin0.weight[:, na:], None, None)        # (k, width)
        # rows of auto_ffn.py:100-106: row r pairs frame r % b with query r % k, i.e. U tiled k times + V tiled b times.  Written as
        # broadcasts (not U[r % b] + V[r % k] with index tensors: their backward is torch's sort-based index_put accumulation, 2 x 2.5 ms
        # per step at b = 32 -- a third of the step; a broadcast's backward is a plain strided sum)
        w_ = U.shape[1]
        h = U.unsqueeze(0).expand(k, batch_size, w_).reshape(batch_size * k, w_) + V.unsqueeze(0).expand(batch_size, k, w_).reshape(batch_size * k, w_)
        preds = self.ffn.forward_from(h, 0).view(batch_size, -1)                      # (b, k)        (:108-109)
        residuals = u[:, query_idxs[:, 0], query_idxs[:, 1]]                          # (:112-113)
        preds = preds + residuals
        if label is not None:
            labels = label[:, 0][:, query_idxs[:, 0], query_idxs[:, 1]]
            return dict(preds=preds, loss=self.loss_fn(labels=labels, preds=preds))
        return dict(preds=preds.view(-1, 1, height, width))

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Tensor) -> Tensor:
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
        batch_size, num_chan, height, width = inputs.shape
        preds = self.forward(inputs, case_params=case_params, mask=mask)["preds"]
        return preds.view(-1, 1, height, width)

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int) -> List[Tensor]:
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        cur_frame = inputs
        preds = []
        for _ in range(steps):
            cur_frame = self.generate(cur_frame, case_params=case_params, mask=mask)
            preds.append(cur_frame)
        return preds
