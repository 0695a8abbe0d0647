"""Auto-FNO with the reference's class names, constructor signatures, ``state_dict`` keys and return dicts
(src/models/fno/fno2d.py:17-295), running on the hand-written gfx950 kernels.

``Fno2d.forward`` is ONE autograd node (functional.FnoForwardFn -> cfd_fno_forward / cfd_fno_backward); the
``nn.Conv2d`` sub-modules only hold parameters (so initialisation, key names and checkpoints are identical to the
reference) and are never called.
"""
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F_
from ..base_model import AutoCfdModel

# The reference seeds the global RNGs when its module is imported (fno2d.py:13-14); a drop-in must
# reproduce that side effect or "same script, same seed" no longer gives the same initial weights.
torch.manual_seed(0)
np.random.seed(0)


class SpectralConv2d_fast(nn.Module):
    """2-D Fourier layer: rfft2 -> per-mode complex channel mixing on the kept corner modes -> irfft2
    (fno2d.py:17-82), evaluated as pruned DFTs on the matrix pipe (csrc/spectral.hip)."""

    def __init__(self, in_channels, out_channels, modes1, modes2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.modes1 = modes1
        self.modes2 = modes2
        self.scale = 1 / (in_channels * out_channels)
        self.weights1 = nn.Parameter(
            self.scale * torch.rand(in_channels, out_channels, self.modes1, self.modes2, dtype=torch.cfloat))
        self.weights2 = nn.Parameter(
            self.scale * torch.rand(in_channels, out_channels, self.modes1, self.modes2, dtype=torch.cfloat))

    def forward(self, x: Tensor) -> Tensor:
        return F_.spectral_conv2d(x, self.weights1, self.weights2)


class FnoBlock(nn.Module):
    """act(spectral(x) + conv1x1(x)) (fno2d.py:85-112).  Parameter container inside Fno2d; callable on its own."""

    def __init__(self, in_chan: int, out_chan: int, modes1: int, modes2: int, act_fn: Optional[nn.Module] = None):
        super().__init__()
        self.in_chan = in_chan
        self.out_chan = out_chan
        self.modes1 = modes1
        self.modes2 = modes2
        self.act_fn = act_fn
        self.conv0 = SpectralConv2d_fast(self.in_chan, self.out_chan, self.modes1, self.modes2)
        self.w0 = nn.Conv2d(self.in_chan, self.out_chan, 1)

    def forward(self, x: Tensor) -> Tensor:
        return F_.fno_block(x, self.conv0.weights1, self.conv0.weights2, self.w0.weight, self.w0.bias,
                            gelu=self.act_fn is not None)


class Fno2d(AutoCfdModel):
    def __init__(self, in_chan: int, out_chan: int, n_case_params: int, loss_fn: nn.Module, num_layers: int,
                 modes1: int = 12, modes2: int = 12, hidden_dim: int = 20, padding: Optional[int] = None):
        super().__init__(loss_fn)
        if padding is not None:
            # init_model always passes None (src/utils/autoregressive.py:115-124); domain padding is not on the hot path
            raise NotImplementedError("cfdbench_amd.Fno2d: padding must be None (as the reference's init_model uses it)")
        self.in_chan = in_chan
        self.out_chan = out_chan
        self.n_case_params = n_case_params
        self.num_layers = num_layers
        self.modes1 = modes1
        self.modes2 = modes2
        self.hidden_dim = hidden_dim
        self.padding = padding
        self.act_fn = nn.GELU()
        # Same construction order as the reference (fno2d.py:147-176) => same RNG draws => same initial weights.
        self.fc0 = nn.Conv2d(in_chan + 1 + 2 + n_case_params, self.hidden_dim, 1, 1, 0)
        self.blocks = nn.Sequential(*[
            FnoBlock(self.hidden_dim, self.hidden_dim, self.modes1, self.modes2, self.act_fn)
            for _ in range(self.num_layers)])
        self.fc1 = nn.Conv2d(self.hidden_dim, 128, 1, 1, 0)
        self.fc2 = nn.Conv2d(128, self.out_chan, 1, 1, 0)

    # ---- parameter order of the C ABI (cfd_fno_params) --------------------------------------------------
    def abi_parameters(self) -> List[nn.Parameter]:
        ps = [self.fc0.weight, self.fc0.bias]
        for blk in self.blocks:
            ps += [blk.conv0.weights1, blk.conv0.weights2, blk.w0.weight, blk.w0.bias]
        ps += [self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias]
        return ps

    def abi_config(self) -> dict:
        return dict(num_layers=self.num_layers, hidden=self.hidden_dim, modes1=self.modes1, modes2=self.modes2,
                    head=128, out_chan=self.out_chan)

    def forward(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor] = None,
                label: Optional[Tensor] = None) -> Dict:
        """inputs (b,c,h,w), case_params (b,p), mask (b,1,h,w)|(b,h,w)|None, label (b,c,h,w)|None ->
        {"preds": (b,c,h,w)[, "loss": {mse,rmse,mae[,nmse]}]}   (fno2d.py:178-242)."""
        if mask is not None and mask.dim() == 3:  # fno2d.py:193-194
            mask = mask.unsqueeze(1)
        cfg = dict(self.abi_config(), grad_enabled=torch.is_grad_enabled())
        preds, sums = F_.FnoForwardFn.apply(cfg, inputs, case_params, mask, label, *self.abi_parameters())
        if label is not None:
            normalize = bool(getattr(self.loss_fn, "normalize", True))
            return dict(preds=preds, loss=F_.scores_from_sums(sums, normalize))
        return dict(preds=preds)

    def get_coords(self, shape, device):  # fno2d.py:244-255 (kept for API compatibility; the kernels build coords in place)
        bsz, c, size_x, size_y = shape
        grid_x = torch.tensor(np.linspace(0, 1, size_x), dtype=torch.float)
        grid_x = grid_x.reshape(1, 1, size_x, 1).repeat([bsz, 1, 1, size_y])
        grid_y = torch.tensor(np.linspace(0, 1, size_y), dtype=torch.float)
        grid_y = grid_y.reshape(1, 1, 1, size_y).repeat([bsz, 1, size_x, 1])
        return torch.cat([grid_x, grid_y], dim=1).to(device)

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return self.forward(inputs=inputs, case_params=case_params, mask=mask)["preds"]  # fno2d.py:257-267

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int) -> List[Tensor]:
        """x (c,h,w)|(b,c,h,w), case_params (p)|(b,p), mask (h,w)|(b,h,w)|(b,1,h,w) -> `steps` frames (fno2d.py:269-295)."""
        assert len(inputs.shape) == len(case_params.shape) + 2
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        assert inputs.shape[0] == case_params.shape[0] == mask.shape[0]
        cur_frame = inputs
        preds = []
        for _ in range(steps):  # fno2d.py:290-294 (callers wrap in no_grad / inference_mode as the reference's do)
            cur_frame = self.generate(inputs=cur_frame, case_params=case_params, mask=mask)
            preds.append(cur_frame)
        return preds
