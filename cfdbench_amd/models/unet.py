"""U-Net with the reference's module tree, constructor and ``state_dict`` (136 tensors incl. BatchNorm buffers under
``in_conv.conv{1,2}.{0,1}.*``, ``down{1-4}.maxpool_conv.1.*``, ``up{1-4}.{up,conv}.*``, ``out_conv.conv.*``;
src/models/unet.py:11-263) on the gfx950 kernels of csrc/conv6.hip (3x3 convs: three-piece split-bf16 implicit GEMMs, fp32-exact class)
and csrc/conv.hip.  The torch sub-modules (nn.Conv2d, nn.BatchNorm2d,
nn.ConvTranspose2d) only hold parameters and buffers; the arithmetic runs in: implicit-GEMM replicate-padded conv (MFMA),
fused BatchNorm+ReLU (batch statistics in training, running statistics in eval), 2x2 max-pool, 2x2/stride-2 transposed
conv, 1x1 output conv and the (x + residual) * mask epilogue.  ``torch.cat`` / zero ``F.pad`` of skip connections are the
only ATen calls (pure data movement).  ``bilinear=True`` (unet.py:74-78: nn.Upsample(scale 2, bilinear, align_corners) +
DoubleConv with in_channels // 2 mid channels, halved channel counts from down4 on) runs on ``cfd_upsample2_bilinear_*``;
``init_model`` never passes it (src/utils/autoregressive.py:105-114), and like the reference it only composes with
``insert_case_params_at="input"`` (the hidden Linear is sized dim * 16 while down4 then emits dim * 8 channels)."""
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import functional as F_
from .base_model import AutoCfdModel


def _conv_bn_relu(seq: nn.Sequential, x: Tensor) -> Tensor:
    conv, bn = seq[0], seq[1]
    if bn.training and x.is_cuda:  # one node: the conv emits the BatchNorm's batch statistics where it can (functional.ConvBnReluFn)
        return F_.ConvBnReluFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, bn.eps,
                                     bn.momentum, *F_.conv_frags(conv))
    x = F_.Conv2dReplicateFn.apply(x, conv.weight, conv.bias, *F_.conv_frags(conv))
    y = F_.BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, True, bn.eps,
                             bn.momentum)
    return y  # (num_batches_tracked: one fused increment per forward, UNet.forward)


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2  (unet.py:11-50)."""

    def __init__(self, in_chan: int, out_chan: int, mid_chan: Optional[int] = None):
        super().__init__()
        if mid_chan is None:
            mid_chan = out_chan
        self.conv1 = nn.Sequential(
            nn.Conv2d(in_chan, mid_chan, kernel_size=3, padding=1, bias=True, padding_mode="replicate"),
            nn.BatchNorm2d(mid_chan), nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(
            nn.Conv2d(mid_chan, out_chan, kernel_size=3, padding=1, bias=True, padding_mode="replicate"),
            nn.BatchNorm2d(out_chan), nn.ReLU(inplace=True))

    def forward(self, x):
        return _conv_bn_relu(self.conv2, _conv_bn_relu(self.conv1, x))


class Down(nn.Module):
    """Downscaling with maxpool then double conv (unet.py:53-63)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv[1](F_.MaxPool2Fn.apply(x))

    def forward_with_skip(self, x):
        """(this block's output, x as the skip connection's input): the pooling node hands x on, so that its backward pass takes
        the skip connection's gradient too (functional.MaxPoolSkipFn)."""
        pooled, skip = F_.MaxPoolSkipFn.apply(x)
        return self.maxpool_conv[1](pooled), skip


class Up(nn.Module):
    """Upscaling then double conv (unet.py:66-99)."""

    def __init__(self, in_channels, out_channels, bilinear=False):
        super().__init__()
        self.bilinear = bilinear
        if bilinear:  # plain convolutions reduce the channel count (unet.py:72-78)
            self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)  # parameter-free; holds the reference's name
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        if not self.bilinear and F_.ConvTransposeCatFn.supported(x1, self.up.weight, x2):
            # the transposed convolution writes behind the skip tensor inside the concatenation (no separate pass over its output)
            return self.conv(F_.ConvTransposeCatFn.apply(x1, self.up.weight, self.up.bias, x2))
        if self.bilinear:
            x1 = F_.UpsampleBilinear2xFn.apply(x1)
        else:
            x1 = F_.ConvTranspose2x2Fn.apply(x1, self.up.weight, self.up.bias)
        diffY = x2.size()[2] - x1.size()[2]
        diffX = x2.size()[3] - x1.size()[3]
        if diffX or diffY:
            x1 = F.pad(x1, [diffX // 2, diffX - diffX // 2, diffY // 2, diffY - diffY // 2])
        return self.conv(torch.cat([x2, x1], dim=1))


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x: Tensor) -> Tensor:
        return F_.Conv2dReplicateFn.apply(x, self.conv.weight, self.conv.bias)


class UNet(AutoCfdModel):
    def __init__(self, in_chan: int, out_chan: int, loss_fn: nn.Module, n_case_params: int,
                 insert_case_params_at: str = "hidden", bilinear: bool = False, dim: int = 8):
        assert insert_case_params_at in ["hidden", "input"]
        super().__init__(loss_fn)
        self.in_chan = in_chan
        self.out_chan = out_chan
        self.n_case_params = n_case_params
        self.insert_case_params_at = insert_case_params_at
        self.bilinear = bilinear
        self.dim = dim
        if insert_case_params_at == "hidden":  # registered first, like the reference (unet.py:132-133): state_dict order
            self.case_params_fc = nn.Linear(n_case_params, dim * 16)
            self.in_conv = DoubleConv(in_chan + 1, dim)  # + 1 for mask
        else:
            self.in_conv = DoubleConv(in_chan + 1 + n_case_params, dim)
        self.down1 = Down(dim, dim * 2)
        self.down2 = Down(dim * 2, dim * 4)
        self.down3 = Down(dim * 4, dim * 8)
        factor = 2 if bilinear else 1  # unet.py:145-150
        self.down4 = Down(dim * 8, dim * 16 // factor)
        self.up1 = Up(dim * 16, dim * 8 // factor, bilinear)
        self.up2 = Up(dim * 8, dim * 4 // factor, bilinear)
        self.up3 = Up(dim * 4, dim * 2 // factor, bilinear)
        self.up4 = Up(dim * 2, dim, bilinear)
        self.out_conv = OutConv(dim, out_chan)
        # MFMA fragments of all 3x3 weights, remade by one launch at the top of every forward pass (functional.PreparedConvWeights)
        self._prep = F_.PreparedConvWeights([m for m in self.modules() if isinstance(m, nn.Conv2d)])

    def forward(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor] = None, label: Optional[Tensor] = None):
        """inputs (B,c,h,w), mask (B,h,w)|(B,1,h,w), label (B,c,h,w), case_params (b,p)  (unet.py:153-223)."""
        batch_size, n_chan, height, width = inputs.shape
        residual = inputs
        if mask is None:
            mask = torch.ones((batch_size, 1, height, width), device=inputs.device)
        elif mask.dim() == 3:
            mask = mask.unsqueeze(1)
        if inputs.is_cuda:
            self._prep.refresh(torch.is_grad_enabled())
        if self.insert_case_params_at == "input":
            cp = case_params.unsqueeze(2).unsqueeze(3).expand(-1, -1, height, width)
            x = torch.cat([inputs, mask, cp], dim=1)
        elif (inputs.is_cuda and inputs.dtype == torch.float32 and mask.dtype == torch.float32 and inputs.is_contiguous() and mask.is_contiguous()
              and not inputs.requires_grad and not mask.requires_grad):
            # [inputs | mask] along the channels = rows (B, c h w) | (B, h w): one streaming launch (cfd_rows_concat2; ATen's batched cat took
            # 21 us for the 6.3 MB of configs[2])
            x = F_.rows_concat2(inputs.view(batch_size, -1), mask.view(batch_size, -1)).view(batch_size, n_chan + 1, height, width)
        else:
            x = torch.cat([inputs, mask], dim=1)
        x1 = self.in_conv(x)
        if x1.is_cuda:
            x2, x1 = self.down1.forward_with_skip(x1)
            x3, x2 = self.down2.forward_with_skip(x2)
            x4, x3 = self.down3.forward_with_skip(x3)
            x5, x4 = self.down4.forward_with_skip(x4)
        else:
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            x4 = self.down3(x3)
            x5 = self.down4(x4)
        if self.insert_case_params_at == "hidden":  # x5 + Linear(case_params)[:, :, None, None]  (unet.py:198-204)
            conds = F_.linear_act(case_params, self.case_params_fc.weight, self.case_params_fc.bias, None)
            x5 = F_.ChannelBiasAddFn.apply(x5, conds)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        x = self.up4(x, x1)
        preds = F_.ResidualMaskFn.apply(self.out_conv(x), residual, mask)  # (out_conv + inputs[:, :out]) * mask
        if self.training:  # nn.BatchNorm2d's step counters: ONE multi-tensor launch instead of 18 one-element increments
            counters = [m.num_batches_tracked for m in self.modules()
                        if isinstance(m, nn.BatchNorm2d) and m.training and m.num_batches_tracked is not None]
            if counters:
                torch._foreach_add_(counters, 1)
        if label is not None:
            label = F_.ResidualMaskFn.apply(label, None, mask)
            return dict(preds=preds, loss=self.loss_fn(labels=label, preds=preds))
        return dict(preds=preds)

    def generate(self, inputs: Tensor, case_params: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        return self.forward(inputs, case_params=case_params, mask=mask)["preds"]

    def generate_many(self, inputs: Tensor, case_params: Tensor, mask: Tensor, steps: int) -> List[Tensor]:
        """unet.py:225-252, including its quirk: ``mask.unsqueeze(0)`` is applied again after the batch dimension was
        added, so a batched (B,h,w) mask becomes (1,B,h,w) -- i.e. the reference only works for B == 1 or an unbatched
        call.  Reproduced for B == 1; larger batches take the (B,1,h,w) mask the rest of the harness uses."""
        preds = []
        if inputs.dim() == 3:
            inputs = inputs.unsqueeze(0)
            case_params = case_params.unsqueeze(0)
            mask = mask.unsqueeze(0)
        cur_frame = inputs
        if mask.dim() == 3:
            mask = mask.unsqueeze(1) if mask.shape[0] == inputs.shape[0] and inputs.shape[0] > 1 else mask.unsqueeze(0)
        for _ in range(steps):
            cur_frame = self.generate(cur_frame, case_params=case_params, mask=mask)
            preds.append(cur_frame)
        return preds
