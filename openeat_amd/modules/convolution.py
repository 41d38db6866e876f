"""Conformer convolution module (/root/reference/openeat/modules/convolution.py:15-120):
mask -> pointwise(d->2d) -> GLU -> depthwise(K) -> LayerNorm -> activation -> pointwise -> mask."""
from typing import Optional

import torch
from torch import nn

from openeat_amd import ops
from openeat_amd.modules.positionwise_feed_forward import act_id_of


class ConvolutionModule(nn.Module):
    def __init__(self, channels: int, kernel_size: int = 15, activation: nn.Module = nn.ReLU(), causal: bool = False):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(channels, 2 * channels, kernel_size=1, stride=1, padding=0, bias=True)
        if causal:
            padding, self.lorder = 0, kernel_size - 1
        else:
            assert (kernel_size - 1) % 2 == 0
            padding, self.lorder = (kernel_size - 1) // 2, 0
        self.depthwise_conv = nn.Conv1d(channels, channels, kernel_size, stride=1, padding=padding, groups=channels, bias=True)
        self.norm = nn.LayerNorm(channels)
        self.pointwise_conv2 = nn.Conv1d(channels, channels, kernel_size=1, stride=1, padding=0, bias=True)
        self.activation = activation
        self.kernel_size = kernel_size
        self._act = act_id_of(activation)

    def forward(self, x: torch.Tensor, mask_pad: torch.Tensor = torch.ones((0, 0, 0), dtype=torch.bool),
                cache: torch.Tensor = torch.zeros((0, 0, 0)), residual: torch.Tensor = None, out_dropout: float = 0.0,
                input_masked: bool = False) -> torch.Tensor:
        """x (B,T,C); mask_pad (B,1,T) non-zero = real frame, empty (the reference's default, convolution.py:75) = no
        mask.  cache (B,C,lorder), causal variant only (convolution.py:92-104): the previous chunk's last `lorder` input
        frames stand where the zero padding would; the module then runs on [cache; x] and the first lorder outputs - whose
        windows reach into the (second) padding - are dropped: the remaining T outputs see exactly the frames the reference's
        `torch.cat((cache, x), dim=2)` gives them."""
        if cache is not None and cache.numel() > 0 and self.lorder > 0:
            assert cache.size(0) == x.size(0) and cache.size(1) == x.size(2)        # equal batch, equal channel (convolution.py:99-100)
            B, T, _ = x.shape
            if mask_pad is not None and mask_pad.numel() > 0:                          # the reference masks x before the concatenation
                x = x.masked_fill(~mask_pad.to(torch.bool).transpose(1, 2), 0.0)
                mask_pad = torch.cat((torch.ones(B, 1, cache.size(2), dtype=mask_pad.dtype, device=mask_pad.device), mask_pad), dim=2)
            xc = torch.cat((cache.transpose(1, 2).to(x.dtype), x), dim=1).contiguous()
            res = None if residual is None else torch.cat((residual.new_zeros(B, cache.size(2), residual.size(2)), residual), dim=1)
            y = self.forward(xc, mask_pad, torch.zeros((0, 0, 0)), res, out_dropout, input_masked=False)
            return y[:, cache.size(2):]
        rowmask = None
        if mask_pad is not None and mask_pad.numel() > 0:
            rowmask = ops.mask_bytes(mask_pad).view(-1)
        dw = self.depthwise_conv
        return ops.conv_module(x, rowmask, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
                               dw.weight.view(dw.weight.shape[0], -1), dw.bias, self.norm.weight, self.norm.bias,
                               self.pointwise_conv2.weight, self.pointwise_conv2.bias, self.kernel_size, self.lorder > 0,
                               self._act, residual, out_dropout if self.training else 0.0, input_masked)
