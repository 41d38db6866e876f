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
        mask.  (The streaming ``cache`` of the causal variant is not part of the training/offline path: a non-empty one
        is refused.)"""
        if cache is not None and cache.numel() > 0:
            raise NotImplementedError("streaming cache is outside the accelerated path")
        rowmask = None
        if mask_pad is not None and mask_pad.numel() > 0:
            m = mask_pad if mask_pad.dtype == torch.uint8 else mask_pad.to(torch.uint8)
            rowmask = m.contiguous().view(-1)
        dw = self.depthwise_conv
        return ops.conv_module(x, rowmask, self.pointwise_conv1.weight, self.pointwise_conv1.bias,
                               dw.weight.view(dw.weight.shape[0], -1), dw.bias, self.norm.weight, self.norm.bias,
                               self.pointwise_conv2.weight, self.pointwise_conv2.bias, self.kernel_size, self.lorder > 0,
                               self._act, residual, out_dropout if self.training else 0.0, input_masked)
