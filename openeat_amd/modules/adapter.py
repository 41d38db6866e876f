"""Bottleneck adapter (/root/reference/openeat/modules/adapter.py): x + scale * dropout(up(dropout(relu(down(LN(x)))))).
It is a position-wise feed-forward block with a scaled residual, i.e. exactly the fused FFN op of the path."""
import torch
from torch import nn

from openeat_amd import ops


class Adapter(nn.Module):
    def __init__(self, d_model, dropout_rate=0.1, down_size=64, adapter_scalar=0.1):
        super().__init__()
        if adapter_scalar == -1:
            raise NotImplementedError("a learnable adapter scale (adapter_scalar = -1) has no gfx950 path; use a fixed scalar")
        self.scale = adapter_scalar
        self.norm = nn.LayerNorm(d_model, eps=1e-12)
        self.down_proj = nn.Linear(d_model, down_size)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(dropout_rate)
        self.up_proj = nn.Linear(down_size, d_model)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p = self.dropout.p if self.training else 0.0
        r, y = ops.pre_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.feed_forward(y, self.down_proj.weight, self.down_proj.bias, self.up_proj.weight, self.up_proj.bias,
                                ops.ACT_RELU, p, r, float(self.scale), p)
