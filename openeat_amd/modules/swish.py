"""Swish activation (/root/reference/openeat/modules/swish.py:12-17)."""
import torch

from openeat_amd import ops


class Swish(torch.nn.Module):
    act_id = ops.ACT_SWISH

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.activation(x, ops.ACT_SWISH)
