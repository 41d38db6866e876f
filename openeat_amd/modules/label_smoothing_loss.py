"""Label-smoothing KL loss (/root/reference/openeat/modules/label_smoothing_loss.py:58-91)."""
import torch
from torch import nn

from openeat_amd import ops


class LabelSmoothingLoss(nn.Module):
    def __init__(self, size: int, padding_idx: int, smoothing: float, normalize_length: bool = False):
        super().__init__()
        self.padding_idx = padding_idx
        self.confidence = 1.0 - smoothing
        self.smoothing = smoothing
        self.size = size
        self.normalize_length = normalize_length

    def forward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """x (B,L,V) logits, target (B,L) with padding_idx on padded positions -> scalar loss."""
        assert x.size(2) == self.size
        return ops.LSMLossFn.apply(x, target, self.smoothing, self.normalize_length, self.padding_idx)

    def fused_head(self, hidden: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, target: torch.Tensor):
        """output_layer + loss + accuracy counts without materialising log-probabilities."""
        return ops.lsm_head(hidden, weight, bias, target, self.smoothing, self.normalize_length, self.padding_idx)
