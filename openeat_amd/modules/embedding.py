"""Sinusoidal positional encodings (/root/reference/openeat/modules/embedding.py).
The table is not a registered buffer in the reference (so it is not in the
checkpoint); same here.  The x*sqrt(d) (+pe) arithmetic itself is fused into
the producing GEMM / embedding kernel by the callers; the stand-alone
``forward`` is kept for API parity."""
import math
from typing import Tuple

import torch

from openeat_amd import hip


def sinusoid_table(d_model: int, max_len: int) -> torch.Tensor:
    pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


class PositionalEncoding(torch.nn.Module):
    """embedding.py:14-60: x*sqrt(d) + pe ; returns (x, pos_emb)."""
    kind = "abs_pos"

    def __init__(self, d_model: int, max_len: int = 5000, reverse: bool = False):
        super().__init__()
        self.d_model = d_model
        self.xscale = math.sqrt(d_model)
        self.max_len = max_len
        self.pe = sinusoid_table(d_model, max_len)

    def table(self, device, size: int) -> torch.Tensor:
        assert size < self.max_len
        if self.pe.device != device:
            self.pe = self.pe.to(device)
        return self.pe[:, :size]

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        pos = self.table(x.device, x.size(1))
        out = torch.empty_like(x)
        T, d = x.size(1), x.size(2)
        pe_b = pos.expand(x.size(0), T, d).contiguous() if self.kind == "abs_pos" else None
        hip.call("oe_axpby", x.contiguous(), pe_b, x.numel(), self.xscale, 1.0, None, out)
        return out, pos


class RelPositionalEncoding(PositionalEncoding):
    """embedding.py:63-88: x*sqrt(d) ; returns (x, pos_emb[:, :T])."""
    kind = "rel_pos"

    def __init__(self, d_model: int, max_len: int = 5000):
        super().__init__(d_model, max_len, reverse=True)
