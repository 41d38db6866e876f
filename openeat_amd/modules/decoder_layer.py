"""Transformer decoder block (/root/reference/openeat/modules/decoder_layer.py:47-111)."""
from typing import Optional

import torch
from torch import nn

from openeat_amd import ops


class DecoderLayer(nn.Module):
    def __init__(self, size: int, self_attn: nn.Module, src_attn: nn.Module, feed_forward: nn.Module,
                 adapter: Optional[nn.Module] = None, dropout_rate: float = 0.1):
        super().__init__()
        self.size = size
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.adapter = adapter
        self.norm1 = nn.LayerNorm(size, eps=1e-12)
        self.norm2 = nn.LayerNorm(size, eps=1e-12)
        self.norm3 = nn.LayerNorm(size, eps=1e-12)
        self.dropout = nn.Dropout(dropout_rate)

    @staticmethod
    def _ln(norm, x):
        return ops.layer_norm(x, norm.weight, norm.bias, norm.eps)

    def forward(self, tgt: torch.Tensor, tgt_mask: torch.Tensor, memory: torch.Tensor, memory_mask: torch.Tensor,
                cache: Optional[torch.Tensor] = None):
        p = self.dropout.p
        # (residual, LN(t)) from one op; sole: t feeds nothing else (false where the adapter also reads it)
        fork = lambda norm, t, sole=True: ops.pre_norm(t, norm.weight, norm.bias, norm.eps, sole_consumer=sole)
        if cache is None:
            residual, y = fork(self.norm1, tgt)
            x = self.self_attn(y, y, y, tgt_mask, residual=residual, out_dropout=p)
        else:
            assert cache.shape == (tgt.shape[0], tgt.shape[1] - 1, self.size)
            residual = tgt
            y = self._ln(self.norm1, tgt)
            q = y[:, -1:, :].contiguous()
            x = self.self_attn(q, y, y, tgt_mask[:, -1:, :], residual=residual[:, -1:, :].contiguous(), out_dropout=p)
        r, y = fork(self.norm2, x)
        x = self.src_attn(y, memory, memory, memory_mask, residual=r, out_dropout=p)
        adapt_x = self.adapter(x) if self.adapter is not None else None      # decoder_layer.py:98-101
        r, y = fork(self.norm3, x, adapt_x is None)
        x = self.feed_forward(y, residual=r, out_scale=1.0, out_dropout=p)
        if adapt_x is not None:
            x = ops.add(x, adapt_x)                                           # decoder_layer.py:106
        if cache is not None:
            x = torch.cat([cache, x], dim=1)
        return x
