"""Conformer / Transformer encoder block (/root/reference/openeat/modules/encoder_layer.py:64-112):
[x + 1/2 FFN(LN x)] -> x + MHA(LN x) -> [x + Conv(LN x)] -> x + s FFN(LN x) -> [LN].  Every
``residual + scale * dropout(.)`` is fused into the last GEMM of its branch."""
from typing import Optional

import torch
from torch import nn

from openeat_amd import ops


class EncoderLayer(nn.Module):
    def __init__(self, size: int, feed_forward_macaron: Optional[nn.Module], self_attn: nn.Module,
                 conv_module: Optional[nn.Module], feed_forward: nn.Module, adapter: Optional[nn.Module],
                 dropout_rate: float = 0.1):
        super().__init__()
        self.size = size
        self.feed_forward_macaron = feed_forward_macaron
        self.self_attn = self_attn
        self.conv_module = conv_module
        self.feed_forward = feed_forward
        self.adapter = adapter
        self.ff_scale = 1
        if feed_forward_macaron is not None:
            self.ff_scale = 0.5
            self.norm_ff_macaron = nn.LayerNorm(size, eps=1e-12)
        self.norm_mha = nn.LayerNorm(size, eps=1e-12)
        if conv_module is not None:
            self.norm_conv = nn.LayerNorm(size, eps=1e-12)
        self.norm_ff = nn.LayerNorm(size, eps=1e-12)
        self.dropout = nn.Dropout(dropout_rate)
        if conv_module is not None:
            self.norm_final = nn.LayerNorm(size, eps=1e-12)

    @staticmethod
    def _ln(norm: nn.LayerNorm, x, rowmask=None, sole=True):
        return ops.layer_norm(x, norm.weight, norm.bias, norm.eps, rowmask, sole_consumer=sole)

    @staticmethod
    def _fork(norm: nn.LayerNorm, x, rowmask=None, sole=True):
        """(residual, LN(x)): the two branches of a pre-norm block from one op (their gradients meet in one kernel).
        sole: x feeds nothing but this fork (true everywhere in this layer except where the adapter also reads x).
        fuse_fwd: every caller below hands the normed branch to ONE op of openeat_amd.ops and to nothing else - that op's first kernel
        may compute the norm on its way in (ops._PENDING_LNF)."""
        return ops.pre_norm(x, norm.weight, norm.bias, norm.eps, rowmask, sole_consumer=sole, fuse_fwd=True)

    def forward(self, x: torch.Tensor, masks: torch.Tensor, pos_emb: torch.Tensor, pre=None, defer_final: bool = False):
        """pre: (residual, norm_ff_macaron(residual)) already computed by the caller - the previous layer's norm_final and this
        layer's first norm as one launch (ops.layer_norm_pair); defer_final: return x BEFORE norm_final (the caller applies
        it together with whatever norm follows).  Both default to the plain reference flow."""
        p = self.dropout.p
        if self.feed_forward_macaron is not None:
            r, y = pre if pre is not None else self._fork(self.norm_ff_macaron, x)
            x = self.feed_forward_macaron(y, residual=r, out_scale=self.ff_scale, out_dropout=p)
        else:
            assert pre is None
        r, y = self._fork(self.norm_mha, x)
        x = self.self_attn(y, y, y, masks, pos_emb, residual=r, out_dropout=p)
        if self.conv_module is not None:
            m8 = ops.mask_bytes(masks)
            rowmask = m8.view(-1)
            # the reference zeroes padded frames of norm_conv's output in place (convolution.py:88-89):
            # fused into the LayerNorm kernel, so the module skips its own input-mask pass.
            r, y = self._fork(self.norm_conv, x, rowmask)
            x = self.conv_module(y, m8, residual=r, out_dropout=p, input_masked=True)
        adapt_x = self.adapter(x) if self.adapter is not None else None      # encoder_layer.py:97-100
        r, y = self._fork(self.norm_ff, x, sole=adapt_x is None)
        x = self.feed_forward(y, residual=r, out_scale=self.ff_scale, out_dropout=p)
        if adapt_x is not None:
            x = ops.add(x, adapt_x)                                           # encoder_layer.py:108
        if self.conv_module is not None and not defer_final:
            x = self._ln(self.norm_final, x)
        return x, masks
