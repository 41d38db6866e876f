"""Attention decoders (/root/reference/openeat/modules/decoder.py):
TransformerDecoder = Embedding*sqrt(d)+pe -> N x DecoderLayer -> LayerNorm -> Linear(d->V);
BiTransformerDecoder = left-to-right + optional right-to-left decoder on the same memory."""
from typing import List, Optional, Tuple

import torch

from openeat_amd import ops
from openeat_amd.modules.attention import MultiHeadedAttention
from openeat_amd.modules.decoder_layer import DecoderLayer
from openeat_amd.modules.embedding import PositionalEncoding
from openeat_amd.modules.positionwise_feed_forward import PositionwiseFeedForward


def _dec_layers(d, dropout_rate, heads, units, use_adapter, n, down_size=64, scalar=0.1):
    from openeat_amd.modules.adapter import Adapter
    return torch.nn.ModuleList([
        DecoderLayer(d, MultiHeadedAttention(heads, d, dropout_rate), MultiHeadedAttention(heads, d, dropout_rate),
                     PositionwiseFeedForward(d, units, dropout_rate),
                     Adapter(d, dropout_rate, down_size, scalar) if use_adapter else None, dropout_rate) for _ in range(n)])


class Decoder(torch.nn.Module):
    """decoder.py:14-108: embedding-free stack."""

    def __init__(self, d_model: int, dropout_rate: float = 0.1, attention_heads: int = 4, linear_units: int = 2048,
                 use_adapter: bool = False, down_size: int = 64, scalar: float = 0.1, num_blocks: int = 6,
                 num_blocks_share: int = 1):
        super().__init__()
        self.num_blocks_share = num_blocks_share
        self.decoders = _dec_layers(d_model, dropout_rate, attention_heads, linear_units, use_adapter,
                                    num_blocks // num_blocks_share, down_size, scalar)

    def forward(self, tgt, tgt_mask, memory, memory_mask):
        x = tgt
        for layer in self.decoders:
            for _ in range(self.num_blocks_share):
                x = layer(x, tgt_mask, memory, memory_mask)
        return x

    def forward_one_step(self, tgt, tgt_mask, memory, memory_mask, cache: Optional[List[torch.Tensor]] = None
                         ) -> Tuple[torch.Tensor, List[torch.Tensor]]:
        """decoder.py:76-108: one decoding step of the embedding-free stack; cache = each block application's previous
        output (B, L-1, d); returns (x (B, L, d), new_cache)."""
        x = tgt
        tm = ops.mask_bytes(tgt_mask)
        mm = ops.mask_bytes(memory_mask)
        new_cache = []
        for i, layer in enumerate(self.decoders):
            for j in range(self.num_blocks_share):
                c = None if cache is None else cache[i * self.num_blocks_share + j]
                x = layer(x, tm, memory, mm, cache=c)
                new_cache.append(x)
        return x, new_cache


class TransformerDecoder(torch.nn.Module):
    def __init__(self, vocab_size: int, d_model: int, dropout_rate: float = 0.1, attention_heads: int = 4,
                 linear_units: int = 2048, use_adapter: bool = False, down_size: int = 64, scalar: float = 0.1,
                 num_blocks: int = 6, num_blocks_share: int = 1, share_embedding: bool = False):
        super().__init__()
        if share_embedding:
            raise NotImplementedError("share_embedding refers to a non-existent attribute in the reference (decoder.py:164-165)")
        self.num_blocks_share = num_blocks_share
        self.embed = torch.nn.Sequential(torch.nn.Embedding(vocab_size, d_model), PositionalEncoding(d_model))
        self.decoders = _dec_layers(d_model, dropout_rate, attention_heads, linear_units, use_adapter,
                                    num_blocks // num_blocks_share, down_size, scalar)
        self.after_norm = torch.nn.LayerNorm(d_model, eps=1e-12)
        self.output_layer = torch.nn.Linear(d_model, vocab_size)

    def _embed(self, tokens: torch.Tensor) -> torch.Tensor:
        pe = self.embed[1]
        table = pe.table(tokens.device, tokens.size(1)).reshape(tokens.size(1), -1).contiguous()
        return ops.embed(tokens, self.embed[0].weight, table, pe.xscale)

    def hidden(self, tgt, tgt_mask, memory, memory_mask) -> torch.Tensor:
        """Everything up to (and including) after_norm: the input of the output layer."""
        x = self._embed(tgt)
        tm = ops.mask_bytes(tgt_mask)
        mm = ops.mask_bytes(memory_mask)
        for layer in self.decoders:
            for _ in range(self.num_blocks_share):
                x = layer(x, tm, memory, mm)
        return ops.layer_norm(x, self.after_norm.weight, self.after_norm.bias, self.after_norm.eps, sole_consumer=True)

    def forward(self, tgt, tgt_mask, memory, memory_mask) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """decoder.py:167-194 -> (logits (B,L,V), olens, pre_logits)."""
        pre = self.hidden(tgt, tgt_mask, memory, memory_mask)
        x = ops.linear(pre, self.output_layer.weight, self.output_layer.bias)
        return x, tgt_mask.sum(1), pre

    def forward_one_step(self, tgt, tgt_mask, memory, memory_mask, cache: Optional[List[torch.Tensor]] = None):
        """decoder.py:196-232: incremental decoding with the per-layer output cache."""
        x = self._embed(tgt)
        tm = ops.mask_bytes(tgt_mask)
        mm = ops.mask_bytes(memory_mask)
        new_cache = []
        for i, layer in enumerate(self.decoders):
            for j in range(self.num_blocks_share):
                c = None if cache is None else cache[i * self.num_blocks_share + j]
                x = layer(x, tm, memory, mm, cache=c)
                new_cache.append(x)
        y = ops.layer_norm(x[:, -1].contiguous(), self.after_norm.weight, self.after_norm.bias, self.after_norm.eps)
        return ops.linear(y, self.output_layer.weight, self.output_layer.bias), new_cache, y


class BiTransformerDecoder(torch.nn.Module):
    def __init__(self, vocab_size: int, d_model: int, dropout_rate: float = 0.1, attention_heads: int = 4,
                 linear_units: int = 2048, use_adapter: bool = False, down_size: int = 64, scalar: float = 0.1,
                 num_blocks: int = 6, r_num_blocks: int = 0, num_blocks_share: int = 1):
        super().__init__()
        self.r_num_blocks = r_num_blocks
        self.left_decoder = TransformerDecoder(vocab_size, d_model, dropout_rate, attention_heads, linear_units,
                                               use_adapter, down_size, scalar, num_blocks, num_blocks_share)
        if r_num_blocks > 0:
            self.right_decoder = TransformerDecoder(vocab_size, d_model, dropout_rate, attention_heads, linear_units,
                                                    use_adapter, down_size, scalar, r_num_blocks, num_blocks_share)

    def forward(self, memory, memory_mask, ys_in_pad, r_ys_in_pad, tgt_mask):
        """decoder.py:278-309 -> (l_x, r_x, pre_l_x)."""
        l_x, _, pre = self.left_decoder(ys_in_pad, tgt_mask, memory, memory_mask)
        r_x = torch.tensor(0.0)
        if self.r_num_blocks > 0:
            r_x, _, _ = self.right_decoder(r_ys_in_pad, tgt_mask, memory, memory_mask)
        return l_x, r_x, pre

    def forward_one_step(self, tgt, tgt_mask, memory, memory_mask, cache=None):
        return self.left_decoder.forward_one_step(tgt, tgt_mask, memory, memory_mask, cache)
