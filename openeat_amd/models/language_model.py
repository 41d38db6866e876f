"""Transformer language model for shallow fusion (/root/reference/openeat/models/language_model.py).

The reference class cannot be constructed (`d_model`, `dropout_rate`, `attention_heads`, `linear_units`,
`RelPositionalEncoding`, `NoPositionalEncoding` are undefined names at language_model.py:53-64) and its call site scores
hypotheses through `lm.encoder(tokens, lengths)` (asr_model.py:498), which is not the signature of the `Encoder` it
holds.  What IS specified is kept: the architecture (Embedding -> positional encoding -> `Encoder` stack without
macaron / conv module -> Linear(d, V); language_model.py:53-67,109-125, conf/train_lm.yaml), the causal mask for the
autoregressive mode, the training loss (label smoothing on the shifted targets, :69-107) and the fusion arithmetic
(asr_model.py:490-528).  The missing hyper-parameters become constructor arguments with the recipe's values.
Parity for this class is therefore UNPINNED against the reference; it is tested against the CPU restatement in
oracle/asr.py and every building block is pinned by the encoder / decoder goldens.
"""
from typing import Tuple

import torch

from openeat_amd import ops
from openeat_amd.modules.embedding import PositionalEncoding, RelPositionalEncoding
from openeat_amd.modules.encoder import Encoder
from openeat_amd.modules.label_smoothing_loss import LabelSmoothingLoss
from openeat_amd.utils.common import IGNORE_ID, add_sos_eos, th_accuracy
from openeat_amd.utils.mask import make_pad_mask, subsequent_mask


class LanguageModel(torch.nn.Module):
    def __init__(self, vocab_size: int, pos_enc_layer_type: str = "abs_pos", encoder_num_blocks: int = 6,
                 activation_type: str = "swish", macaron_style: bool = False, use_cnn_module: bool = False,
                 cnn_module_kernel: int = 15, causal: bool = False, lsm_weight: float = 0.1,
                 length_normalized_loss: bool = False, ignore_id: int = IGNORE_ID, autoregressive: bool = True,
                 d_model: int = 256, attention_heads: int = 4, linear_units: int = 1024, dropout_rate: float = 0.1):
        super().__init__()
        self.sos = self.eos = vocab_size - 1
        self.vocab_size = vocab_size
        self.ignore_id = ignore_id
        self.autoregressive = autoregressive
        self.embedding = torch.nn.Embedding(vocab_size, d_model)
        pos = {"abs_pos": PositionalEncoding, "rel_pos": RelPositionalEncoding}
        if pos_enc_layer_type not in pos:
            raise ValueError("unknown pos_enc_layer: " + pos_enc_layer_type)
        if pos_enc_layer_type == "rel_pos" or use_cnn_module:
            raise NotImplementedError("the LM runs plain self-attention under a causal mask (abs_pos, no conv module)")
        self.pos_encoding = pos[pos_enc_layer_type](d_model)
        self.encoder = Encoder(d_model, dropout_rate, attention_heads, linear_units, activation_type, macaron_style,
                               use_cnn_module, cnn_module_kernel, causal, num_blocks=encoder_num_blocks)
        self.proj_layer = torch.nn.Linear(d_model, vocab_size)
        self.criterion_att = LabelSmoothingLoss(size=vocab_size, padding_idx=ignore_id, smoothing=lsm_weight,
                                                normalize_length=length_normalized_loss)

    def _hidden(self, tokens: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """language_model.py:109-123 up to the projection: (B, L) int64 -> (B, L, d)."""
        L = tokens.size(1)
        mask = (~make_pad_mask(lengths, L)).unsqueeze(1).to(tokens.device)
        if self.autoregressive:
            mask = mask & subsequent_mask(L, device=tokens.device).unsqueeze(0)
        pos = self.pos_encoding.table(tokens.device, L)
        xs = ops.embed(tokens, self.embedding.weight, pos.reshape(L, -1).contiguous(), self.pos_encoding.xscale)
        if self.autoregressive:
            with ops.causal_self_attention():
                xs, _, _ = self.encoder(xs, mask, pos)
        else:
            xs, _, _ = self.encoder(xs, mask, pos)
        return xs

    def _forward_encoder(self, tokens: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """-> logits (B, L, V) (language_model.py:109-125)."""
        return ops.linear(self._hidden(tokens, lengths), self.proj_layer.weight, self.proj_layer.bias)

    def logits(self, tokens: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """Un-normalised scores (B, L, V): rescoring takes its log-probabilities from them one token at a time (ops.logprob_gather)."""
        return self._forward_encoder(tokens, lengths)

    def log_probs(self, tokens: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        """log_softmax of the logits: what attention rescoring indexes as lm_output[i][j][w] (asr_model.py:498-499)."""
        return ops.log_softmax_rows(self._forward_encoder(tokens, lengths))

    def forward(self, input_targets: torch.Tensor, output_targets: torch.Tensor,
                targets_length: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """language_model.py:69-107."""
        assert targets_length.dim() == 1, targets_length.shape
        assert input_targets.shape[0] == targets_length.shape[0], (input_targets.shape, targets_length.shape)
        ops.predrop_clear()
        if self.autoregressive:
            ys_in, ys_out = add_sos_eos(input_targets, self.sos, self.eos, self.ignore_id)
            in_lens = targets_length + 1
        else:
            ys_in = input_targets.masked_fill(input_targets == self.ignore_id, self.eos)
            ys_out, in_lens = output_targets, targets_length
        ys_in, ys_out = ys_in.long(), ys_out.long()
        logits = self._forward_encoder(ys_in, in_lens)
        loss = self.criterion_att(logits, ys_out)
        acc = th_accuracy(logits.view(-1, self.vocab_size), ys_out, ignore_label=self.ignore_id)
        return loss, acc
