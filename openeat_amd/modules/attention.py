"""Multi-head attention layers (/root/reference/openeat/modules/attention.py).
Projections, scores, mask, softmax, dropout and the context product run as
fused HIP kernels; the (B,H,T1,T2) score tensor is never materialised."""
from typing import Optional, Tuple

import torch
from torch import nn

from openeat_amd import ops


class MultiHeadedAttention(nn.Module):
    """attention.py:14-117."""

    def __init__(self, n_head: int, n_feat: int, dropout_rate: float):
        super().__init__()
        assert n_feat % n_head == 0
        self.d_k = n_feat // n_head
        self.h = n_head
        self.linear_q = nn.Linear(n_feat, n_feat)
        self.linear_k = nn.Linear(n_feat, n_feat)
        self.linear_v = nn.Linear(n_feat, n_feat)
        self.linear_out = nn.Linear(n_feat, n_feat)
        self.dropout = nn.Dropout(p=dropout_rate)

    def _run(self, query, key, value, mask, pos_emb, residual, out_dropout):
        if key is not value:
            raise NotImplementedError("key and value must be the same tensor (true for every call site of the path)")
        xkv = None if query is key else key
        p = self.dropout.p if self.training else 0.0
        rel = isinstance(self, RelPositionMultiHeadedAttention)
        pp = None
        ahead = getattr(self, "_pp_ahead", None)                  # (linear_pos(pos_emb), event) from the encoder's side stream
        if rel and ahead is not None:
            pp, ev = ahead
            torch.cuda.current_stream().wait_event(ev)
            pp.record_stream(torch.cuda.current_stream())
        return ops.attention(query, xkv, self.linear_q.weight, self.linear_q.bias, self.linear_k.weight,
                             self.linear_k.bias, self.linear_v.weight, self.linear_v.bias, self.linear_out.weight,
                             self.linear_out.bias, mask, self.h, p,
                             pos_emb if rel else None, self.linear_pos.weight if rel else None,
                             self.pos_bias_u if rel else None, self.pos_bias_v if rel else None,
                             residual, out_dropout if self.training else 0.0, pp)

    def forward_qkv(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """attention.py:36-63: the three projections as (B, h, T, d_k) views (head axis transposed forward, as the
        reference returns them).  Module-API surface; forward() runs the fused projection + attention kernels."""
        B = query.size(0)
        lin = lambda m, x: ops.linear(x, m.weight, m.bias).view(B, -1, self.h, self.d_k).transpose(1, 2)
        return lin(self.linear_q, query), lin(self.linear_k, key), lin(self.linear_v, value)

    def forward_attention(self, value: torch.Tensor, scores: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
        """attention.py:65-97 on a materialised (B, h, T1, T2) score tensor: masked softmax, 0-fill, dropout, @ value,
        merge heads, linear_out.  value (B, h, T2, d_k); mask (B,1,T2) or (B,T1,T2) or None."""
        x = ops.scores_attention(value, scores, mask, self.dropout.p if self.training else 0.0)
        return ops.linear(x, self.linear_out.weight, self.linear_out.bias)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, mask: Optional[torch.Tensor],
                pos_emb: Optional[torch.Tensor] = None, residual: torch.Tensor = None,
                out_dropout: float = 0.0) -> torch.Tensor:
        """mask: (B,1,T2) or (B,T1,T2), non-zero = attend.  ``residual``/``out_dropout`` fuse the
        caller's ``residual + dropout(.)`` into the output projection."""
        return self._run(query, key, value, mask, None, residual, out_dropout)


class RelPositionMultiHeadedAttention(MultiHeadedAttention):
    """attention.py:120-209 (rel_shift disabled there, :202-204)."""

    def __init__(self, n_head, n_feat, dropout_rate):
        super().__init__(n_head, n_feat, dropout_rate)
        self.linear_pos = nn.Linear(n_feat, n_feat, bias=False)
        self.pos_bias_u = nn.Parameter(torch.Tensor(self.h, self.d_k))
        self.pos_bias_v = nn.Parameter(torch.Tensor(self.h, self.d_k))
        torch.nn.init.xavier_uniform_(self.pos_bias_u)
        torch.nn.init.xavier_uniform_(self.pos_bias_v)

    def rel_shift(self, x: torch.Tensor, zero_triu: bool = False) -> torch.Tensor:
        """attention.py:140-164 (unused by forward there, :202-204, and here): row i of the (T1, T2) score matrix moves
        T1 - 1 - i places to the left through the flattened matrix.  Pure re-indexing, no arithmetic: one zero column in
        front, the buffer re-read with the two trailing sizes swapped, first row dropped."""
        b, h, t1, t2 = x.shape
        padded = torch.nn.functional.pad(x, (1, 0))                       # (b, h, t1, t2 + 1)
        x = padded.reshape(b, h, t2 + 1, t1)[:, :, 1:].reshape(b, h, t1, t2)
        if zero_triu:
            keep = torch.ones(t1, t2, device=x.device, dtype=x.dtype).tril(t2 - t1)
            x = x * keep
        return x

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, mask: Optional[torch.Tensor],
                pos_emb: torch.Tensor, residual: torch.Tensor = None, out_dropout: float = 0.0):
        if query is not key:
            raise NotImplementedError("relative-position attention is self-attention on this path")
        return self._run(query, key, value, mask, pos_emb, residual, out_dropout)
