"""Independent alpha/beta restatement of CTC loss + gradient (oracle; test-only).

The reference calls ``torch.nn.CTCLoss(reduction='sum', zero_infinity=True)``
on ``log_softmax`` outputs and divides by the batch size
(``/root/reference/openeat/modules/ctc.py:27-45``).  The arithmetic of that
loss lives in PyTorch (aten ``_ctc_loss`` / ``_ctc_loss_backward``; PyTorch
2.10.0 in this image), not in the reference tree.  This file restates the
published algorithm (Graves et al. 2006, eq. 6-16, log domain) in numpy
float64 so the tests can pin both aten's result and the HIP kernel against a
third, loop-level implementation.

Conventions: blank = 0; extended label sequence l' = [0, y1, 0, y2, ..., 0] of
S = 2L+1 states; gradient is taken w.r.t. the *logits* (log_softmax folded in):

    d nll_b / d logit[b,t,c] = softmax[b,t,c] - (1/P_b) * sum_{s: l'_s = c} a_t(s) b_t(s) / y_t(c)

for t < hlen_b, and 0 for padded frames and for infeasible utterances
(zero_infinity).  beta here includes the emission at t, so alpha*beta counts
y_t(l'_s) twice - hence the division.
"""
from __future__ import annotations

import numpy as np

NEG = -np.inf


def _lse(*xs):
    m = max(xs)
    if m == NEG:
        return NEG
    return m + np.log(sum(np.exp(x - m) for x in xs))


def ctc_nll_and_grad(logits: np.ndarray, hlens, targets: np.ndarray, tlens):
    """logits (B, T, V) float; targets (B, Lmax) int (pad anything); returns
    nll (B,) float64 with inf->0 applied (zero_infinity) and dlogits (B, T, V)
    = d(sum_b nll_b)/d logits (float64)."""
    logits = np.asarray(logits, dtype=np.float64)
    B, T, V = logits.shape
    m = logits.max(-1, keepdims=True)
    lse = m + np.log(np.exp(logits - m).sum(-1, keepdims=True))
    logp = logits - lse
    nll = np.zeros(B)
    grad = np.zeros_like(logits)
    for b in range(B):
        Tb, L = int(hlens[b]), int(tlens[b])
        ext = np.zeros(2 * L + 1, dtype=np.int64)
        ext[1::2] = targets[b, :L]
        S = ext.size
        if Tb == 0:
            # aten: no frames -> loss 0 if the target is empty, inf otherwise
            nll[b] = 0.0
            continue
        alpha = np.full((Tb, S), NEG)
        alpha[0, 0] = logp[b, 0, 0]
        if S > 1:
            alpha[0, 1] = logp[b, 0, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                a = alpha[t - 1, s]
                a1 = alpha[t - 1, s - 1] if s >= 1 else NEG
                a2 = alpha[t - 1, s - 2] if (s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]) else NEG
                alpha[t, s] = _lse(a, a1, a2) + logp[b, t, ext[s]]
        ll = _lse(alpha[Tb - 1, S - 1], alpha[Tb - 1, S - 2] if S > 1 else NEG)
        if ll == NEG:
            nll[b] = 0.0  # zero_infinity
            continue
        nll[b] = -ll
        beta = np.full((Tb, S), NEG)
        beta[Tb - 1, S - 1] = logp[b, Tb - 1, ext[S - 1]]
        if S > 1:
            beta[Tb - 1, S - 2] = logp[b, Tb - 1, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                c = beta[t + 1, s]
                c1 = beta[t + 1, s + 1] if s + 1 < S else NEG
                c2 = beta[t + 1, s + 2] if (s + 2 < S and ext[s + 2] != 0 and ext[s + 2] != ext[s]) else NEG
                beta[t, s] = _lse(c, c1, c2) + logp[b, t, ext[s]]
        for t in range(Tb):
            occ = np.full(V, NEG)
            for s in range(S):
                occ[ext[s]] = _lse(occ[ext[s]], alpha[t, s] + beta[t, s])
            grad[b, t] = np.exp(logp[b, t]) - np.exp(occ - ll - logp[b, t])
    return nll, grad
