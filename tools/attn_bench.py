#!/usr/bin/env python3
"""Attention forward / backward kernels alone at the config-2 encoder shape (B=32, H=4, T=248, D=64, dropout 0.1, key mask,
key bias): median HIP-event time per launch.  Run under rocprofv3 --kernel-trace for the per-kernel split."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, H, T, D = 32, 4, 248, 64
dev = "cuda"
q, k, v, do = (torch.randn(B, T, H, D, device=dev) for _ in range(4))
out, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
lse, delta = torch.empty(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
mask = torch.ones(B, 1, T, dtype=torch.uint8, device=dev)
mask[:, :, 230:] = 0
kbias, dkb = torch.randn(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
st = (T * H * D, H * D)
kw = dict(q_strides=st, k_strides=st, v_strides=st, o_strides=st, mask=mask, mask_strides=(T, 0), keybias=kbias, drop_p=0.1, seed=1,
          precision=prec)
af = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), **kw)
ab = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), d_out=do, dq=dq, dk=dk, dv=dv, dkeybias=dkb, delta=delta, **kw)


def med(f, n=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for _ in range(3):
        f()
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3


print(f"precision {prec}: forward {med(lambda: hip.attention_fwd(af)):.1f} us   backward (delta + dQ + dK/dV) {med(lambda: hip.attention_bwd(ab)):.1f} us")
