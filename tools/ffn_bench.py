#!/usr/bin/env python3
"""The fused feed-forward kernel (csrc/ffn.hip) alone at the config-2 encoder shape (7936 rows, d = 256, ff = 1024, swish,
dropout 0.1 on both sides) next to the two-GEMM path: run under tools/kprof.sh for per-kernel durations.
    python tools/ffn_bench.py [precision] [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, ops  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 7936
d, ff = 256, 1024
hip.GEMM_PRECISION = prec
dev = "cuda"
torch.manual_seed(0)
x, res = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
w1, b1 = torch.randn(ff, d, device=dev) / 16, torch.randn(ff, device=dev)
w2, b2 = torch.randn(d, ff, device=dev) / 32, torch.randn(d, device=dev)
nb = hip.lib().oe_ffn_packed_bytes(d, ff, prec)
w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
pre, a, y = torch.empty(rows, ff, device=dev), torch.empty(rows, ff, device=dev), torch.empty(rows, d, device=dev)
big = torch.empty(64 << 20, device=dev)          # 256 MB: evicts the caches between calls, as the step's other kernels do


def fused(nout):
    hip.call("oe_ffn_pack_weights", w1, w2, d, ff, prec, w1p, w2p)
    hip.ffn_fwd(x, w1p, b1, w2p, b2, rows, d, ff, 2, drop_in=0.1, seed_in=1, drop_out=0.1, seed_out=2, pre_out=pre if nout >= 1 else None,
                act_out=a if nout == 2 else None, residual=res, ldr=d, beta=0.5, y=y)


def unfused():
    aa = ops.gemm_nt(x, w1, b1, act=2, preact_out=pre, ld_aux=ff, drop_p=0.1, seed=1)
    ops.gemm_nt(aa, w2, b2, drop_p=0.1, seed=2, residual=res, ldr=d, beta=0.5)


for it in range(6):
    for nout in (2, 0):
        big.zero_()
        fused(nout)
    big.zero_()
    unfused()
torch.cuda.synchronize()
print("done")
