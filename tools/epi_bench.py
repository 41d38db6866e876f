#!/usr/bin/env python3
"""Cost of the GEMM epilogue variants on the FFN shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

dev = "cuda"
M, FF, D = 7936, 1024, 256


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
x, w1, b1 = torch.randn(M, D, device=dev), torch.randn(FF, D, device=dev), torch.randn(FF, device=dev)
g2, w2 = torch.randn(M, D, device=dev), torch.randn(D, FF, device=dev)
pre, a, dh, res, y = (torch.randn(M, FF, device=dev) for _ in range(2)) if False else (None,) * 5
pre = torch.randn(M, FF, device=dev)
a = torch.empty(M, FF, device=dev)
dh = torch.empty(M, FF, device=dev)
res = torch.randn(M, D, device=dev)
y = torch.empty(M, D, device=dev)
cases = {
    "w1 fwd plain": lambda: hip.gemm(x, w1, a, M, FF, D, lda=D, ldb=D, ldc=FF, precision=prec),
    "w1 fwd +bias+swish+preact": lambda: hip.gemm(x, w1, a, M, FF, D, lda=D, ldb=D, ldc=FF, bias=b1, act=2, preact_out=pre, ld_aux=FF, precision=prec),
    "w1 fwd ... +dropout": lambda: hip.gemm(x, w1, a, M, FF, D, lda=D, ldb=D, ldc=FF, bias=b1, act=2, preact_out=pre, ld_aux=FF, drop_p=0.1, seed=7, precision=prec),
    "w2 fwd plain": lambda: hip.gemm(a, w2, y, M, D, FF, lda=FF, ldb=FF, ldc=D, precision=prec),
    "w2 fwd +residual+dropout": lambda: hip.gemm(a, w2, y, M, D, FF, lda=FF, ldb=FF, ldc=D, residual=res, ldr=D, beta=0.5, drop_p=0.1, seed=9, precision=prec),
    "dh plain": lambda: hip.gemm(g2, w2, dh, M, FF, D, lda=D, ldb=FF, ldc=FF, b_kmajor=True, precision=prec),
    "dh +actgrad": lambda: hip.gemm(g2, w2, dh, M, FF, D, lda=D, ldb=FF, ldc=FF, b_kmajor=True, act=2, actgrad_in=pre, ld_aux=FF, precision=prec),
    "dh +actgrad+dropout": lambda: hip.gemm(g2, w2, dh, M, FF, D, lda=D, ldb=FF, ldc=FF, b_kmajor=True, act=2, actgrad_in=pre, ld_aux=FF, drop_p=0.1, seed=7, precision=prec),
}
for k, f in cases.items():
    print(f"{k:32s} {timeit(f):8.1f} us", flush=True)
