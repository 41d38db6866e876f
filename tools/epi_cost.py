#!/usr/bin/env python3
"""What a feed-forward's first GEMM (7936 x 1024 x 256, precision 6) pays for its epilogue and for cold operands: the plain product
against + bias + swish + pre-activation copy against + dropout, on warm operands (back to back) and with 1 GB written between calls."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

hip.GEMM_PRECISION = 6
dev = "cuda"
M, N, K = 7936, 1024, 256
x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.randn(N, device=dev)
y, pre = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
res = torch.randn(M, K, device=dev)
y2 = torch.empty(M, K, device=dev)
w2 = torch.randn(K, N, device=dev) * 0.1
flush = torch.empty(256 * 1024 * 1024, device=dev)


def timeit(f, cold, reps=20):
    ts = []
    for _ in range(reps):
        if cold:
            flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


cases = [
    ("W1: plain product", lambda: hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N)),
    ("W1: + bias + swish", lambda: hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2)),
    ("W1: + pre-activation copy", lambda: hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2, preact_out=pre, ld_aux=N)),
    ("W1: + dropout 0.1 (the step's call)", lambda: hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2, preact_out=pre, ld_aux=N, drop_p=0.1, seed=7)),
    ("W2: plain product (7936 x 256 x 1024)", lambda: hip.gemm(y, w2, y2, M, K, N, lda=N, ldb=N, ldc=K)),
    ("W2: + bias + dropout + residual (the step's call)", lambda: hip.gemm(y, w2, y2, M, K, N, lda=N, ldb=N, ldc=K, bias=b[:K], drop_p=0.1, seed=9, residual=res, ldr=K, beta=0.5)),
]
print(f"{'case':52s} {'warm us':>8s} {'cold us':>8s}   (event pair overhead ~4.7 us included in both)")
for name, f in cases:
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    print(f"{name:52s} {timeit(f, False):8.1f} {timeit(f, True):8.1f}")
