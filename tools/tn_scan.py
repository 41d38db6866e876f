#!/usr/bin/env python3
"""wgrad (TN) GEMM time vs split-K: latency-bound kernels speed up with more blocks in flight."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

dev = "cuda"
for (m, n, k) in ((1024, 256, 7936), (256, 1024, 7936), (256, 256, 7936)):
    a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
    c = torch.zeros(m, n, device=dev)
    row = []
    for sk in (1, 2, 4, 8, 12, 16, 24, 31):
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True, precision=1)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        row.append(f"sk={sk}:{e0.elapsed_time(e1) * 100:.1f}us")
    print(f"tn out {m}x{n} k={k}:  " + "  ".join(row), flush=True)
