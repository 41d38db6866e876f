#!/usr/bin/env python3
"""Activation-gradient GEMMs with a deep reduction and a narrow output (dx = dlogits @ W through a vocabulary
projection: M x 256 out of K = 3246): time vs split-K with atomic accumulation (memset of the output included)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

dev = "cuda"
V, Vp, d = 3246, 3248, 256
for M in (7936, 992, 25472, 3136):
    dl = torch.randn(M, Vp, device=dev)
    w = torch.randn(V, d, device=dev)
    c = torch.zeros(M, d, device=dev)
    row = []
    for sk in (1, 2, 3, 4, 6, 8, 12):
        def f():
            if sk > 1:
                c.zero_()
            hip.gemm(dl, w, c, M, d, V, lda=Vp, ldb=d, ldc=d, b_kmajor=True, split_k=sk, atomic_out=sk > 1, precision=3)
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
    print(f"nn out {M}x{d} k={V}:  " + "  ".join(row), flush=True)
