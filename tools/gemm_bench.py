#!/usr/bin/env python3
"""Per-shape GEMM micro-benchmark (GPU box): the shapes the 12L Conformer step launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

M = int(os.environ.get("GEMM_BENCH_M", "7936"))          # B=32 x T'=248 rows (25472 = the 64 x 16 s batch)
SHAPES = [
    # name, kind, (m, n, k)   kinds: nt = x W^T, nn = dy W, tn = dy^T x
    ("ffn.w1 fwd", "nt", (M, 1024, 256)),
    ("ffn.w2 fwd", "nt", (M, 256, 1024)),
    ("qkv fwd", "nt", (M, 768, 256)),
    ("attn.out fwd", "nt", (M, 256, 256)),
    ("pw1 fwd", "nt", (M, 512, 256)),
    ("ctc logits", "nt", (M, 3246, 256)),
    ("ffn.w2 dgrad", "nn", (M, 1024, 256)),
    ("ffn.w1 dgrad", "nn", (M, 256, 1024)),
    ("qkv dgrad", "nn", (M, 256, 768)),
    ("ffn.w1 wgrad", "tn", (1024, 256, M)),
    ("ffn.w2 wgrad", "tn", (256, 1024, M)),
    ("qkv wgrad", "tn", (768, 256, M)),
    ("out wgrad", "tn", (256, 256, M)),
    ("ctc wgrad", "tn", (3246, 256, M)),
    ("dec out (rows 992)", "nt", (992, 3246, 256)),
]


def run(kind, m, n, k, prec, reps=20):
    dev = "cuda"
    if kind == "nt":
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
        c = torch.empty(m, n, device=dev)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=k, ldb=k, ldc=n, precision=prec)
    elif kind == "nn":
        a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)      # dy (m,k) @ W (k,n)
        c = torch.empty(m, n, device=dev)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=k, ldb=n, ldc=n, b_kmajor=True, precision=prec)
    else:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)      # dy (K,m)^T x (K,n)
        c = torch.zeros(m, n, device=dev)
        from openeat_amd.ops import _split_k
        sk = _split_k(m, n, k)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True,
                             precision=prec)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * m * n * k / us / 1e6


if __name__ == "__main__":
    precs = [int(p) for p in (sys.argv[1] if len(sys.argv) > 1 else "0,3,1").split(",")]
    print(f"{'shape':22s} {'kind':3s} {'m':>6s} {'n':>6s} {'k':>6s} " + " ".join(f"{'p'+str(p)+' us':>9s} {'TF/s':>7s}" for p in precs))
    for name, kind, (m, n, k) in SHAPES:
        cells = []
        for p in precs:
            us, tf = run(kind, m, n, k, p)
            cells.append(f"{us:9.1f} {tf:7.1f}")
        print(f"{name:22s} {kind:3s} {m:6d} {n:6d} {k:6d} " + " ".join(cells), flush=True)
