#!/usr/bin/env python3
"""csrc/gemm_hyb.hip (weight operand pre-split, activation split on the fragment) per shape and tile against the ring kernel
(both operands fp32): back-to-back launches on warm operands.  OE_HYB_TILE is read once per process, so tiles are compared by
running this tool once per setting:  OE_HYB_TILE=22|24|21 python tools/hyb_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, planes  # noqa: E402

hip.GEMM_PRECISION = 6
dev = "cuda"


def timeit(f, reps=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"OE_HYB_TILE={os.environ.get('OE_HYB_TILE', '(auto)')}")
print(f"{'shape':18s} {'kind':4s} {'m':>6s} {'n':>6s} {'k':>6s} {'hyb us':>8s} {'TF/s':>7s} {'ring us':>8s} {'TF/s':>7s}")
for name, kind, m, n, k in (("ffn.w1 fwd", "nt", 7936, 1024, 256), ("ffn.w2 fwd", "nt", 7936, 256, 1024), ("qkv fwd", "nt", 7936, 768, 256),
                            ("pw1 fwd", "nt", 7936, 512, 256), ("ctc logits", "nt", 7936, 3246, 256), ("ffn.w2 dgrad", "nn", 7936, 1024, 256),
                            ("north w1 fwd", "nt", 25472, 1024, 256), ("north w2 fwd", "nt", 25472, 256, 1024), ("north w2 dgrad", "nn", 25472, 1024, 256),
                            ("north w1 dgrad", "nn", 25472, 256, 1024), ("c5 w1 fwd", "nt", 12000, 2048, 512)):
    a = torch.randn(m, k, device=dev)
    if kind == "nt":
        w = torch.randn(n, k, device=dev)
        kw = dict(lda=k, ldb=k, ldc=n)
    else:
        w = torch.randn(k, n, device=dev)                    # (reduction, output): k-major B
        kw = dict(lda=k, ldb=n, ldc=n, b_kmajor=True)
    c = torch.empty(m, n, device=dev)
    wp = planes.of(w, force=True)
    th = timeit(lambda: hip.gemm(a, w, c, m, n, k, b_planes=wp, **kw))
    tr = timeit(lambda: hip.gemm(a, w, c, m, n, k, **kw))
    fl = 2.0 * m * n * k
    print(f"{name:18s} {kind:4s} {m:6d} {n:6d} {k:6d} {th:8.1f} {fl / th / 1e6:7.1f} {tr:8.1f} {fl / tr / 1e6:7.1f}")
