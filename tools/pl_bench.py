#!/usr/bin/env python3
"""gemm_pl.hip (pre-split operands, precision 6) against the kernels that split fp32 operands in the loop, on the GEMM shapes
of the config-2 step; operands pre-split outside the timed region (in the step their producers write the planes).

    python tools/pl_bench.py            # OE_PL_TILE / OE_PL_BK / OE_PL_MIN_BLOCKS force a variant"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, planes  # noqa: E402

M = 7936
SHAPES = [("ffn.w1 fwd", "nt", (M, 1024, 256)), ("ffn.w2 fwd", "nt", (M, 256, 1024)), ("qkv fwd", "nt", (M, 768, 256)),
          ("out fwd", "nt", (M, 256, 256)), ("pw1 fwd", "nt", (M, 512, 256)), ("big nt", "nt", (M, 1024, 1024)),
          ("lin fwd", "nt", (M, 256, 4864)),
          ("ffn.w2 dgrad", "nn", (M, 1024, 256)), ("ffn.w1 dgrad", "nn", (M, 256, 1024)), ("qkv dgrad", "nn", (M, 256, 768)),
          ("out dgrad", "nn", (M, 256, 256)), ("lin dgrad", "nn", (M, 4864, 256)),
          ("ffn.w1 wgrad", "tn", (1024, 256, M)), ("ffn.w2 wgrad", "tn", (256, 1024, M)), ("qkv wgrad", "tn", (768, 256, M)),
          ("out wgrad", "tn", (256, 256, M)), ("lin wgrad", "tn", (256, 4864, M)),
          ("dec ffn (992)", "nt", (992, 1024, 256)), ("north nt", "nt", (25472, 1024, 256)), ("north tn", "tn", (1024, 256, 25472))]


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps * 1e3)
    return sorted(ts)[2]


def run(kind, m, n, k):
    dev = "cuda"
    from openeat_amd.ops import _split_k
    if kind == "nt":
        a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
        c = torch.empty(m, n, device=dev)
        kw = dict(lda=k, ldb=k, ldc=n)
    elif kind == "nn":
        a, b = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev)
        c = torch.empty(m, n, device=dev)
        kw = dict(lda=k, ldb=n, ldc=n, b_kmajor=True)
    else:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
        c = torch.zeros(m, n, device=dev)
        kw = dict(lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=_split_k(m, n, k), atomic_out=True)
    ap, bp = planes.of(a, force=True), planes.of(b, force=True)
    n0 = hip.lib().oe_gemm_pl_launches()
    t_pl = timeit(lambda: hip.gemm(a, b, c, m, n, k, precision=6, a_planes=ap, b_planes=bp, **kw))
    used = hip.lib().oe_gemm_pl_launches() > n0
    t_old = timeit(lambda: hip.gemm(a, b, c, m, n, k, precision=6, **kw))
    t_p3 = timeit(lambda: hip.gemm(a, b, c, m, n, k, precision=3, **kw))
    return t_pl, t_old, t_p3, used, kw.get("split_k", 1)


if __name__ == "__main__":
    hip.GEMM_PRECISION = 6
    planes.MIN_SPLIT_ELEMS = 0
    print(f"{'shape':16s} {'kind':3s} {'m':>6s} {'n':>6s} {'k':>6s} {'sk':>3s} {'planes us':>10s} {'TF/s':>7s} {'MFMA%':>6s} {'split-in-loop us':>17s} {'p3 us':>8s}")
    for name, kind, (m, n, k) in SHAPES:
        t_pl, t_old, t_p3, used, sk = run(kind, m, n, k)
        fl = 2.0 * m * n * k
        print(f"{name:16s} {kind:3s} {m:6d} {n:6d} {k:6d} {sk:3d} {t_pl:10.1f} {fl / t_pl / 1e6:7.1f} {6 * fl / t_pl / 1e6 / 2500 * 100:6.1f} {t_old:17.1f} {t_p3:8.1f}"
              + ("" if used else "   (planes kernel declined)"), flush=True)
