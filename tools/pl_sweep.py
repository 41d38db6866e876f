#!/usr/bin/env python3
"""gemm_pl.hip tile variants per shape: python tools/pl_sweep.py  (tile, waves, bk) forced through oe_gemm_pl_config."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, planes  # noqa: E402
from tools.pl_bench import timeit  # noqa: E402

M = int(os.environ.get("PL_M", "7936"))
SHAPES = [("ffn.w1 fwd", "nt", (M, 1024, 256)), ("ffn.w2 fwd", "nt", (M, 256, 1024)), ("qkv fwd", "nt", (M, 768, 256)),
          ("pw1 fwd", "nt", (M, 512, 256)), ("out fwd", "nt", (M, 256, 256)),
          ("ffn.w2 dgrad", "nn", (M, 1024, 256)), ("ffn.w1 dgrad", "nn", (M, 256, 1024)), ("qkv dgrad", "nn", (M, 256, 768))]
VARIANTS = [(22, 8, 16), (22, 8, 32), (22, 4, 16), (22, 4, 32), (24, 8, 0), (44, 8, 0), (11, 8, 0)]

if __name__ == "__main__":
    hip.GEMM_PRECISION = 6
    planes.POLICY, planes.MIN_SPLIT_ELEMS = "all", 0
    dev = "cuda"
    print(f"{'shape':14s} {'kind':3s} " + " ".join(f"{f't{t}w{w}k{b}':>10s}" for t, w, b in VARIANTS))
    for name, kind, (m, n, k) in SHAPES:
        a = torch.randn(m, k, device=dev)
        b = torch.randn(n, k, device=dev) if kind == "nt" else torch.randn(k, n, device=dev)
        c = torch.empty(m, n, device=dev)
        kw = dict(lda=k, ldb=k, ldc=n) if kind == "nt" else dict(lda=k, ldb=n, ldc=n, b_kmajor=True)
        ap, bp = planes.of(a, force=True), planes.of(b, force=True)
        out = []
        for t, w, bk in VARIANTS:
            hip.lib().oe_gemm_pl_config(0, t, bk, w)
            n0 = hip.lib().oe_gemm_pl_launches()
            try:
                us = timeit(lambda: hip.gemm(a, b, c, m, n, k, precision=6, a_planes=ap, b_planes=bp, **kw))
                used = hip.lib().oe_gemm_pl_launches() > n0
            except Exception:
                us, used = float("nan"), False
            out.append(f"{us:10.1f}" if used else f"{'-':>10s}")
        print(f"{name:14s} {kind:3s} " + " ".join(out), flush=True)
