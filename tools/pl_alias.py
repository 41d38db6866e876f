#!/usr/bin/env python3
"""Is gemm_pl.hip bound by where its tiles come from?  Same launch, three operand placements: real (rows spread over the
matrices), lda = ldb = 0 (every tile row aliases row 0: all DMA traffic is L1 / L2 hits; results are garbage, timing only).
    python tools/pl_alias.py m n k [waves] [bk] [tile]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, planes  # noqa: E402

m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 8
bk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tile = int(sys.argv[6]) if len(sys.argv) > 6 else 0
hip.GEMM_PRECISION = 6
planes.POLICY, planes.MIN_SPLIT_ELEMS = "all", 0
hip.lib().oe_gemm_pl_config(0, tile, bk, waves)
dev = "cuda"
a, b, c = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
ap, bp = planes.of(a, force=True), planes.of(b, force=True)


def t(lda, ldb, reps=20):
    f = lambda: hip.gemm(a, b, c, m, n, k, precision=6, a_planes=ap, b_planes=bp, lda=lda, ldb=ldb, ldc=n)
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


fl = 2.0 * m * n * k * 6 / 1e6 / 2500 * 100
for name, (la, lb) in (("real operands", (k, k)), ("A rows aliased (lda = 0)", (0, k)), ("B rows aliased (ldb = 0)", (k, 0)), ("both aliased", (0, 0))):
    us = t(la, lb)
    print(f"{name:28s} {us:8.1f} us   MFMA {fl / us:5.1f} %", flush=True)
