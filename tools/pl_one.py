#!/usr/bin/env python3
"""One GEMM shape on gemm_pl.hip, a few launches (for rocprofv3 --pmc): python tools/pl_one.py kind m n k [waves] [bk] [tile]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, planes  # noqa: E402

kind, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
waves = int(sys.argv[5]) if len(sys.argv) > 5 else 8
bk = int(sys.argv[6]) if len(sys.argv) > 6 else 0
tile = int(sys.argv[7]) if len(sys.argv) > 7 else 0
hip.GEMM_PRECISION = 6
planes.POLICY, planes.MIN_SPLIT_ELEMS = "all", 0
hip.lib().oe_gemm_pl_config(0, tile, bk, waves)
dev = "cuda"
if kind == "nt":
    a, b, c = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
    kw = dict(lda=k, ldb=k, ldc=n)
elif kind == "nn":
    a, b, c = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
    kw = dict(lda=k, ldb=n, ldc=n, b_kmajor=True)
else:
    from openeat_amd.ops import _split_k
    a, b, c = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.zeros(m, n, device=dev)
    kw = dict(lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=_split_k(m, n, k), atomic_out=True)
ap, bp = planes.of(a, force=True), planes.of(b, force=True)
n0 = hip.lib().oe_gemm_pl_launches()
for _ in range(6):
    hip.gemm(a, b, c, m, n, k, precision=6, a_planes=ap, b_planes=bp, **kw)
torch.cuda.synchronize()
assert hip.lib().oe_gemm_pl_launches() - n0 == 6
