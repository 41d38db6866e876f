#!/usr/bin/env python3
"""Phase split of the weight-gradient planes kernel (gemm_tn.hip), wave 0 of every block (GPU box, diagnostic library).
Build: OE_DIAG=1 bash openeat_amd/csrc/build.sh ; run with OE_HIP_LIB=openeat_amd/lib/libopeneat_hip_diag.so"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = C.CDLL(os.environ["OE_HIP_LIB"])
buf = torch.zeros(2048 * 8, dtype=torch.int64, device="cuda")
assert lib.oe_debug_set_tn_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
names = ["first chunk (load, split, store, barrier)", "load issue", "fragment reads + MFMA issue", "rest of the loads' latency",
         "split + LDS store", "barrier (+ MFMA drain)", "bias gradient + exchange + output", "total"]
for (m, n, k) in ((1024, 256, 7936), (256, 256, 7936), (3246, 256, 7936)):
    a, b = torch.randn(k, m, device="cuda"), torch.randn(k, n, device="cuda")
    c = torch.zeros(m, n, device="cuda")
    f = lambda: hip.gemm(a, b, c, m, n, k, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=16, atomic_out=True, precision=prec)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf.zero_()
    f()
    torch.cuda.synchronize()
    s = buf.view(2048, 8).cpu().double()
    s = s[s[:, 7] > 0]
    if s.shape[0] == 0:
        print(f"dW {m} x {n}, K = {k}: not on the planes kernel")
        continue
    med = s.median(0).values
    print(f"dW {m} x {n}, K = {k}: {s.shape[0]} blocks")
    for nm, v in zip(names, med):
        print(f"   {nm:44s} {v:9.0f} ticks ({100 * v / med[7]:5.1f} %)")
