#!/usr/bin/env python3
"""Weight-gradient GEMMs (both operands k-major) of the training step, one by one: the bf16-planes kernel (gemm_tn.hip)
against the kernels it replaces (OE_GEMM_TN_PLANES=0: LDS-DMA ring / register-staged), one child process per arm (the
switch is read once per process).  Back-to-back launches between one pair of HIP events, split-K as ops._split_k asks.

    python tools/tn_bench.py [precision]"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(1024, 256, 7936), (256, 1024, 7936), (256, 256, 7936), (512, 256, 7936), (768, 256, 7936), (3246, 256, 7936),
          (256, 4864, 7936), (256, 256, 992), (1024, 256, 992), (3246, 256, 992), (1024, 256, 25472), (256, 256, 25472),
          (2048, 512, 47744), (512, 512, 47744), (1792, 256, 7936)]


def run(prec):
    from openeat_amd import hip
    from openeat_amd.ops import _split_k
    hip.GEMM_PRECISION = prec
    dev = "cuda"
    for (m, n, k) in SHAPES:
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
        c = torch.zeros(m, n, device=dev)
        db = torch.zeros(m, device=dev)
        sk = _split_k(m, n, k)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True,
                             a_colsum=db, precision=prec)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        us = sorted(ts)[2]
        print(f"  dW {m:5d} x {n:5d}  K = {k:6d}  split asked {sk:3d}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        run(int(sys.argv[2]))
    else:
        prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
        for arm, label in (("2", "bf16 planes, two K-groups per block (gemm_tn.hip), forced wherever it qualifies"),
                           ("1", "default dispatch"), ("0", "previous kernels (LDS-DMA ring / register-staged)")):
            print(f"precision {prec}: {label}", flush=True)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", str(prec)], env=dict(os.environ, OE_GEMM_TN_PLANES=arm))
