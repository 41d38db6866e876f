#!/usr/bin/env python3
"""Probe: the feed-forward weight gradients (1024 x 256 and 256 x 1024 outputs over K = 7936) one launch each (16-way split, atomics)
against grouped launches of n of them at several splits of the reduction."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openeat_amd import hip  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
dev = "cuda"
torch.manual_seed(0)
NP = 16
dys = [torch.randn(K, 1024 if i % 2 == 0 else 256, device=dev) for i in range(NP)]
xs = [torch.randn(K, 256 if i % 2 == 0 else 1024, device=dev) for i in range(NP)]
outs = [torch.zeros(d.shape[1], x.shape[1], device=dev) for d, x in zip(dys, xs)]
flops = sum(2.0 * K * d.shape[1] * x.shape[1] for d, x in zip(dys, xs))


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def single():
    for d, x, o in zip(dys, xs, outs):
        m, n = d.shape[1], x.shape[1]
        hip.gemm(d, x, o, m, n, K, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=16, atomic_out=True, precision=6)


t = timeit(single)
print(f"one launch each ({NP} launches): {t:8.1f} us  {flops / t / 1e6:6.1f} TFLOP/s")
ref = None
for o in outs:
    o.zero_()
single()
ref = [o.clone() for o in outs]
for n in (4, 8, 16):
    for target in (256, 512, 1024, 2048):
        descs = [dict(dy=dys[i], x=xs[i], out=outs[i], alpha=1.0, alpha_dev=None, bias_out=None) for i in range(n)]
        host, total = hip.tn_grouped_plan(descs, target)
        table = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
        f = lambda: hip.tn_grouped_launch(table, n, total, precision=6)
        t = timeit(f)
        fl = flops * n / NP
        for o in outs:
            o.zero_()
        f()
        torch.cuda.synchronize()
        err = max(float((outs[i] - ref[i]).abs().max()) for i in range(n))
        print(f"grouped n={n:2d} target {target:5d} -> {total:5d} blocks: {t:8.1f} us  {t / n:6.1f} us each  {fl / t / 1e6:6.1f} TFLOP/s   max diff {err:.2e}")
