#!/usr/bin/env python3
"""Row-block GEMM and fused feed-forward, HIP-event timed back to back (warm) and behind a 512 MB fill (cold), for the library in
OE_HIP_LIB - run once with the shipped library (weight ring of 4 register sets of one 3 KiB piece: 3 stages = 72 KiB per block in flight) and once with
builds of ffn6.hip with other ring shapes, e.g. -DOE_F6_FR=2 -DOE_F6_NSET=4 (the round's first choice: 144 KiB in flight).  Results are
identical across shapes (same order of the reduction; tests/test_gpu_rowgemm6.py, test_gpu_ffn6.py hold the shipped one).  (GPU box.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from openeat_amd import hip  # noqa: E402

hip.GEMM_PRECISION = 6
dev = "cuda"
torch.manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(f, cold, n=15):
    for _ in range(3):
        f()
    ts = []
    for i in range(n):
        if cold:
            flush.fill_(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


rows = 7936
for n, k in ((256, 256), (512, 256), (256, 512), (768, 256)):
    x, res = torch.randn(rows, k, device=dev), torch.randn(rows, n, device=dev)
    w, b = torch.randn(n, k, device=dev) / 16, torch.randn(n, device=dev)
    y = torch.empty(rows, n, device=dev)
    wp = torch.empty(n * k * 6, dtype=torch.uint8, device=dev)
    table = torch.tensor([w.data_ptr(), wp.data_ptr(), n, k, k, 0], dtype=torch.int64, device=dev)
    hip.call("oe_rowgemm6_pack_table", table, 1, (n // 32) * (k // 16))
    f = lambda: hip.rowgemm6(x, wp, y, rows, k, n, bias=b, residual=res, ldr=n, beta=1.0, drop_p=0.1, seed=3)
    print(f"rowgemm6 {rows} x {n} <- {k}: warm {timed(f, False):6.1f} us  cold {timed(f, True):6.1f} us   (weights {n * k * 6 / 1024:.0f} KiB per block)")
d, ff = 256, 1024
x, res = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
w1, b1 = torch.randn(ff, d, device=dev) / 16, torch.randn(ff, device=dev)
w2, b2 = torch.randn(d, ff, device=dev) / 32, torch.randn(d, device=dev)
nb = hip.lib().oe_ffn_packed_bytes(d, ff, 6)
w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
hip.call("oe_ffn_pack_weights", w1, w2, d, ff, 6, w1p, w2p)
pre, a, y = torch.empty(rows, ff, device=dev), torch.empty(rows, ff, device=dev), torch.empty(rows, d, device=dev)
f = lambda: hip.ffn_fwd(x, w1p, b1, w2p, b2, rows, d, ff, 2, drop_in=0.1, seed_in=5, drop_out=0.1, seed_out=7, pre_out=pre, act_out=a, residual=res, ldr=d,
                        beta=0.5, y=y)
print(f"ffn6 forward {rows} x {d} x {ff}: warm {timed(f, False):6.1f} us  cold {timed(f, True):6.1f} us   (weights {2 * d * ff * 6 / 1024:.0f} KiB per block)")
