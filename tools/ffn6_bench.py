#!/usr/bin/env python3
"""The precision-6 fused feed-forward (csrc/ffn6.hip) against the two-GEMM path of the same arithmetic, forward and input
gradient, warm (back to back) and cold (256 MB written between calls), HIP-event timed.  (GPU box.)
    python tools/ffn6_bench.py [rows d ff]..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, ops, planes  # noqa: E402

hip.GEMM_PRECISION = 6
dev = "cuda"
shapes = [(7936, 256, 1024), (25472, 256, 1024), (12000, 512, 2048), (992, 256, 1024)]
if len(sys.argv) > 3:
    v = [int(a) for a in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
big = torch.empty(64 << 20, device=dev)


def timed(f, cold, n=12):
    for _ in range(3):
        f()
    ts = []
    for _ in range(n):
        if cold:
            big.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for rows, d, ff in shapes:
    torch.manual_seed(0)
    x, res = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
    w1, b1 = torch.randn(ff, d, device=dev) / 16, torch.randn(ff, device=dev)
    w2, b2 = torch.randn(d, ff, device=dev) / 32, torch.randn(d, device=dev)
    dy = torch.randn(rows, d, device=dev)
    nb = hip.lib().oe_ffn_packed_bytes(d, ff, 6)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
    w2tp, w1tp = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
    pre, a, y = torch.empty(rows, ff, device=dev), torch.empty(rows, ff, device=dev), torch.empty(rows, d, device=dev)
    dh, dx = torch.empty(rows, ff, device=dev), torch.empty(rows, d, device=dev)

    def pack():
        hip.call("oe_ffn_pack_weights", w1, w2, d, ff, 6, w1p, w2p)

    def pack_bwd():
        hip.call("oe_ffn_pack_weights_bwd", w1, w2, d, ff, 6, w2tp, w1tp)

    def fused(nout):
        hip.ffn_fwd(x, w1p, b1, w2p, b2, rows, d, ff, 2, drop_in=0.1, seed_in=1, drop_out=0.1, seed_out=2, pre_out=pre if nout >= 1 else None,
                    act_out=a if nout == 2 else None, residual=res, ldr=d, beta=0.5, y=y)

    def unfused():
        aa = ops.gemm_nt(x, w1, b1, act=2, preact_out=pre, ld_aux=ff, drop_p=0.1, seed=1)
        ops.gemm_nt(aa, w2, b2, drop_p=0.1, seed=2, residual=res, ldr=d, beta=0.5)

    def fused_bwd():
        hip.ffn_bwd(dy, w2tp, w1tp, rows, d, ff, 2, drop_in=0.1, seed_in=1, pre=pre, dh=dh, dx=dx)

    def unfused_bwd():
        g = ops.gemm_nn(dy, w2, act=2, actgrad_in=pre, ld_aux=ff, drop_p=0.1, seed=1)
        ops.gemm_nn(g, w1)

    pack(); pack_bwd(); fused(2)
    flops = 4.0 * rows * d * ff
    print(f"rows {rows} d {d} ff {ff}  ({flops / 1e9:.2f} GFLOP each way; two-GEMM path: {planes.POLICY} policy, HYB_MIN_ROWS {planes.HYB_MIN_ROWS})")
    for cold in (False, True):
        tag = "cold" if cold else "warm"
        row = []
        for name, f in (("pack", pack), ("fused fwd nout=2", lambda: fused(2)), ("fused fwd nout=1", lambda: fused(1)), ("fused fwd nout=0", lambda: fused(0)),
                        ("two GEMMs fwd", unfused), ("pack bwd", pack_bwd), ("fused bwd", fused_bwd), ("two GEMMs bwd", unfused_bwd)):
            us = timed(f, cold)
            row.append(f"{name} {us:7.1f} us" + (f" ({flops / us / 1e6:5.0f} TF/s)" if "pack" not in name else ""))
        print(f"  {tag}: " + " | ".join(row), flush=True)
