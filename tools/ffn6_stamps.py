#!/usr/bin/env python3
"""Where one block of the precision-6 fused feed-forward (csrc/ffn6.hip) spends its cycles: s_memtime stamps of block 0's four
waves at the phase boundaries, from the DIAGNOSTIC library (OE_DIAG=1 bash openeat_amd/csrc/build.sh).  (GPU box.)
    OE_HIP_LIB=openeat_amd/lib/libopeneat_hip_diag.so python tools/ffn6_stamps.py [rows [mode]]   (mode: oe_ffn6_config)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
d, ff = 256, 1024
dev = "cuda"
torch.manual_seed(0)
x, res = torch.randn(rows, d, device=dev), torch.randn(rows, d, device=dev)
w1, b1 = torch.randn(ff, d, device=dev) / 16, torch.randn(ff, device=dev)
w2, b2 = torch.randn(d, ff, device=dev) / 32, torch.randn(d, device=dev)
nb = hip.lib().oe_ffn_packed_bytes(d, ff, 6)
w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=dev), torch.empty(nb, dtype=torch.uint8, device=dev)
pre, a, y = torch.empty(rows, ff, device=dev), torch.empty(rows, ff, device=dev), torch.empty(rows, d, device=dev)
hip.call("oe_ffn_pack_weights", w1, w2, d, ff, 6, w1p, w2p)
stamps = torch.zeros(2 * 8 * 128, dtype=torch.int64, device=dev)
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
NG = 1 if mode in (1, 2) else 2
L = hip.lib()
L.oe_ffn6_config(mode)
L.oe_ffn6_set_stamps.restype = ctypes.c_int
L.oe_ffn6_set_stamps.argtypes = [ctypes.c_void_p]
assert L.oe_ffn6_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
for _ in range(5):
    hip.ffn_fwd(x, w1p, b1, w2p, b2, rows, d, ff, 2, drop_in=0.1, seed_in=1, drop_out=0.1, seed_out=2, pre_out=pre, act_out=a, residual=res,
                ldr=d, beta=0.5, y=y, precision=6)
torch.cuda.synchronize()
s = stamps.cpu().view(2, 8, 128)[0]
nch = ff // 128 // NG
for w in range(4 * NG):
    t = s[w]
    t0 = int(t[0])
    print(f"wave {w}: x split -> {int(t[0]) - t0:6d} | barrier {int(t[1] - t[0]):6d}")
    g1 = e1 = bx = by = g2 = 0
    for c in range(nch):
        prev_end = int(t[1]) if c == 0 else int(t[6 + 5 * (c - 1)])
        a_ = int(t[2 + 5 * c]) - prev_end
        b_ = int(t[3 + 5 * c] - t[2 + 5 * c])
        c_ = int(t[4 + 5 * c] - t[3 + 5 * c])
        d_ = int(t[5 + 5 * c] - t[4 + 5 * c])
        e_ = int(t[6 + 5 * c] - t[5 + 5 * c])
        print(f"   chunk {c}: GEMM1 {a_:6d} | barrier X {b_:6d} | epilogue 1 {c_:6d} | barrier Y {d_:6d} | GEMM2 {e_:6d}")
        g1 += a_; bx += b_; e1 += c_; by += d_; g2 += e_
    end = int(t[3 + 5 * nch] - t[2 + 5 * nch])
    total = int(t[3 + 5 * nch]) - t0
    print(f"   sums: GEMM1 {g1} barrier X {bx} epilogue 1 {e1} barrier Y {by} GEMM2 {g2} | final barrier + epilogue 2 {end} | total after x split {total} cycles "
          f"(matrix-pipe floor of this wave {nch * (192 if mode != 1 else 384) * 32})")
