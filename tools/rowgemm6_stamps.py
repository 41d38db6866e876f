#!/usr/bin/env python3
"""Where one block of the row-block GEMM (csrc/ffn6.hip::rowgemm6_kernel) spends its cycles: s_memtime stamps of the first and the
last block's waves, from the DIAGNOSTIC library (OE_DIAG=1 bash openeat_amd/csrc/build.sh).  (GPU box.)
    OE_HIP_LIB=openeat_amd/lib/libopeneat_hip_diag.so python tools/rowgemm6_stamps.py [rows [n [k]]]"""
import ctypes
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
k = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda"
torch.manual_seed(0)
x, res = torch.randn(rows, k, device=dev), torch.randn(rows, n, device=dev)
w, b = torch.randn(n, k, device=dev) / 16, torch.randn(n, device=dev)
y = torch.empty(rows, n, device=dev)
L = hip.lib()
wp = torch.empty(n * k * 6, dtype=torch.uint8, device=dev)
table = torch.tensor([w.data_ptr(), wp.data_ptr(), n, k, k, 0], dtype=torch.int64, device=dev)
hip.call("oe_rowgemm6_pack_table", table, 1, (n // 32) * (k // 16))
stamps = torch.zeros(2 * 8 * 128, dtype=torch.int64, device=dev)
L.oe_ffn6_set_stamps.restype = ctypes.c_int
L.oe_ffn6_set_stamps.argtypes = [ctypes.c_void_p]
assert L.oe_ffn6_set_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for it in range(4):
    if it >= 2:
        flush.fill_(it)                       # cold operands, as in the step
    hip.rowgemm6(x, wp, y, rows, k, n, bias=b, residual=res, ldr=n, beta=1.0, drop_p=0.1, seed=3)
    torch.cuda.synchronize()
    s = stamps.cpu().view(2, 8, 128)
    print(f"--- launch {it} ({'cold' if it >= 2 else 'warm'} operands), rows {rows} n {n} k {k}; cycles since the block's first stamp")
    nch = max(1, n // 128 // 2)
    for blk in range(2):
        t00 = int(s[blk, :, 0].min())
        for wv in range(8):
            t = s[blk, wv]
            line = f"  block {'first' if blk == 0 else 'last '} wave {wv}: start {int(t[0]) - t00:6d} | prologue+rows {int(t[1] - t[0]):6d} | barrier {int(t[2] - t[1]):6d}"
            prev = int(t[2])
            for c in range(nch):
                line += f" | mma {int(t[3 + 3 * c]) - prev:6d} patch {int(t[4 + 3 * c] - t[3 + 3 * c]):5d} store {int(t[5 + 3 * c] - t[4 + 3 * c]):5d}"
                prev = int(t[5 + 3 * c])
            line += f" | total {prev - t00:6d}"
            print(line)
    print(f"  last block's first stamp - first block's first stamp: {int(s[1, :, 0].min()) - int(s[0, :, 0].min())} cycles")
