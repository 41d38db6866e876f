#!/usr/bin/env python3
"""Phase split of the attention forward kernel (diagnostic library; GPU box).
Build: OE_DIAG=1 bash openeat_amd/csrc/build.sh ; run with OE_HIP_LIB=openeat_amd/lib/libopeneat_hip_diag.so"""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, H, T, D = 32, 4, 248, 64
lib = C.CDLL(os.environ["OE_HIP_LIB"])
buf = torch.zeros(32 * 4 * 8, dtype=torch.int64, device="cuda")
assert lib.oe_debug_set_attn_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
q, k, v = (torch.randn(B, T, H, D, device="cuda") for _ in range(3))
out = torch.empty_like(q)
lse = torch.empty(B, H, T, device="cuda")
st = (T * H * D, H * D)
a = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), q_strides=st, k_strides=st, v_strides=st, o_strides=st,
                  drop_p=0.1, seed=1, precision=prec)
planes = hasattr(lib, "oe_debug_set_attn_planes_stamp_buffer") and os.environ.get("OE_ATTN_PLANES", "1") != "0"
if planes:
    buf2 = torch.zeros(32 * 4 * 12, dtype=torch.int64, device="cuda")
    assert lib.oe_debug_set_attn_planes_stamp_buffer(C.c_void_p(buf2.data_ptr())) == 0
for _ in range(3):
    hip.attention_fwd(a)
torch.cuda.synchronize()
if planes:
    s2 = buf2.view(128, 12).cpu().double()
    names2 = ["Q fragments + first prefetch issue", "first chunk lands + written + barrier", "prefetch issue", "K row frags + S MFMA issue",
              "softmax (+ S wait) + O rescale", "dropout", "P split + V col frags + PV MFMA issue", "MFMA drain", "commit next chunk",
              "barrier", "total"]
    med2 = s2.median(0).values
    for n, m in zip(names2, med2):
        print(f"{n:40s} {m:9.0f} cycles ({100 * m / med2[10]:5.1f} %)")
    sys.exit(0)
s = buf.view(128, 8).cpu().double()
names = ["prologue", "wait barrier 1", "tile store + barrier 2", "prefetch issue", "S frags + MFMA issue", "softmax (+S wait)",
         "dropout + PV issue + loop tail", "total"]
med = s.median(0).values
for n, m in zip(names, med):
    print(f"{n:32s} {m:9.0f} cycles ({100 * m / med[7]:5.1f} %)")
