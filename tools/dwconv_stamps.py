#!/usr/bin/env python3
"""Phase split of dwconv_glu_bwd_kernel at config-2 size (diagnostic library; GPU box)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

B, T, d, K = 32, 248, 256, 15
lib = C.CDLL(os.environ["OE_HIP_LIB"])
buf = torch.zeros(8 * 16 * 8, dtype=torch.int64, device="cuda")
assert lib.oe_debug_set_dw_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
a, dy = torch.randn(B, T, 2 * d, device="cuda"), torch.randn(B, T, d, device="cuda")
w = torch.randn(d, K, device="cuda")
da = torch.empty_like(a)
dw, db = torch.zeros(d, K, device="cuda"), torch.zeros(d, device="cuda")
ws = torch.empty(hip.lib().oe_dwconv_glu_bwd_workspace_floats(B, T, d, K), device="cuda")
for _ in range(3):
    hip.call("oe_dwconv_glu_bwd", a, dy, w, None, B, T, d, K, 0, da, dw, db, None, ws)
torch.cuda.synchronize()
s = buf.view(128, 8).cpu().double()
names = ["staging (issue + LDS stores)", "wait at barrier", "weights load", "GLU-input preload", "frame loop", "partials store"]
med = [(s[:, i + 1] - s[:, i]).median().item() for i in range(5)]
print("total", (s[:, 5] - s[:, 0]).median().item())
for n, m in zip(["staging", "barrier", "weights+preload issue", "frame loop", "partials store"], med):
    print(f"{n:28s} {m:9.0f} cycles")
