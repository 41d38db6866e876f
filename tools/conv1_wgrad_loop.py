#!/usr/bin/env python3
"""oe_conv1_wgrad at config-2 size in a loop (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

B, T, F, C = 32, 998, 80, 256
T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
x = torch.randn(B, T, F, device="cuda")
dy = torch.randn(B, T1, F1, C, device="cuda")
dw, db = torch.zeros(C, 9, device="cuda"), torch.zeros(C, device="cuda")
for _ in range(10):
    hip.call("oe_conv1_wgrad", x, dy, B, T, F, C, dw, db)
torch.cuda.synchronize()
w1, b1 = torch.randn(C, 9, device="cuda"), torch.randn(C, device="cuda")
y = torch.empty(B, T1, F1, C, device="cuda")
for _ in range(10):
    hip.call("oe_conv1_fwd", x, w1, b1, B, T, F, C, y)
torch.cuda.synchronize()
# event timing (cold operands: a 1 GB buffer is written between calls so neither dy nor x sits in the Infinity Cache)
flush = torch.empty(256 * 1024 * 1024, device="cuda")
ts = []
for _ in range(6):
    flush.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.call("oe_conv1_wgrad", x, dy, B, T, F, C, dw, db)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
nbytes = dy.numel() * 4 + x.numel() * 4
print(f"oe_conv1_wgrad cold: median {sorted(ts)[len(ts) // 2]:.1f} us = {nbytes / sorted(ts)[len(ts) // 2] / 1e6:.2f} TB/s of dy + x ({nbytes / 1e6:.0f} MB)")
