#!/usr/bin/env python3
"""LayerNorm forward / backward at the encoder shape (7936 x 256) in a loop (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

rows, d = 7936, 256
x, dy, add = (torch.randn(rows, d, device="cuda") for _ in range(3))
g, b = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
y, dx, gq = (torch.empty(rows, d, device="cuda") for _ in range(3))
stats = torch.empty(rows, 2, device="cuda")
ws = torch.empty(hip.lib().oe_layernorm_bwd_workspace_floats(rows, d), device="cuda")
for _ in range(20):
    hip.call("oe_layernorm_fwd", x, g, b, 1e-5, rows, d, None, 0, y, stats)
    hip.call("oe_layernorm_bwd_dx", dy, x, g, b, 0, stats, rows, d, None, add, dx, ws)
    hip.call("oe_layernorm_bwd_dx_drop", dy, x, g, b, 0, stats, rows, d, None, add, dx, gq, 0.5, 0.1, 7, None, None, ws)
torch.cuda.synchronize()
