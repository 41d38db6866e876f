#!/usr/bin/env python3
"""oe_relpos_prepare / oe_relpos_backward at the config-2 shape in a loop (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

B, T, H, D = 32, 248, 4, 64
d = H * D
dev = "cuda"
qkv = torch.randn(B * T, 3 * d, device=dev)
k = qkv[:, d:2 * d]
pp = torch.randn(T, d, device=dev)
pu, pv = torch.randn(H, D, device=dev), torch.randn(H, D, device=dev)
kp, kb = torch.empty(B, T, d, device=dev), torch.empty(B, H, T, device=dev)
dkp, dkb = torch.randn(B, T, d, device=dev), torch.randn(B, H, T, device=dev)
dqkv = torch.empty_like(qkv)
dk = dqkv[:, d:2 * d]
dpp, dpu, dpv = torch.empty(T, d, device=dev), torch.zeros(H, D, device=dev), torch.zeros(H, D, device=dev)
for _ in range(20):
    hip.call("oe_relpos_prepare", k, T * 3 * d, 3 * d, pp, d, pu, pv, B, T, H, D, 0.125, kp, kb)
    hip.call("oe_relpos_backward", dkp, dkb, k, T * 3 * d, 3 * d, pp, d, pu, pv, B, T, H, D, 0.125, dk, dpp, d, dpu, dpv)
torch.cuda.synchronize()
# event timing of the backward alone, warm and cold (1 GB written between calls)
flush = torch.empty(256 * 1024 * 1024, device=dev)
for label, cold in (("warm", False), ("cold", True)):
    ts = []
    for _ in range(12):
        if cold:
            flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.call("oe_relpos_backward", dkp, dkb, k, T * 3 * d, 3 * d, pp, d, pu, pv, B, T, H, D, 0.125, dk, dpp, d, dpu, dpv)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"oe_relpos_backward {label}: median {sorted(ts)[len(ts) // 2]:.1f} us (OE_RELPOS_BWD_ROWS={os.environ.get('OE_RELPOS_BWD_ROWS', '1')})")
