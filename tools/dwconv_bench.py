#!/usr/bin/env python3
"""GLU + depthwise conv forward / backward alone at config 2 (B=32, T'=248, d=256, K=15): median HIP-event time. (GPU box.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip
B, T, d, K = 32, 248, 256, 15
a, dy = torch.randn(B, T, 2 * d, device="cuda"), torch.randn(B, T, d, device="cuda")
w, bias = torch.randn(d, K, device="cuda"), torch.randn(d, device="cuda")
y, da = torch.empty(B, T, d, device="cuda"), torch.empty_like(a)
dw, db = torch.zeros(d, K, device="cuda"), torch.zeros(d, device="cuda")
ws = torch.empty(hip.lib().oe_dwconv_glu_bwd_workspace_floats(B, T, d, K), device="cuda")
def med(f, n=40):
    for _ in range(3): f()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in ev)[n // 2] * 1e3
print(f"dwconv_glu fwd {med(lambda: hip.call('oe_dwconv_glu_fwd', a, w, bias, None, B, T, d, K, 0, y)):.1f} us   "
      f"bwd (+ parameter reduce) {med(lambda: hip.call('oe_dwconv_glu_bwd', a, dy, w, None, B, T, d, K, 0, da, dw, db, None, ws)):.1f} us")
