#!/usr/bin/env python3
"""oe_ctc_loss_fused replayed from a HIP graph (how the training step runs it): sequential form against the pipelined form
(time chunks on two internal streams), north-star and config-2 shapes.  python tools/ctc_graph_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

V = 3246
Vp = (V + 3) // 4 * 4
for name, B, T, L in [("config 2 (B=32 x 10 s)", 32, 248, 30), ("north star (B=64 x 16 s)", 64, 398, 48)]:
    torch.manual_seed(0)
    src = torch.randn(B, T, Vp, device="cuda")
    logits = torch.empty_like(src)
    hl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ys = torch.randint(1, V, (B, L), dtype=torch.int32, device="cuda")
    yl = torch.full((B,), L, dtype=torch.int32, device="cuda")
    ws = torch.empty(hip.lib().oe_ctc_workspace_floats(B, T, L), device="cuda")
    nll, tot = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")
    alg = 2 * B * T * V * 4 + 2 * 2 * B * T * (2 * L + 1) * 4
    for mode, chunks in ((0, 4), (2, 2), (2, 3), (2, 4), (2, 6)):
        hip.lib().oe_ctc_config(mode, chunks)
        call = lambda: hip.call("oe_ctc_loss_fused", logits, Vp, B, T, V, hl, ys, L, yl, 1.0 / B, None, nll, tot, logits, ws)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            call()                                  # resources made outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            call()
        n = 30
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n + 3):
            logits.copy_(src)
            if i >= 3:
                ev[i - 3][0].record()
            g.replay()
            if i >= 3:
                ev[i - 3][1].record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[n // 2]
        print(f"{name}: {'sequential' if mode == 0 else f'pipelined, {chunks} chunks':22s} {us:8.1f} us per replay = {100 * alg / (us * 1e-6) / 8e12:.1f} % of 8 TB/s; loss {float(tot):.3f}", flush=True)
hip.lib().oe_ctc_config(0, 4)
