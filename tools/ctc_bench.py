#!/usr/bin/env python3
"""CTC head (oe_ctc_loss_fused) on its own at the config-2 and north-star shapes (GPU box).

  python tools/ctc_bench.py                 # wall time per call (HIP events) and GB/s against the algorithmic bytes
  rocprofv3 --kernel-trace --stats -d gpurun_out/ctc_prof -- python3 tools/ctc_bench.py     # per-kernel split

Algorithmic bytes: 2*B*T'*V*4 (logits read once, gradient written once) + 2*2*B*T'*(2L+1)*4 (alpha, beta).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

V = 3246
Vp = (V + 3) // 4 * 4
PEAK = 8.0e12
FORMS = {0: "three launches", 3: "overlapped where profitable", 4: "overlapped", 1: "chunk pipeline", 2: "chunk pipeline (forced)"}
for name, B, T, L in [("config 2 (B=32 x 10 s)", 32, 248, 30), ("north star (B=64 x 16 s)", 64, 398, 48)]:
    torch.manual_seed(0)
    src = torch.randn(B, T, Vp, device="cuda")
    logits = torch.empty_like(src)
    hl = torch.full((B,), T, dtype=torch.int32, device="cuda")
    ys = torch.randint(1, V, (B, L), dtype=torch.int32, device="cuda")
    yl = torch.full((B,), L, dtype=torch.int32, device="cuda")
    ws = torch.empty(hip.lib().oe_ctc_workspace_floats(B, T, L), device="cuda")
    nll, tot = torch.empty(B, device="cuda"), torch.empty(1, device="cuda")

    def call():
        hip.call("oe_ctc_loss_fused", logits, Vp, B, T, V, hl, ys, L, yl, 1.0 / B, None, nll, tot, logits, ws)
    for form in (int(x) for x in os.environ.get("CTC_BENCH_FORMS", "0,4").split(",")):
        hip.lib().oe_ctc_config(form, 4)
        n = 20
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i in range(n + 3):
            logits.copy_(src)               # the gradient overwrites the logits: fresh logits each call (not timed)
            if i >= 3:
                ev[i - 3][0].record()
            call()
            if i >= 3:
                ev[i - 3][1].record()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[n // 2]
        alg = 2 * B * T * V * 4 + 2 * 2 * B * T * (2 * L + 1) * 4
        print(f"{name} [{FORMS.get(form, form)}]: {us:8.1f} us per call, algorithmic {alg / 1e6:.0f} MB -> {alg / us / 1e6:.2f} TB/s = "
              f"{100 * alg / (us * 1e-6) / PEAK:.1f} % of 8 TB/s; loss {float(tot):.3f}")
hip.lib().oe_ctc_config(0, 4)
