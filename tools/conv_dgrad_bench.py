#!/usr/bin/env python3
"""conv2 input gradient at config-2 size (B=32, 498 x 39 -> 248 x 19, C=256): column buffer + col2im vs the four
parity-class implicit GEMMs (ops._conv_dgrad_k3s2); per-launch HIP-event times of the latter."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, ops  # noqa: E402

hip.GEMM_PRECISION = 3
B, Ti, Fi, C = 32, 498, 39, 256
To, Fo = (Ti - 3) // 2 + 1, (Fi - 3) // 2 + 1
dev = "cuda"
dy = torch.randn(B * To * Fo, C, device=dev)
wk = torch.randn(C, C, 3, 3, device=dev) * 0.05
yin = torch.randn(B, Ti, Fi, C, device=dev)
wg = torch.empty(C, 9 * C, device=dev)
hip.call("oe_swap_last2", wk, C, C, 9, wg, 0)


def old():
    dcol = ops.gemm_nn(dy, wg)
    out = torch.empty_like(yin)
    hip.call("oe_col2im_relu_ks", dcol, yin, B, Ti, Fi, C, 3, 2, out)
    return out


def new():
    return ops._conv_dgrad_k3s2(dy, wk, yin, B, Ti, Fi, To, Fo, C)


def t(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a, b = old(), new()
torch.cuda.synchronize()
print("max abs difference old vs new:", float((a - b).abs().max()), "of", float(a.abs().max()))
print(f"column buffer + col2im: {t(old):.0f} us    implicit (pad + 4 weight selections + 4 GEMMs): {t(new):.0f} us")
hip.PROFILE = []
new()
torch.cuda.synchronize()
for e0, e1, fl, key in hip.PROFILE:
    us = e0.elapsed_time(e1) * 1e3
    print(f"  class GEMM m={key[0]} n={key[1]} k={key[2]}: {us:.0f} us  {fl / us / 1e6:.0f} TFLOP/s")
hip.PROFILE = None
