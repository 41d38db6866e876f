#!/usr/bin/env python3
"""The conv2 implicit GEMMs of Conv2dSubsampling4 at config 2 (B=32, 498 x 39 x 256 -> 248 x 19 x 256), forward and the four
parity classes of the input gradient, at the block tile OE_GEMM_TILE selects (default dispatch when unset).  (GPU box.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip, ops  # noqa: E402

hip.GEMM_PRECISION = 3
dev = "cuda"
B, Ti, Fi, C = 32, 498, 39, 256
To, Fo = (Ti - 3) // 2 + 1, (Fi - 3) // 2 + 1
torch.manual_seed(0)
y1 = torch.randn(B, Ti, Fi, C, device=dev).relu_()
wg = torch.randn(C, 9 * C, device=dev) / 48
bias = torch.randn(C, device=dev)
yo = torch.empty(B * To * Fo, C, device=dev)


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


fwd = lambda: hip.gemm(y1, wg, yo, B * To * Fo, C, 9 * C, lda=0, ldb=9 * C, ldc=C, bias=bias, act=1, conv=(Ti, Fi, To, Fo, C, 3, 2),
                       conv_gather=hip.GATHER_A)
us = t(fwd)
print(f"tile {os.environ.get('OE_GEMM_TILE', 'default')}: conv2 forward {us:7.1f} us  {2.0 * B * To * Fo * C * 9 * C / us / 1e6:6.1f} TFLOP/s")
dy = torch.randn(B * To * Fo, C, device=dev)
wk = torch.randn(C, C, 3, 3, device=dev) / 48
us = t(lambda: ops._conv_dgrad_k3s2(dy, wk, y1, B, Ti, Fi, To, Fo, C))
print(f"tile {os.environ.get('OE_GEMM_TILE', 'default')}: conv2 input gradient (pad + weights + 4 GEMMs) {us:7.1f} us")
