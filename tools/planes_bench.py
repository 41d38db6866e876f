#!/usr/bin/env python3
"""Plane GEMM (oe_gemm_planes on bf16 hi/lo planes) vs oe_gemm_f32 on the shapes of the Conformer step (GPU box).
Reports: max error vs float64, time of the split passes, of the GEMM, and of the current fp32-operand GEMM."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
M = 7936
SHAPES = [("ffn.w1 fwd", (M, 1024, 256)), ("ffn.w2 fwd", (M, 256, 1024)), ("qkv fwd", (M, 768, 256)), ("attn.out fwd", (M, 256, 256)),
          ("ffn.w1 wgrad (as NT on ^T planes)", (1024, 256, M)), ("out wgrad", (256, 256, M)), ("dec 992", (992, 256, 1024))]
L = hip.lib()
dev = "cuda"


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def split(x, transpose=False):
    r, c = x.shape
    shp = (c, r) if transpose else (r, c)
    hi = torch.empty(shp, dtype=torch.bfloat16, device=dev)
    lo = torch.empty(shp, dtype=torch.bfloat16, device=dev)
    f = lambda: hip.check(L.oe_split_bf16(hip.ptr(x), x.stride(0), r, c, int(transpose), hip.ptr(hi), hip.ptr(lo), shp[1], hip.stream()), "split")
    f()
    return hi, lo, f


print(f"precision {prec}  OE_PLANES_NST={os.environ.get('OE_PLANES_NST', '4')} OE_GEMM_TILE={os.environ.get('OE_GEMM_TILE', '-')}")
print(f"{'shape':36s} {'m':>6s} {'n':>6s} {'k':>6s} {'err/sqrtK':>10s} {'split A us':>10s} {'planes us':>10s} {'TF/s':>7s} {'fp32-op us':>10s}")
for name, (m, n, k) in SHAPES:
    torch.manual_seed(0)
    a, b = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev)
    bias = torch.randn(n, device=dev)
    ah, al, fa = split(a)
    bh, bl, _ = split(b)
    c = torch.zeros(m, n, device=dev)
    sk = 1
    atomic = False
    if k >= 4096:                                   # weight-gradient shape: split the long reduction
        from openeat_amd.ops import _split_k
        sk, atomic = _split_k(m, n, k), True
    g = hip.GemmArgs()
    g.a = g.b = None
    g.lda, g.ldb, g.c, g.ldc = k, k, c.data_ptr(), n
    g.m, g.n, g.k, g.split_k = m, n, k, sk
    g.alpha, g.beta, g.precision, g.atomic_out = 1.0, 1.0, prec, int(atomic)
    g.bias = None if atomic else bias.data_ptr()
    run = lambda: hip.check(L.oe_gemm_planes(C.byref(g), hip.ptr(ah), hip.ptr(al), hip.ptr(bh), hip.ptr(bl), hip.stream()), "planes")
    c.zero_()
    run()
    torch.cuda.synchronize()
    ref = a.double() @ b.double().T + (0 if atomic else bias.double())
    err = float((c.double() - ref).abs().max()) / math.sqrt(k)
    t_split = timeit(fa)
    t_pl = timeit(run)
    c2 = torch.zeros(m, n, device=dev)
    t_old = timeit(lambda: hip.gemm(a, b, c2, m, n, k, lda=k, ldb=k, ldc=n, bias=None if atomic else bias, split_k=sk, atomic_out=atomic, precision=prec))
    print(f"{name:36s} {m:6d} {n:6d} {k:6d} {err:10.2e} {t_split:10.1f} {t_pl:10.1f} {2.0 * m * n * k / t_pl / 1e6:7.1f} {t_old:10.1f}", flush=True)
