#!/usr/bin/env python3
"""Where does a workgroup of the LDS-DMA GEMM spend its cycles?  (GPU box, diagnostic library.)

Build:  OE_DIAG=1 bash openeat_amd/csrc/build.sh
Run:    OE_HIP_LIB=openeat_amd/lib/libopeneat_hip_diag.so python tools/gemm_stamps.py [precision] [tile]
Stamps (s_memtime, shader cycles) per workgroup: 0 entry, 1 ring primed (DMA issued), 2 first tile landed,
3 K-loop done, 4 epilogue done.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = 7936
SHAPES = [("ffn.w1 fwd", "nt", (M, 1024, 256)), ("ffn.w2 fwd", "nt", (M, 256, 1024)), ("attn.out fwd", "nt", (M, 256, 256)),
          ("ffn.w1 dgrad", "nn", (M, 256, 1024)), ("ffn.w1 wgrad", "tn", (1024, 256, M))]
lib = C.CDLL(os.environ["OE_HIP_LIB"])
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
assert lib.oe_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
for name, kind, (m, n, k) in SHAPES:
    dev = "cuda"
    if kind == "nt":
        a, b, c = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=k, ldb=k, ldc=n, precision=prec)
    elif kind == "nn":
        a, b, c = torch.randn(m, k, device=dev), torch.randn(k, n, device=dev), torch.empty(m, n, device=dev)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=k, ldb=n, ldc=n, b_kmajor=True, precision=prec)
    else:
        from openeat_amd.ops import _split_k
        a, b, c = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev), torch.zeros(m, n, device=dev)
        sk = _split_k(m, n, k)
        f = lambda: hip.gemm(a, b, c, m, n, k, lda=m, ldb=n, ldc=n, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True, precision=prec)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf.zero_()
    f()
    torch.cuda.synchronize()
    s = buf.view(4096, 8).cpu()
    live = s[:, 4] > 0
    s = s[live].double()
    nb = int(live.sum())
    if nb == 0:
        print(f"{name:14s} not on the LDS-DMA kernel (no stamps)")
        continue
    t0 = s[:, 0].min()
    seg = [(s[:, i + 1] - s[:, i]).median().item() for i in range(4)]
    e = [(s[:, b] - s[:, a]).median().item() for a, b in ((3, 5), (5, 6), (6, 7), (7, 4))]
    print(f"{name:14s} epilogue split: barrier {e[0]:6.0f}  tile0 loads+patch {e[1]:6.0f}  tile0 math+stores {e[2]:6.0f}  remaining tiles {e[3]:6.0f}")
    print(f"{name:14s} blocks {nb:5d}  span {int(s[:, 4].max() - t0):7d} cyc | median per block: prime {seg[0]:6.0f}  first-tile {seg[1]:6.0f}  "
          f"k-loop {seg[2]:7.0f}  epilogue {seg[3]:6.0f}  total {(s[:, 4] - s[:, 0]).median().item():7.0f} | "
          f"start spread p50/p99 {(s[:, 0] - t0).median().item():7.0f}/{(s[:, 0] - t0).quantile(0.99).item():7.0f}", flush=True)
