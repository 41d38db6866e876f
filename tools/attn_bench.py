#!/usr/bin/env python3
"""Attention forward / backward kernels alone at the config-2 encoder shape (B=32, H=4, T=248, D=64, dropout 0.1, key mask,
key bias) and at the north-star shape (B=64, T=398): median HIP-event time per launch, for the LDS-plane kernels
(attention_bf16.hip) and the first-generation kernels (attention.hip, OE_ATTN_PLANES=0), one child process per arm
(the dispatch switch is read once per process).  OE_BENCH_DROP overrides the dropout rate 0.1.

    python tools/attn_bench.py [precision] [B T]
Run under rocprofv3 --kernel-trace for the per-kernel split."""
import math
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(prec, B, T, H=4, D=64):
    from openeat_amd import hip
    dev = "cuda"
    torch.manual_seed(0)
    q, k, v, do = (torch.randn(B, T, H, D, device=dev) for _ in range(4))
    out, dq, dk, dv = (torch.empty_like(q) for _ in range(4))
    lse, delta = torch.empty(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
    mask = torch.ones(B, 1, T, dtype=torch.uint8, device=dev)
    mask[:, :, T - 18:] = 0
    kbias, dkb = torch.randn(B, H, T, device=dev), torch.empty(B, H, T, device=dev)
    st = (T * H * D, H * D)
    kw = dict(q_strides=st, k_strides=st, v_strides=st, o_strides=st, mask=mask, mask_strides=(T, 0), keybias=kbias, drop_p=float(os.environ.get("OE_BENCH_DROP", "0.1")), seed=1,
              precision=prec)
    af = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), **kw)
    ab = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), d_out=do, dq=dq, dk=dk, dv=dv, dkeybias=dkb, delta=delta, **kw)

    def med(f, n=30):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for _ in range(3):
            f()
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)[n // 2] * 1e3

    fw, bw = med(lambda: hip.attention_fwd(af)), med(lambda: hip.attention_bwd(ab))
    fl = 4.0 * B * H * T * T * D
    tag = "planes" if os.environ.get("OE_ATTN_PLANES", "1") != "0" else "gen-1 "
    print(f"{tag} precision {prec} B={B} T={T}: forward {fw:7.1f} us ({fl / fw / 1e6:6.1f} TF/s)   backward (dQ + dK/dV) {bw:7.1f} us "
          f"({3.5 * fl / bw / 1e6:6.1f} TF/s)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
        shapes = [(int(sys.argv[2]), int(sys.argv[3]))] if len(sys.argv) > 3 else [(32, 248), (64, 398)]
        for B, T in shapes:
            for planes in ("1", "0"):                 # the dispatch switch is read once per process: one child per arm
                env = dict(os.environ, OE_ATTN_PLANES=planes)
                subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", str(prec), str(B), str(T)], env=env)
