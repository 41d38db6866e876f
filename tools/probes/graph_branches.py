#!/usr/bin/env python3
"""How a captured HIP graph executes PARALLEL chains of short dependent kernels (the two decoders + the CTC head of the step):
S chains of M kernels forked from the capturing stream and joined again, replay time against S.  (GPU box.)"""
import sys
import time

import torch

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def run(S, n_elem, M):
    xs = [torch.zeros(n_elem, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S - 1)]
    g = torch.cuda.CUDAGraph()

    def body():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        for s, st in enumerate(streams):
            st.wait_event(ev)
            with torch.cuda.stream(st):
                for _ in range(M):
                    xs[s + 1].add_(1.0)
        for _ in range(M):
            xs[0].add_(1.0)
        for st in streams:
            e = torch.cuda.Event()
            e.record(st)
            main.wait_event(e)

    cs = torch.cuda.Stream()
    with torch.cuda.stream(cs):
        body()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=cs):
            body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    R = 20
    for _ in range(R):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / R * 1e6


for n_elem in (1024, 1 << 20, 1 << 22):
    for S in (1, 2, 3, 4):
        us = run(S, n_elem, M)
        print(f"elements {n_elem:8d}  chains {S}  x {M} kernels: replay {us:8.1f} us = {us / M:6.2f} us per kernel of a chain")
