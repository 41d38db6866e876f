#!/usr/bin/env python3
"""S single-chain HIP graphs replayed on S streams at once (against tools/probes/graph_branches.py: the same chains as branches of ONE graph)."""
import sys
import time

import torch

dev = "cuda"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def chain_graph(x, M, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        for _ in range(3):
            x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(M):
                x.add_(1.0)
    torch.cuda.synchronize()
    return g


def run(S, n_elem, M):
    xs = [torch.zeros(n_elem, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs = [chain_graph(xs[s], M, streams[s]) for s in range(S)]
    main = torch.cuda.current_stream()

    def once():
        ev = torch.cuda.Event()
        ev.record(main)
        for s in range(S):
            streams[s].wait_event(ev)
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
            e = torch.cuda.Event()
            e.record(streams[s])
            main.wait_event(e)

    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t = time.perf_counter()
    R = 20
    for _ in range(R):
        once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / R * 1e6


for n_elem in (1024, 1 << 20, 1 << 22):
    for S in (1, 2, 3, 4):
        us = run(S, n_elem, M)
        print(f"elements {n_elem:8d}  {S} graphs on {S} streams x {M} kernels: {us:8.1f} us per round = {us / M:6.2f} us per kernel of a chain")
