#!/usr/bin/env python3
"""Do the two decoders' latency chains overlap when they are TWO single-chain HIP graphs on two streams, instead of two branches of
one graph?  Forward only (no_grad), config-2 shapes (B = 32, T' = 248, L = 31).  (GPU box.)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from openeat_amd import hip, ops, planes  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

dev = torch.device("cuda:0")
hip.GEMM_PRECISION = 6
torch.manual_seed(0)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).eval()
B, T, L = 32, 248, 31
enc = torch.randn(B, T, 256, device=dev)
mask = torch.ones(B, 1, T, dtype=torch.bool, device=dev)
ys = torch.randint(1, bench.V - 1, (B, L), device=dev)
tgt_mask = torch.ones(B, L, L, dtype=torch.bool, device=dev).tril()
dec = model.decoder


def left():
    return dec.left_decoder.hidden(ys, tgt_mask, enc, mask)


def right():
    return dec.right_decoder.hidden(ys, tgt_mask, enc, mask)


def timeit(fn, n=30):
    """GPU-side duration: a 1.5 ms spin parks the stream while the host enqueues fn, events bracket fn on the device."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        torch.cuda._sleep(3_000_000)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / n


with torch.no_grad():
    left(); right()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    # (a) each decoder alone as a single-chain graph
    with planes.capture_scope():
        gl = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gl, stream=s1):
            left()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s2):
            right()
        # (b) one graph, both decoders as branches
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=s1):
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                right()
            left()
            s1.wait_stream(s2)
        # (c) one graph, serial
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, stream=s1):
            left(); right()
    main = torch.cuda.current_stream()

    def two_streams():
        e = torch.cuda.Event(); e.record(main)
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1):
            gl.replay()
        with torch.cuda.stream(s2):
            gr.replay()
        main.wait_stream(s1); main.wait_stream(s2)

    print(f"left decoder alone (one chain):            {timeit(gl.replay):8.1f} us")
    print(f"right decoder alone (one chain):           {timeit(gr.replay):8.1f} us")
    print(f"one graph, left then right:                {timeit(gs.replay):8.1f} us")
    print(f"one graph, two branches:                   {timeit(gb.replay):8.1f} us")
    print(f"two graphs on two streams:                 {timeit(two_streams):8.1f} us")
