#!/usr/bin/env python3
"""Floor measurements on the GPU box: back-to-back launch cost and plain streaming kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd import hip  # noqa: E402

dev = "cuda"


def timeit(f, reps=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (256, 2 * 1024 * 1024, 8 * 1024 * 1024):
    x = torch.randn(n, device=dev)
    y = torch.empty(n, device=dev)
    print(f"axpby n={n:9d}: {timeit(lambda: hip.call('oe_axpby', x, None, n, 2.0, 0.0, None, y)):7.2f} us   torch mul: {timeit(lambda: torch.mul(x, 2.0, out=y)):7.2f} us", flush=True)
# graph replay of 20 tiny launches
g = torch.cuda.CUDAGraph()
x = torch.randn(256, device=dev)
y = torch.empty(256, device=dev)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        hip.call("oe_axpby", x, None, 256, 2.0, 0.0, None, y)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    for _ in range(20):
        hip.call("oe_axpby", x, None, 256, 2.0, 0.0, None, y)
print(f"graph of 20 tiny launches: {timeit(lambda: g.replay(), 20) / 20:7.2f} us per kernel", flush=True)
M, N, K = 7936, 1024, 32
a, b, c = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
print(f"gemm 7936x1024x32 p1 eager: {timeit(lambda: hip.gemm(a, b, c, M, N, K, lda=K, ldb=K, ldc=N, precision=1)):7.2f} us")
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    for _ in range(20):
        hip.gemm(a, b, c, M, N, K, lda=K, ldb=K, ldc=N, precision=1)
print(f"gemm 7936x1024x32 p1 in graph: {timeit(lambda: g2.replay(), 20) / 20:7.2f} us per kernel", flush=True)
