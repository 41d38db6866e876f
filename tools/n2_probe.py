#!/usr/bin/env python3
"""Two ranks on ONE GPU over gloo (no second GPU / RCCL on the build's box): where does a data-parallel step at the bench
shape spend its time?  Times, per rank 0: the eager step (all-reduces started from backward hooks), the captured step as
one graph followed by the whole-arena all-reduce (OE_SEGMENTED=0), the captured step as a chain of graphs with the
all-reduces between them (default), and the pieces of the single-graph form (replay / all-reduce / optimizer).
gloo moves the 125 MB gradient arena through host memory: the collective's absolute cost here says nothing about RCCL
over xGMI; what carries over is which phases overlap.

    OE_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/n2_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openeat_amd import ddp, hip  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

hip.GEMM_PRECISION = 3
rank, local, world = ddp.init_from_env(backend="gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.manual_seed(777)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
eng = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
T = 998
batch = dict(features=torch.randn(32, T, 80, device=dev), features_length=torch.full((32,), T, dtype=torch.int32, device=dev),
             targets=torch.randint(2, 3000, (32, 30), dtype=torch.int32, device=dev), targets_length=torch.full((32,), 30, dtype=torch.int32, device=dev))


def timed(f, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); torch.distributed.barrier(); t0 = time.perf_counter()
        f()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[n // 2]


say = lambda *a: print(*a, flush=True) if rank == 0 else None
eng.step(batch); torch.cuda.synchronize()
say(f"[n2] eager step (hooks, overlap)            {timed(lambda: eng.step(batch)):9.1f} ms")
eng.capture(batch, warmup=1)
say(f"[n2] captured, segmented={eng.segmented}: {len(eng._segments or [1])} graph(s)   {timed(lambda: eng.replay()):9.1f} ms/step")
if eng._segments:
    def pieces():
        for g, _ in eng._segments:
            g.replay()
    say(f"[n2]    the graphs alone                      {timed(pieces):9.1f} ms")
else:
    say(f"[n2]    graph alone                           {timed(lambda: eng._graph.replay()):9.1f} ms")
say(f"[n2]    whole-arena all-reduce alone (4 chunks) {timed(lambda: eng.reducer()):9.1f} ms")
say(f"[n2]    optimizer alone                         {timed(lambda: eng.optimizer.step(lr_from_device=True)):9.1f} ms")
torch.distributed.barrier()
torch.distributed.destroy_process_group()
