"""Data-parallel gradient exchange over RCCL / xGMI
(/root/reference/openeat/bin/train_ddp.py:127-134,212-219 use DistributedDataParallel).

One process per GPU.  Gradients live in one flat fp32 arena, so the exchange is
a handful of large all-reduces over contiguous memory (no bucketing copies, no
per-parameter hooks).  The arena is cut into `n_chunks` slices that are issued
back to back on a side stream; RCCL pipelines them over the 7 xGMI links.  The
sum is divided by the world size inside the collective (ReduceOp.AVG), which is
DDP's gradient averaging.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("OE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local, world


class GradAllReduce:
    def __init__(self, flat_grad: torch.Tensor, n_chunks: int = 4, process_group=None):
        self.grad = flat_grad
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        step = -(-n // max(1, n_chunks))
        step = (step + 1023) // 1024 * 1024
        self.chunks = [flat_grad[i:min(n, i + step)] for i in range(0, n, step)]
        self._avg = (hasattr(dist.ReduceOp, "AVG") and flat_grad.is_cuda and dist.is_initialized()
                     and dist.get_backend(process_group) == "nccl")

    def broadcast_parameters(self, flat_params: torch.Tensor, src: int = 0):
        """DDP's construction-time parameter broadcast."""
        if self.world > 1:
            dist.broadcast(flat_params, src=src, group=self.group)

    def __call__(self):
        if self.world == 1:
            return
        works = []
        for c in self.chunks:
            if self._avg:
                works.append(dist.all_reduce(c, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
            else:
                works.append(dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if not self._avg:
            self.grad.div_(self.world)
