"""Data-parallel gradient exchange over RCCL / xGMI
(/root/reference/openeat/bin/train_ddp.py:127-134,212-219 use DistributedDataParallel).

One process per GPU.  Gradients live in one flat fp32 arena, so the exchange is
a handful of large all-reduces over contiguous memory (no bucketing copies, no
per-parameter hooks).  The arena is laid out in forward order, so the part whose
gradients are final at any point of backward is a contiguous tail: `reduce_tail`
(called from tensor hooks: gradient of the encoder output ready = heads done, then
the gradient of the input of the encoder layers at the quarter points of the stack)
sends that tail to RCCL while the rest of backward keeps running; `__call__` after backward
sends what is left in `n_chunks` slices and waits.  The sum is divided by the
world size inside the collective (ReduceOp.AVG), which is DDP's gradient averaging.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def forced() -> bool:
    """OE_DDP_FORCE=1: run the whole data-parallel machinery - process group, hooks, all-reduces of the arena tails between the
    segment graphs, ReduceOp.AVG - also with ONE rank.  A one-GPU box can then execute every RCCL call of the N > 1 path
    (the collectives are real RCCL kernels on the stream; with one rank they reduce over a group of one)."""
    return os.environ.get("OE_DDP_FORCE", "0") == "1"


def init_from_env(backend: str = None):
    """torchrun-style env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
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
        # collectives are issued when there is more than one rank - or when forced (one-rank rehearsal of the RCCL path)
        self.active = self.world > 1 or (forced() and dist.is_initialized())
        self.issued = 0                # all-reduce calls handed to the backend so far (tests)
        self._cpu_group = None         # host-side group of agree_min (None: not yet made; False: unavailable)
        n = flat_grad.numel()
        step = -(-n // max(1, n_chunks))
        step = (step + 1023) // 1024 * 1024
        self.n_chunks = max(1, n_chunks)
        self.chunks = [flat_grad[i:min(n, i + step)] for i in range(0, n, step)]
        self._done_from = n            # floats [_done_from, n) have already been handed to the collective this step
        self._works = []
        self.overlap_enabled = True    # False on the non-boundary micro-steps of gradient accumulation (DDP's no_sync)
        self._avg = (hasattr(dist.ReduceOp, "AVG") and flat_grad.is_cuda and dist.is_initialized()
                     and dist.get_backend(process_group) == "nccl")

    def broadcast_parameters(self, flat_params: torch.Tensor, src: int = 0):
        """DDP's construction-time parameter broadcast."""
        if self.active:
            dist.broadcast(flat_params, src=src, group=self.group)

    def _issue(self, t: torch.Tensor):
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if not self._avg and t.is_cuda:
            # a host-staged backend (gloo rehearsals on one GPU): its device-to-host copy behind a HIP-graph launch that has
            # not finished took SECONDS per call on this stack (tools/n2_probe.py: graphs alone 35 ms, arena all-reduce
            # alone 32 ms, back to back 6-12 s); behind a finished stream it takes its 32 ms.  RCCL's collectives are
            # kernels ordered by stream events and never come here.
            torch.cuda.current_stream().synchronize()
        self._works.append(dist.all_reduce(t, op=op, group=self.group, async_op=True))
        self.issued += 1

    def reduce_tail(self, start: int):
        """Gradients of floats [start, end of what is still pending) are final: start their all-reduce now (async; the
        collective is ordered after everything already enqueued on the current stream)."""
        if not self.active or not self.overlap_enabled or start >= self._done_from:
            return
        if start > 0:
            start = (start + 1023) // 1024 * 1024      # 4 KiB aligned slices; the few floats skipped go with the next tail
            if start >= self._done_from:
                return
        self._issue(self.grad[start:self._done_from])
        self._done_from = start

    def agree_min(self, value: int) -> int:
        """MIN over the ranks of a small integer (the loop's "do we all have a batch" handshake).  Runs on a host-side gloo
        group (created on first use - a collective call itself, which every rank makes at its first handshake), so the
        per-batch handshake never synchronises the host with the GPU stream the way an RCCL all-reduce + .item() would."""
        if not self.active:
            return value
        if self._cpu_group is None:
            try:
                self._cpu_group = self.group if dist.get_backend(self.group) == "gloo" else dist.new_group(backend="gloo")
            except Exception:                    # noqa: BLE001 - no gloo in this build: fall back to the device group
                self._cpu_group = False
        if self._cpu_group is False:
            t = torch.tensor([value], dtype=torch.int32, device=self.grad.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            return int(t.item())
        t = torch.tensor([value], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._cpu_group)
        return int(t.item())

    def __call__(self):
        """After backward: reduce what no hook has sent yet, wait for everything, average."""
        if not self.active:
            return
        n = self._done_from
        if n > 0:
            step = -(-n // self.n_chunks)
            step = (step + 1023) // 1024 * 1024
            for i in range(0, n, step):
                self._issue(self.grad[i:min(n, i + step)])
        for w in self._works:
            w.wait()
        self._works = []
        self._done_from = self.grad.numel()
        if not self._avg:
            self.grad.div_(self.world)
