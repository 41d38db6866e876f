"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (flat-arena all-reduce with
averaging + construction-time parameter broadcast), as DistributedDataParallel does in the
reference (train_ddp.py:212-219)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from openeat_amd import ddp
    r, l, w = ddp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    grad = torch.randn(5000)
    params = torch.randn(5000)
    mine = grad.clone()
    red = ddp.GradAllReduce(grad, n_chunks=3)
    red.broadcast_parameters(params, src=0)
    red()
    gathered = [torch.zeros(5000) for _ in range(world)]
    dist.all_gather(gathered, mine)
    torch.testing.assert_close(grad, torch.stack(gathered).mean(0))
    p0 = [torch.zeros(5000) for _ in range(world)]
    dist.all_gather(p0, params)
    assert torch.equal(p0[0], p0[1])
    assert len(red.chunks) == 3 and sum(c.numel() for c in red.chunks) == 5000
    dist.destroy_process_group()
    out.put(rank)


def test_flat_gradient_allreduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


class _Toy(torch.nn.Module):
    """loss = mean((w * x)^2) on a flat parameter: stands in for the model in the epoch loop (CPU, no kernels)."""

    def __init__(self, flat):
        super().__init__()
        self.w = torch.nn.Parameter(flat)

    def forward(self, x):
        return ((self.w * x) ** 2).mean(), None


class _Log:
    def info(self, *_):
        pass


def _loop_worker(rank, world, port, out):
    """Executor.train with ragged loaders: rank 1 has one batch fewer AND one empty batch; accum_grad 2."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from openeat_amd import ddp
    from openeat_amd.utils.executor import Executor
    ddp.init_from_env(backend="gloo")
    torch.manual_seed(0)
    flat = torch.randn(64)
    model = _Toy(flat.clone())
    model.w.grad = torch.zeros(64)
    red = ddp.GradAllReduce(model.w.grad, n_chunks=2)
    calls = []
    orig = red.__class__.__call__
    red_call = lambda: (calls.append(1), orig(red))
    opt = torch.optim.SGD([model.w], lr=0.1)
    opt.zero_grad = lambda *a, **k: model.w.grad.zero_()      # keep the flat gradient buffer the reducer holds (as FusedAdam does)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    g = torch.Generator().manual_seed(7 + rank)
    n = 6 if rank == 0 else 5
    data = [(["u"], {"x": torch.randn(64, generator=g)}) for _ in range(n)]
    if rank == 1:
        data[2] = ([], {"x": torch.zeros(64)})               # a batch whose utterances could not be read
    class Red:                                                # count the collective calls of this rank
        world = red.world
        def agree_min(self, v): return red.agree_min(v)
        def __call__(self): red_call()
        overlap_enabled = True
    ex = Executor()
    ex.train(_Log(), model, opt, sch, data, "cpu", {"accum_grad": 2, "grad_reducer": Red(), "grad_clip": 1e9}, rank)
    ws = [torch.zeros(64) for _ in range(world)]
    dist.all_gather(ws, model.w.detach())
    assert torch.equal(ws[0], ws[1]), "ranks diverged"
    # 5 common iterations (rank 0's 6th is dropped), iteration 2 skipped everywhere: boundaries at idx 0 and 4
    assert ex.step == 2 and len(calls) == 2, (ex.step, len(calls))
    dist.destroy_process_group()
    out.put(rank)


def test_epoch_loop_with_ragged_and_empty_batches_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
