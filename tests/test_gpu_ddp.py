"""GPU, two ranks (both on cuda:0, gloo so that no second GPU / RCCL is needed): one TrainEngine step per rank on
different minibatches must leave both ranks with identical parameters, equal to what a single process gets by
averaging the two ranks' gradients before the same clip + Adam update (DistributedDataParallel semantics,
train_ddp.py:212-219), including the all-reduce phases that start inside backward."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONF = dict(encoder_num_blocks=4, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4, linear_units=64,
            dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=15,
            pos_enc_layer_type="rel_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3)
V = 40


def _batch(seed, dev):
    g = torch.Generator().manual_seed(seed)
    B, T, L = 3, 83, 6
    feats = torch.randn(B, T, 80, generator=g)
    flen = torch.tensor([83, 64, 41], dtype=torch.int32)
    tgt = torch.randint(2, V - 1, (B, L), generator=g, dtype=torch.int32)
    tlen = torch.tensor([6, 4, 3], dtype=torch.int32)
    for b in range(B):
        tgt[b, int(tlen[b]):] = -1
    return dict(features=feats.to(dev), features_length=flen.to(dev), targets=tgt.to(dev), targets_length=tlen.to(dev))


def _model(dev):
    from openeat_amd.models.asr_model import ASRModel
    torch.manual_seed(7)
    return ASRModel(80, V, **CONF).to(dev).train()


def worker():
    sys.path.insert(0, ROOT)
    from openeat_amd import ddp
    from openeat_amd.engine import TrainEngine
    rank, _, world = ddp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    model = _model(dev)
    # as bench.py runs it at N > 1: right decoder and CTC head on their own streams, collectives started from backward hooks
    eng = TrainEngine(model, lr=1e-2, grad_clip=5.0, parallel_decoders=True)
    assert eng.reducer.world == 2 and hasattr(model, "grad_ready_hooks") and hasattr(model.encoder, "grad_ready_hooks")
    issued = []
    orig = eng.reducer.reduce_tail
    eng.reducer.reduce_tail = lambda start: (issued.append(start), orig(start))[1]
    loss, _ = eng.step(_batch(100 + rank, dev))
    torch.cuda.synchronize()
    # heads first, then the encoder's quarter points from the top (4 layers: layers 3, 2, 1)
    assert len(issued) == 4 and all(a > b for a, b in zip(issued, issued[1:])) and issued[-1] > 0, issued
    out = {"flat": eng.arena.flat.detach().cpu(), "loss": float(loss), "issued": issued}
    torch.save(out, os.environ["OE_TEST_OUT"] + f".{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_matches_gradient_averaging(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ddp_out")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2",
                   OE_TEST_OUT=out, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-4000:]
    got = [torch.load(out + f".{r}") for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])                       # both ranks hold the same parameters

    # single process: average the two ranks' gradients, then the same clip + Adam
    from openeat_amd.engine import TrainEngine
    dev = torch.device("cuda:0")
    model = _model(dev)
    eng = TrainEngine(model, lr=1e-2, grad_clip=5.0)
    grads = []
    for r in range(2):
        eng.arena.zero_grad()
        eng._fwd_bwd(_batch(100 + r, dev))
        grads.append(eng.arena.grad.clone())
    eng.arena.grad.copy_((grads[0] + grads[1]) / 2)
    eng.optimizer.step()
    torch.cuda.synchronize()
    ref = eng.arena.flat.detach().cpu()
    # Adam's first step moves every parameter by ~lr * sign(g): compare the updates, tolerance a fraction of lr
    torch.manual_seed(7)
    init = _model(dev)
    from openeat_amd.arena import ParamArena
    eng.arena.deactivate()
    init_flat = ParamArena(init).flat.detach().cpu()
    du_ref, du_got = ref - init_flat, got[0]["flat"] - init_flat
    assert float(du_ref.abs().max()) > 5e-3                                   # the step did move the parameters
    bad = (du_ref - du_got).abs() > 2e-3 * 1e-2 + 0.02 * du_ref.abs()
    # parameters whose averaged gradient is ~0 get a sign-of-noise update from Adam: allow a small fraction of those
    assert float(bad.float().mean()) < 2e-3, float(bad.float().mean())


def worker_segmented():
    """Both ranks: three eager steps (hooks start the all-reduces inside backward) on one copy of the model, and
    warm-up + two replays of the segmented capture (all-reduces between the graphs) on another; same batch every step."""
    sys.path.insert(0, ROOT)
    from openeat_amd import ddp, ops
    from openeat_amd.engine import TrainEngine
    rank, _, world = ddp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    batch = _batch(200 + rank, dev)
    out = {}
    for tag in ("eager", "segmented"):
        model = _model(dev)
        eng = TrainEngine(model, lr=1e-2, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
        assert eng.segmented and eng.reducer.world == 2
        issued = []
        orig = eng.reducer.reduce_tail
        eng.reducer.reduce_tail = lambda start, orig=orig, issued=issued: (issued.append(start), orig(start))[1]
        if tag == "eager":
            for _ in range(3):
                loss, _ = eng.step(batch)
        else:
            eng.capture(batch, warmup=1)                     # one real step, then the capture
            assert len(eng._segments) == 5
            n0 = len(issued)
            for _ in range(2):
                loss, _ = eng.replay()
            assert len(issued) - n0 == 8, issued             # four tails per replay, started between the graphs
        torch.cuda.synchronize()
        out[tag] = (eng.arena.flat.detach().cpu(), float(loss))
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
    torch.save(out, os.environ["OE_TEST_OUT"] + f".{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_segmented_capture_matches_eager_steps(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "seg_out")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2",
                   OE_TEST_OUT=out, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker_segmented"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-4000:]
    got = [torch.load(out + f".{r}") for r in range(2)]
    for tag in ("eager", "segmented"):
        assert torch.equal(got[0][tag][0], got[1][tag][0])                   # both ranks hold the same parameters
    a, b = got[0]["eager"][0], got[0]["segmented"][0]
    lr, steps = 1e-2, 3
    diff = (a - b).abs()
    assert float(diff.max()) <= 2.05 * lr * steps
    assert float((diff > 0.1 * lr * steps + 1e-3 * a.abs()).float().mean()) < 0.02
    assert abs(got[0]["eager"][1] - got[0]["segmented"][1]) < 1e-3 * abs(got[0]["eager"][1])


def worker_rccl_one_rank():
    """ONE rank, backend nccl (= RCCL), OE_DDP_FORCE=1: every collective of the N > 1 path really runs on the GPU - the
    parameter broadcast, the arena-tail all-reduces the backward hooks start inside an eager step, ReduceOp.AVG, the
    all-reduces between the five graphs of the segmented capture (RCCL kernels beside graph replays, the capture in
    thread-local mode next to RCCL's watchdog thread), clip + Adam behind the last collective.  A reduction over a group of
    one changes nothing, so both paths must equal a plain single-process engine step for step."""
    sys.path.insert(0, ROOT)
    from openeat_amd import ddp, ops
    from openeat_amd.engine import TrainEngine
    assert ddp.forced()
    rank, _, world = ddp.init_from_env(backend="nccl")
    assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
    dev = torch.device("cuda:0")
    batch = _batch(300, dev)
    out = {}
    for tag in ("eager", "segmented"):
        model = _model(dev)
        eng = TrainEngine(model, lr=1e-2, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
        assert eng.reducer.active and eng.reducer._avg and eng.segmented and hasattr(model, "grad_ready_hooks")
        n0 = eng.reducer.issued
        if tag == "eager":
            for _ in range(3):
                loss, _ = eng.step(batch)
            assert eng.reducer.issued - n0 >= 3 * 5              # per step: four tails from the hooks + what is left after backward
        else:
            eng.capture(batch, warmup=1)
            assert len(eng._segments) == 5
            n1 = eng.reducer.issued
            for _ in range(2):
                loss, _ = eng.replay()
            assert eng.reducer.issued - n1 >= 2 * 5              # four tails between the graphs + the rest, per replay
        torch.cuda.synchronize()
        out[tag] = (eng.arena.flat.detach().cpu(), float(loss))
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
    torch.save(out, os.environ["OE_TEST_OUT"])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_rccl_collectives_run_with_one_rank(tmp_path):
    """The RCCL branch of the data-parallel path executed on one GPU (see worker_rccl_one_rank) and compared with an engine
    that has no process group at all: three eager steps / one warm-up step + two segmented replays on the same batch."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rccl1_out")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", OE_DDP_FORCE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", OE_TEST_OUT=out, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker_rccl_one_rank"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    log = p.communicate(timeout=600)[0]
    assert p.returncode == 0, log[-4000:]
    got = torch.load(out)
    from openeat_amd import ops
    from openeat_amd.engine import TrainEngine
    dev = torch.device("cuda:0")
    model = _model(dev)
    eng = TrainEngine(model, lr=1e-2, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
    assert not eng.reducer.active
    try:
        for _ in range(3):
            loss, _ = eng.step(_batch(300, dev))
        torch.cuda.synchronize()
        ref = eng.arena.flat.detach().cpu()
    finally:
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False
    lr, steps = 1e-2, 3
    for tag in ("eager", "segmented"):
        diff = (got[tag][0] - ref).abs()
        assert float(diff.max()) <= 2.05 * lr * steps, tag
        assert float((diff > 0.1 * lr * steps + 1e-3 * ref.abs()).float().mean()) < 0.02, tag
        assert abs(got[tag][1] - float(loss)) < 1e-3 * abs(float(loss)), tag


def _batch_shape(seed, dev, T, L):
    g = torch.Generator().manual_seed(seed)
    B = 3
    feats = torch.randn(B, T, 80, generator=g)
    flen = torch.tensor([T, T - 11, T - 30], dtype=torch.int32)
    tgt = torch.randint(2, V - 1, (B, L), generator=g, dtype=torch.int32)
    tlen = torch.tensor([L, L - 1, L - 2], dtype=torch.int32)
    for b in range(B):
        tgt[b, int(tlen[b]):] = -1
    return dict(features=feats.to(dev), features_length=flen.to(dev), targets=tgt.to(dev), targets_length=tlen.to(dev))


def worker_cached_ragged():
    """step_cached with DIFFERENT shape orders per rank (ragged buckets, dataset.py:337-364): at some calls one rank holds a
    graph for its batch and the other sees its shape for the first time.  The ranks must agree the launch mode per call
    (round 3's step_cached let the missing rank run a capture-time handshake the other rank never answered) - with and
    without gradient accumulation, where the replayed and the eager boundary step issue different collectives."""
    sys.path.insert(0, ROOT)
    from openeat_amd import ddp, ops
    from openeat_amd.engine import TrainEngine
    rank, _, world = ddp.init_from_env(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    shapes = {"a": (83, 6), "b": (67, 5)}
    order = ["a", "b", "a", "a", "b", "a", "b", "b"] if rank == 0 else ["b", "b", "a", "b", "a", "a", "b", "a"]
    out = {}
    for accum in (1, 2):
        model = _model(dev)
        eng = TrainEngine(model, lr=1e-2, grad_clip=5.0, accum_grad=accum, static_shapes=True, parallel_decoders=True)
        modes = []
        for i, name in enumerate(order):
            h0 = eng.cache_hits
            loss, _ = eng.step_cached(_batch_shape(1000 * accum + 10 * i + rank, dev, *shapes[name]))
            modes.append(eng.cache_hits - h0)
            assert bool(torch.isfinite(loss))
        torch.cuda.synchronize()
        out[accum] = (eng.arena.flat.detach().cpu(), modes, eng.cache_misses)
        eng.drop_graph()
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
    torch.save(out, os.environ["OE_TEST_OUT"] + f".{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_step_cached_with_different_shape_orders(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "ragged_out")
    procs = []
    for r in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2",
                   OE_TEST_OUT=out, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker_cached_ragged"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:                    # ranks that disagree about a collective hang: kill exactly these two
            for q in procs:
                q.kill()
            raise AssertionError("the ranks hung (mismatched collectives)")
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-4000:]
    got = [torch.load(out + f".{r}") for r in range(2)]
    for accum in (1, 2):
        assert torch.equal(got[0][accum][0], got[1][accum][0]), accum             # the ranks never diverged
        assert got[0][accum][1] == got[1][accum][1], (accum, got[0][accum][1], got[1][accum][1])   # same launch mode at every call
        assert sum(got[0][accum][1]) >= 2                                         # and graphs were replayed once both ranks held them
        assert got[0][accum][2] == 2 and got[1][accum][2] == 2                    # each shape captured once per rank


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worker_rccl_one_rank":
    worker_rccl_one_rank()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worker_segmented":
    worker_segmented()
if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "worker_cached_ragged":
    worker_cached_ragged()
