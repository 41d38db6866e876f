"""GPU: front end, parameter arena, fused optimiser, training engine (eager and HIP-graph replay)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, load_golden_json  # noqa: E402
from openeat_amd import arena as A  # noqa: E402
from openeat_amd import ops  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.frontend import Fbank, utt_normalize_  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402
from oracle import asr as O  # noqa: E402
from oracle import fbank as FB  # noqa: E402

DEV = "cuda"


def test_fbank_and_utt_norm_match_oracle():
    """fp32; tolerance rtol 2e-4 / atol 2e-3 on log-mel (FFT summation order differs from pocketfft;
    quiet bins amplify it through the log).  The oracle itself is UNPINNED against torchaudio."""
    torch.manual_seed(3)
    lens = [16000, 12345, 4000, 399]
    wav = torch.zeros(4, 16000)
    for b, n in enumerate(lens):
        wav[b, :n] = (torch.rand(n) - 0.5) * (0.9 if b != 2 else 0.01)
    fb = Fbank(80, device=DEV)
    feats, nfr = fb(wav.to(DEV), torch.tensor(lens, device=DEV))
    torch.cuda.synchronize()
    assert nfr.tolist() == [98, 75, 23, 0]
    for b, n in enumerate(lens):
        ref = FB.fbank(wav[b, :n])
        got = feats[b, : ref.shape[0]].cpu()
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-4, atol=2e-3)
        assert torch.all(feats[b, ref.shape[0]:] == 0)              # padded frames are zero
    f2 = feats.clone()
    utt_normalize_(f2, nfr)
    torch.cuda.synchronize()
    for b in range(3):
        n = int(nfr[b])
        np.testing.assert_allclose(f2[b, :n].cpu().numpy(), FB.utt_normalize(feats[b, :n].cpu()).numpy(), rtol=1e-4, atol=1e-4)
    # fused global CMVN
    mean, istd = torch.randn(80), torch.rand(80) + 0.5
    f3, _ = fb(wav.to(DEV), torch.tensor(lens, device=DEV), cmvn=(mean.to(DEV), istd.to(DEV)))
    torch.cuda.synchronize()
    np.testing.assert_allclose(f3[0].cpu().numpy(), ((feats[0].cpu() - mean) * istd).numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rate,frame_ms,n_mel", [(8000.0, 25.0, 40), (16000.0, 20.0, 80), (16000.0, 32.0, 64), (16000.0, 50.0, 80), (8000.0, 10.0, 23)])
def test_fbank_other_sample_rates_and_frame_lengths(rate, frame_ms, n_mel):
    """dataset.py:93-100 passes the corpus's sample_frequency: 8 kHz telephone speech is a 200-sample window on a 256-point
    FFT; other frame lengths give 128 / 512 / 1024 points (kaldi rounds the window up to a power of two)."""
    torch.manual_seed(31)
    n = int(rate * 1.3)
    wav = (torch.rand(2, n) - 0.5) * 0.8
    lens = [n, int(0.61 * n)]
    fb = Fbank(n_mel, sample_rate=rate, frame_length_ms=frame_ms, device=DEV)
    feats, nfr = fb(wav.to(DEV), torch.tensor(lens, device=DEV))
    torch.cuda.synchronize()
    for b, ln in enumerate(lens):
        ref = FB.fbank(wav[b, :ln], num_mel_bins=n_mel, sample_rate=rate, frame_length_ms=frame_ms)
        assert int(nfr[b]) == ref.shape[0]
        np.testing.assert_allclose(feats[b, : ref.shape[0]].cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-3)
        assert torch.all(feats[b, ref.shape[0]:] == 0)


def tiny(seed=0, dropout=0.0):
    torch.manual_seed(seed)
    return ASRModel(80, 40, encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32,
                    attention_heads=4, linear_units=64, dropout_rate=dropout, ctc_weight=0.3, lsm_weight=0.1,
                    reverse_weight=0.3)


def batch_of(B=3, T=95, L=7, seed=1):
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=g)
    flen = torch.full((B,), T, dtype=torch.int32)
    tgt = torch.randint(2, 39, (B, L), generator=g, dtype=torch.int32)
    tlen = torch.full((B,), L, dtype=torch.int32)
    return {k: v.to(DEV) for k, v in dict(features=feats, features_length=flen, targets=tgt, targets_length=tlen).items()}


def check_updates(got, want, start, steps, lr=1e-3):
    """Adam moves every element by ~lr per step whatever the gradient's size, so elements whose true
    gradient is ~0 (e.g. the key bias: softmax is shift invariant) follow rounding noise.  Compare the
    parameters with a tolerance of a fraction of lr on (almost) all elements and a hard cap of the
    largest possible divergence, 2*lr*steps."""
    for k, v in got.items():
        a, b = v.detach().float().cpu(), want[k].detach().float().cpu()
        diff = (a - b).abs()
        assert float(diff.max()) <= 2.05 * lr * steps, (k, float(diff.max()))
        frac_bad = float((diff > 0.1 * lr * steps + 1e-3 * b.abs()).float().mean())
        assert frac_bad <= (1.0 if k.endswith("linear_k.bias") else 0.02), (k, frac_bad)


def test_arena_gradients_equal_autograd_gradients():
    """The same model with and without the flat arena: identical loss, identical gradients."""
    m1, m2 = tiny().to(DEV), tiny().to(DEV)
    b = batch_of()
    l1, _ = m1(**b)
    l1.backward()
    ar = A.ParamArena(m2).activate()
    try:
        ar.zero_grad()
        l2, _ = m2(**b)
        l2.backward()
        torch.cuda.synchronize()
        assert torch.equal(l1, l2)
        assert list(m2.state_dict().keys()) == list(m1.state_dict().keys())      # layout untouched
        for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert p2.grad is not None and p2.grad.data_ptr() == ar.by_ptr[p2.data_ptr()].data_ptr(), k
            torch.testing.assert_close(p2.grad, p1.grad, rtol=1e-4, atol=1e-5 * max(1.0, float(p1.grad.abs().max())),
                                       msg=lambda s, k=k: f"{k}: {s}")
        att = m2.encoder.encoders[0].self_attn
        assert ar.adjacent(att.linear_q.weight, att.linear_k.weight, att.linear_v.weight)
    finally:
        ar.deactivate()


def test_engine_step_matches_reference_training_step():
    """One engine step == the reference Executor's step (clip_grad_norm_ 5.0 + torch Adam) run by the
    CPU oracle on the same parameters and batch."""
    model = tiny(seed=5).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    b = batch_of(seed=2)
    eng = TrainEngine(model, lr=1e-3, grad_clip=5.0)
    try:
        loss, acc = eng.step(b)
        loss2, _ = eng.step(b)
        torch.cuda.synchronize()
    finally:
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
    cfg = O.Config(input_size=80, vocab_size=40, encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1,
                   d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, ctc_weight=0.3, lsm_weight=0.1,
                   reverse_weight=0.3)
    sd = {k: v.clone().requires_grad_() for k, v in sd0.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    cb = {k: v.cpu() for k, v in b.items()}
    ref_losses = []
    for _ in range(2):
        opt.zero_grad()
        l, _ = O.forward(sd, cfg, cb["features"], cb["features_length"], cb["targets"], cb["targets_length"])
        l.backward()
        torch.nn.utils.clip_grad_norm_(list(sd.values()), 5.0)
        opt.step()
        ref_losses.append(float(l))
    assert abs(float(loss) - ref_losses[0]) < 2e-4 * abs(ref_losses[0])
    assert abs(float(loss2) - ref_losses[1]) < 5e-4 * abs(ref_losses[1])
    check_updates(model.state_dict(), sd, sd0, steps=2)


def test_graph_replay_equals_eager_steps():
    m1, m2 = tiny(seed=7).to(DEV).train(), tiny(seed=7).to(DEV).train()
    b = batch_of(seed=3)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True)
    for _ in range(5):
        l_eager, _ = e1.step(b)
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True)
    try:
        e2.capture(b, warmup=2)                 # 2 warm-up steps happen inside capture()
        for _ in range(3):
            l_graph, _ = e2.replay()
        torch.cuda.synchronize()
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
    torch.testing.assert_close(l_graph, l_eager, rtol=1e-4, atol=1e-5)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=5)


@pytest.mark.parametrize("graph", [False, True])
def test_right_decoder_on_its_own_stream_gives_the_same_training(graph):
    """ops.PARALLEL_DECODERS: the right-to-left decoder and its loss head run on a second stream (one fork, one join,
    mirrored by autograd in backward).  Same losses and the same parameters after 5 steps as the single-stream engine,
    eager and as a captured HIP graph (the fork must be part of the capture)."""
    m1, m2 = tiny(seed=9).to(DEV).train(), tiny(seed=9).to(DEV).train()
    b = batch_of(seed=4)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True)
    for _ in range(5):
        l_seq, _ = e1.step(b)
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
    try:
        assert ops.PARALLEL_DECODERS
        if graph:
            e2.capture(b, warmup=2)
            for _ in range(3):
                l_par, _ = e2.replay()
        else:
            for _ in range(5):
                l_par, _ = e2.step(b)
        torch.cuda.synchronize()
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False
    torch.testing.assert_close(l_par, l_seq, rtol=1e-4, atol=1e-5)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=5)


def test_failed_capture_leaves_the_engine_usable(monkeypatch):
    """bench.py falls back to eager steps when a capture raises: after a capture that fails midway (here: at the end of
    backward, with weight gradients still deferred) an eager step of the same engine must equal a fresh engine's step."""
    m1, m2 = tiny(seed=13).to(DEV).train(), tiny(seed=13).to(DEV).train()
    b = batch_of(seed=8)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
    l_ref, _ = e1.step(b)
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
    try:
        real_flush = ops.ln_table_flush

        def boom():
            if e2._capturing:                                       # only inside the capture itself, not in its warm-up step
                raise RuntimeError("injected failure inside the capture")
            return real_flush()
        monkeypatch.setattr(ops, "ln_table_flush", boom)
        with pytest.raises(RuntimeError):                           # the injected error, or the runtime's complaint at capture end
            e2.capture(b, warmup=1)                                 # (the warm-up step is one real optimizer step)
        monkeypatch.undo()
        torch.cuda.synchronize()
        assert e2._graph is None and ops.WGRAD_DEFER == 0 and ops.LN_TABLE is None and not ops._deferred
        assert not e2._capturing
        l_eager, _ = e2.step(b)
        torch.cuda.synchronize()
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False
    e1.arena.activate()
    l_ref2, _ = e1.step(b)                                          # the reference engine's second step
    torch.cuda.synchronize()
    e1.arena.deactivate()
    torch.testing.assert_close(l_eager, l_ref2, rtol=1e-4, atol=1e-5)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=2)


def test_capture_refuses_a_fork_that_never_rejoined(monkeypatch):
    """TrainEngine.capture checks, before it ends the capture, that every stream the step forked onto has been led back
    (oe_capture_unjoined_streams walks the graph under construction).  The regular multi-stream configuration (side
    weight-gradient stream + right decoder + CTC head on their own streams) passes; a launch left on a fourth stream with no
    join is reported as an error instead of reaching capture_end unjoined, and the engine stays usable."""
    from openeat_amd import hip
    m1, m2 = tiny(seed=21).to(DEV).train(), tiny(seed=21).to(DEV).train()
    b = batch_of(seed=5)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
    calls = []
    real = hip.capture_unjoined_streams
    monkeypatch.setattr(hip, "capture_unjoined_streams", lambda o, s: calls.append((len(s), real(o, s))) or calls[-1][1])
    stray = torch.cuda.Stream()
    try:
        e1.capture(b, warmup=1)
        assert calls and calls[-1][0] >= 2 and calls[-1][1][0] == 0          # >= 2 forked streams known, none unjoined
        e1.replay()
        torch.cuda.synchronize()
        e1.drop_graph()
        e1.arena.deactivate()
        # the same engine configuration with one launch left behind on a stream nobody joins
        e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
        ops._extra_streams.append(stray)
        real_flush = ops.ln_table_flush
        scratch = torch.zeros(64, device=DEV)

        def flush_and_stray():
            if e2._capturing:
                stray.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stray):
                    scratch.add_(1.0)
            return real_flush()
        monkeypatch.setattr(ops, "ln_table_flush", flush_and_stray)
        with pytest.raises(RuntimeError, match="had not rejoined"):
            e2.capture(b, warmup=1)
        monkeypatch.setattr(ops, "ln_table_flush", real_flush)
        torch.cuda.synchronize()
        assert e2._graph is None and not e2._capturing
        l2, _ = e2.step(b)                                                    # still a working engine
        torch.cuda.synchronize()
        assert torch.isfinite(l2)
    finally:
        ops._extra_streams.clear()
        for e in (e1, locals().get("e2")):
            if e is not None:
                e.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False


def test_step_cached_replays_one_graph_per_batch_shape():
    """TrainEngine.step_cached (ragged training, dataset.py:337-364 buckets): the first batch of a shape runs eagerly and is
    captured, later batches of that shape replay the graph; the parameter trajectory equals the eager engine's on the same
    batch sequence; the cache is LRU-bounded; label padding by pad_targets changes nothing."""
    from openeat_amd.engine import pad_targets
    shapes = [dict(B=3, T=95, L=7), dict(B=2, T=131, L=9), dict(B=4, T=67, L=5)]
    seq = [0, 1, 0, 2, 1, 0, 2, 2, 1, 0]
    batches = []
    for i, k in enumerate(seq):
        b = batch_of(seed=40 + i, **shapes[k])
        b["features_length"] = b["features_length"] - torch.arange(b["features"].shape[0], dtype=torch.int32, device=DEV) * 4   # ragged
        b["targets_length"][-1] -= 2
        b["targets"][-1, -2:] = -1
        batches.append(b)
    m1, m2, m3 = (tiny(seed=31).to(DEV).train() for _ in range(3))
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True)
    ref = [float(e1.step(b)[0]) for b in batches]
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0)
    try:
        got = [float(e2.step_cached(dict(b, targets=pad_targets(b["targets"], 4)))[0]) for b in batches]
        torch.cuda.synchronize()
        assert (e2.cache_misses, e2.cache_hits) == (3, 7) and len(e2._cache) == 3
        assert all(r is not None for r in e2._cache.values())            # every shape captured
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
    for g, r in zip(got, ref):
        assert abs(g - r) < 2e-4 * abs(r) + 1e-4, (got, ref)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=len(seq))
    e3 = TrainEngine(m3, lr=1e-3, grad_clip=5.0, segmented=True)       # the multi-rank form of the capture, per shape, LRU of one
    try:
        got3 = [float(e3.step_cached(b, max_graphs=1)[0]) for b in batches]
        torch.cuda.synchronize()
        assert len(e3._cache) == 1 and e3.cache_hits == 1                # only "2, 2" repeats back to back
        e3.drop_graph()
        assert not e3._cache
    finally:
        e3.arena.deactivate()
        ops.set_seed_device_counter(None)
    for g, r in zip(got3, ref):
        assert abs(g - r) < 2e-4 * abs(r) + 1e-4
    check_updates(m3.state_dict(), m1.state_dict(), None, steps=len(seq))


def test_step_cached_reports_shapes_that_do_not_capture():
    """A capture refusal leaves the shape on the eager path WITH a warning and a counter (once per shape); a fork that never
    rejoined or a device error is not a property of the shape and is re-raised."""
    import warnings
    m = tiny(seed=33).to(DEV).train()
    e = TrainEngine(m, lr=1e-3, grad_clip=5.0)
    b = batch_of(seed=50, B=2, T=83, L=6)
    real_capture = e.capture
    try:
        def refuse(*a, **k):
            raise RuntimeError("operation not permitted when stream is capturing")
        e.capture = refuse
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            l1 = float(e.step_cached(b)[0])                       # first sight: eager step + refused capture
            l2 = float(e.step_cached(b)[0])                       # known-uncapturable shape: eager, no second warning
        assert e.cache_uncapturable == 1 and sum("does not capture" in str(x.message) for x in w) == 1
        assert l2 < l1 and e.cache_hits == 0

        def broken(*a, **k):
            raise RuntimeError("TrainEngine.capture: 1 forked stream(s) had not rejoined the capturing stream at the end of the step")
        e.capture = broken
        b2 = batch_of(seed=51, B=3, T=59, L=5)
        with pytest.raises(RuntimeError, match="had not rejoined"):
            e.step_cached(b2)
    finally:
        e.capture = real_capture
        e.arena.deactivate()
        ops.set_seed_device_counter(None)


@pytest.mark.parametrize("parallel", [False, True])
def test_segmented_capture_equals_eager_steps(parallel):
    """TrainEngine(segmented=True): the step captured as a chain of graphs cut at the encoder output and at the quarter
    points of the encoder stack (the multi-rank capture; on one rank the collectives between the graphs are no-ops).  Same
    losses and parameters as eager steps; the cuts are where the arena's units start."""
    from openeat_amd.models.asr_model import ASRModel

    def four_layers():
        torch.manual_seed(17)
        return ASRModel(80, 40, encoder_num_blocks=4, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
                        linear_units=64, dropout_rate=0.0, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3).to(DEV).train()
    m1, m2 = four_layers(), four_layers()
    b = batch_of(seed=6)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True)
    for _ in range(4):
        l_eager, _ = e1.step(b)
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=parallel, segmented=True)
    try:
        sent = []
        e2.reducer.reduce_tail = lambda start: sent.append(start)
        e2.capture(b, warmup=1)
        us = e2.arena.unit_start
        assert [f for _, f in e2._segments] == [us["heads"], us["enc3"], us["enc2"], us["enc1"], None]
        for _ in range(3):
            l_graph, _ = e2.replay()
        torch.cuda.synchronize()
        assert sent == [us["heads"], us["enc3"], us["enc2"], us["enc1"]] * 3
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False
    torch.testing.assert_close(l_graph, l_eager, rtol=1e-4, atol=1e-5)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=4)


def test_two_graph_form_of_the_segmented_capture_equals_eager_steps():
    """TrainEngine.set_overlap_cuts([k], heads=False): the multi-rank step as TWO graphs cut at encoder layer k's input (bench.py
    tries this form, the chain of five and the single graph at N > 1 and times the fastest).  The one tail handed to the collective
    between the replays starts where that layer's arena unit does; losses and parameters equal eager steps; the default cut set
    comes back with set_overlap_cuts()."""
    from openeat_amd.models.asr_model import ASRModel

    def four_layers():
        torch.manual_seed(17)
        return ASRModel(80, 40, encoder_num_blocks=4, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
                        linear_units=64, dropout_rate=0.0, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3).to(DEV).train()
    m1, m2 = four_layers(), four_layers()
    b = batch_of(seed=6)
    e1 = TrainEngine(m1, lr=1e-3, grad_clip=5.0, static_shapes=True)
    for _ in range(4):
        l_eager, _ = e1.step(b)
    torch.cuda.synchronize()
    e1.arena.deactivate()
    e2 = TrainEngine(m2, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True, segmented=True)
    try:
        sent = []
        e2.reducer.reduce_tail = lambda start: sent.append(start)
        e2.set_overlap_cuts([1], heads=False)
        e2.capture(b, warmup=1)
        us = e2.arena.unit_start
        assert [f for _, f in e2._segments] == [us["enc1"], None]
        for _ in range(3):
            l_graph, _ = e2.replay()
        torch.cuda.synchronize()
        assert sent == [us["enc1"]] * 3
        e2.set_overlap_cuts()                      # the default again: encoder output + quarter points (graph dropped)
        assert e2._graph is None and sorted(m2.encoder.grad_ready_hooks) == [1, 2, 3] and "encoder_out" in m2.grad_ready_hooks
    finally:
        e2.arena.deactivate()
        ops.set_seed_device_counter(None)
        ops.PARALLEL_DECODERS = False
        ops.POS_PROJ_AHEAD = False
    torch.testing.assert_close(l_graph, l_eager, rtol=1e-4, atol=1e-5)
    check_updates(m2.state_dict(), m1.state_dict(), None, steps=4)


def test_dropout_training_step_runs_and_is_seed_dependent():
    m = tiny(seed=9, dropout=0.1).to(DEV).train()
    b = batch_of(seed=4)
    l1, _ = m(**b)
    l2, _ = m(**b)
    m.eval()
    l3, _ = m(**b)
    l4, _ = m(**b)
    torch.cuda.synchronize()
    assert torch.isfinite(l1) and l1 != l2 and torch.equal(l3, l4)


@pytest.mark.parametrize("prec", [6, 3, 1])
def test_arena_with_bf16_gemms_and_fused_bias_gradients(prec):
    """Arena path on the bf16 matrix cores (bias gradients are column sums fused into the wgrad GEMM):
    gradients agree with the exact-fp32 autograd path within the mode's tolerance."""
    from openeat_amd import hip
    m1, m2 = tiny().to(DEV), tiny().to(DEV)
    b = batch_of()
    l1, _ = m1(**b)
    l1.backward()
    ar = A.ParamArena(m2).activate()
    old = hip.GEMM_PRECISION
    hip.GEMM_PRECISION = prec
    try:
        ar.zero_grad()
        l2, _ = m2(**b)
        l2.backward()
        torch.cuda.synchronize()
    finally:
        hip.GEMM_PRECISION = old
        ar.deactivate()
    tol = 2e-4 if prec in (3, 6) else 3e-2
    assert abs(float(l1) - float(l2)) <= tol * abs(float(l1))
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        scale = max(1e-3, float(p1.grad.abs().max()))
        err = float((p2.grad - p1.grad).abs().max()) / scale
        assert err <= (5e-3 if prec in (3, 6) else 0.25), (k, err)


def test_spec_augment_on_device_matches_reference_bit_for_bit():
    """SpecAugment + spec-substitute on the padded batch (kernels in augment.hip, draws on the host in the reference's
    order): identical to the reference's per-utterance numpy result for the same random seed; padding untouched."""
    import random
    from conftest import load_golden
    from openeat_amd.augment import spec_augment_, spec_substitute_
    g = load_golden("f17_spec_augment")
    xs = [g["in"][f"x{i}"] for i in range(3)]
    lens = [x.shape[0] for x in xs]
    Tmax = max(lens) + 5

    def batch():
        b = torch.full((3, Tmax, 80), 7.0)
        for i, x in enumerate(xs):
            b[i, : lens[i]] = x
        return b.to(DEV)
    random.seed(17)
    y = spec_augment_(spec_substitute_(batch(), lens, max_t=20, num_t_sub=3), lens, num_t_mask=2, num_f_mask=2, max_t=50, max_f=10).cpu()
    for i in range(3):
        assert torch.equal(y[i, : lens[i]], g["out"][f"y{i}"])
        assert bool((y[i, lens[i]:] == 7.0).all())
    random.seed(18)
    z = spec_augment_(batch(), lens, num_t_mask=3, num_f_mask=1, max_t=10, max_f=30).cpu()
    for i in range(3):
        assert torch.equal(z[i, : lens[i]], g["out_aug_only"][f"z{i}"])
    # config-2 size (32 x 998 x 80): masked cells are exactly the union of the drawn stripes
    torch.manual_seed(0)
    big = torch.randn(32, 998, 80, device=DEV) + 3.0
    ref = big.clone()
    random.seed(5)
    out = spec_augment_(big, [998] * 32).cpu()
    zero = out == 0
    assert 0.02 < float(zero.float().mean()) < 0.35
    assert torch.equal(out[~zero], ref.cpu()[~zero])
    rows_all = zero.all(2)
    cols_all = zero.all(1)
    assert bool((zero == (rows_all[:, :, None] | cols_all[:, None, :])).all())


def test_feature_dither_on_device_distribution():
    """dataset.py:197-201: x + (U[0,1) - 0.5) * a with a = random.uniform(0, feature_dither): amplitude drawn like the
    reference, noise uniform in [-a/2, a/2) with the right moments, padding untouched, deterministic for a seed."""
    import random
    from openeat_amd.augment import feature_dither_
    lens = [998, 700, 333, 998]
    base = torch.zeros(4, 998, 80, device=DEV)
    random.seed(9)
    a = random.uniform(0, 0.5)
    random.seed(9)
    y = feature_dither_(base.clone(), lens, 0.5).cpu()
    random.seed(9)
    y2 = feature_dither_(base.clone(), lens, 0.5).cpu()
    assert torch.equal(y, y2)
    for b, n in enumerate(lens):
        v = y[b, :n]
        assert float(v.min()) >= -a / 2 - 1e-6 and float(v.max()) < a / 2 + 1e-6
        assert abs(float(v.mean())) < a * 0.01 and abs(float(v.var()) - a * a / 12) < a * a / 12 * 0.03
        assert bool((y[b, n:] == 0).all())
    assert not torch.equal(y[0, :333], y[2, :333])


@pytest.mark.parametrize("mode", ["eager", "captured", "cached"])
def test_gradient_accumulation_matches_reference_loop(mode):
    """TrainEngine(accum_grad=2): two micro-steps, ONE optimizer step whose gradient is the sum of the two micro-batch
    gradients of loss / 2 (executor.py:42-63) - checked against the CPU oracle doing exactly that with torch Adam, over
    three accumulated steps.  Between the boundaries nothing is exchanged, clipped or applied.
    eager: TrainEngine.step.  captured: the first optimizer step eager, then the micro-step graph + the clip/Adam graph
    (TrainEngine.capture / replay).  cached: step_cached all the way (first micro-step eager + capture, the rest replays)."""
    model = tiny(seed=21).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batches = [batch_of(seed=30 + i) for i in range(6)]
    eng = TrainEngine(model, lr=1e-3, grad_clip=5.0, accum_grad=2, static_shapes=(mode != "eager"))
    try:
        before = eng.arena.flat.clone()
        first = eng.step_cached if mode == "cached" else eng.step
        l0 = first(batches[0])[0].clone()                           # (a replay returns the graph's static output tensor)
        assert torch.equal(eng.arena.flat, before), "a non-boundary micro-step must not move the parameters"
        assert float(eng.optimizer.step_state[0]) == 0.0
        l1 = first(batches[1])[0].clone()
        assert float(eng.optimizer.step_state[0]) == 1.0 and not torch.equal(eng.arena.flat, before)
        if mode == "captured":
            eng.capture(batches[0], _warm=True)                    # executes nothing: the trajectory is untouched
            nxt = eng.replay
        else:
            nxt = first
        losses = [l0, l1]
        for i in range(2, 6):
            mid = eng.arena.flat.clone()
            l, _ = nxt(batches[i])
            losses.append(l.clone())
            if i % 2 == 0:
                assert torch.equal(eng.arena.flat, mid), "a non-boundary micro-step must not move the parameters"
        torch.cuda.synchronize()
        assert float(eng.optimizer.step_state[0]) == 3.0
        if mode == "cached":
            assert (eng.cache_misses, eng.cache_hits) == (1, 5) and eng.cache_uncapturable == 0
    finally:
        eng.arena.deactivate()
        ops.set_seed_device_counter(None)
    cfg = O.Config(input_size=80, vocab_size=40, encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1,
                   d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, ctc_weight=0.3, lsm_weight=0.1,
                   reverse_weight=0.3)
    sd = {k: v.clone().requires_grad_() for k, v in sd0.items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-3)
    ref = []
    for i, b in enumerate(batches):
        cb = {k: v.cpu() for k, v in b.items()}
        if i % 2 == 0:
            opt.zero_grad()
        l, _ = O.forward(sd, cfg, cb["features"], cb["features_length"], cb["targets"], cb["targets_length"])
        (l / 2).backward()
        ref.append(float(l) / 2)
        if i % 2 == 1:
            torch.nn.utils.clip_grad_norm_(list(sd.values()), 5.0)
            opt.step()
    for got, want in zip(losses, ref):
        assert abs(float(got) - want) < 5e-4 * abs(want), (float(got), want)
    check_updates(model.state_dict(), sd, sd0, steps=3)


def test_eager_step_beside_a_captured_graph_uses_the_current_learning_rate():
    """After capture(), an eager step with a new lr must apply that lr (not the one the last replay left on the device),
    and a replay without an lr argument must apply the optimizer's current one."""
    m = tiny(seed=23).to(DEV).train()
    b = batch_of(seed=5)
    e = TrainEngine(m, lr=1e-3, grad_clip=5.0, static_shapes=True)
    try:
        e.capture(b, warmup=1)
        e.replay(lr=1e-3)
        p0 = e.arena.flat.clone()
        e.step(b, lr=0.0)                                   # eager, lr 0: parameters must stay put
        torch.cuda.synchronize()
        assert torch.equal(e.arena.flat, p0)
        e.replay()                                          # lr is still 0 in param_groups
        torch.cuda.synchronize()
        assert torch.equal(e.arena.flat, p0)
        e.replay(lr=1e-3)
        torch.cuda.synchronize()
        assert not torch.equal(e.arena.flat, p0)
    finally:
        e.arena.deactivate()
        ops.set_seed_device_counter(None)


def test_arena_weight_planes_are_never_stale():
    """planes.arena_weight (the weight operand of csrc/gemm_hyb.hip): the arena's bf16 planes are split again by the first reader of
    every forward pass and after every optimizer step - eager, captured and replayed - because Adam moves the weights through a raw
    kernel no tensor version counter sees.  p0 + p1 + p2 must equal the CURRENT weight exactly at every point a GEMM could read it."""
    from openeat_amd import hip, planes
    old = (hip.GEMM_PRECISION, planes.HYB_MIN_ROWS)
    hip.GEMM_PRECISION, planes.HYB_MIN_ROWS = 6, 0
    m = tiny(seed=21).to(DEV).train()
    e = TrainEngine(m, lr=1e-2, grad_clip=5.0, static_shapes=True)
    b = batch_of(seed=8)
    w = m.encoder.encoders[0].feed_forward.w_1.weight

    def planes_equal_weight():
        pl = planes.arena_weight(w)
        assert pl is not None and pl.rows == w.shape[0] and pl.cols == w.shape[1]
        off = (pl.ptr - e.arena.planes.data_ptr()) // 2
        p = torch.stack([e.arena.planes[n, off:off + w.numel()].view_as(w).float() for n in range(3)])
        torch.cuda.synchronize()
        return torch.equal(p[0] + p[1] + p[2], w.detach())
    try:
        assert planes.weights_presplit()
        e.step(b)
        w0 = w.detach().clone()
        assert planes_equal_weight()                      # right after an eager optimizer step (first reader splits)
        e.step(b)
        assert not torch.equal(w.detach(), w0) and planes_equal_weight()
        e.capture(b, warmup=1)
        for _ in range(3):
            e.replay()
        assert planes_equal_weight()                      # after replays (their Adam ran without any Python)
        with torch.no_grad():
            w.mul_(1.5)                                   # a torch write (a loaded checkpoint, an average): seen by the next pass
        planes.new_pass()                                 # ... which every forward / decode entry point announces
        assert planes_equal_weight()
    finally:
        hip.GEMM_PRECISION, planes.HYB_MIN_ROWS = old
        e.arena.deactivate()
        ops.set_seed_device_counter(None)
        planes.clear()


@pytest.mark.parametrize("policy", ["ln", "all", "conv"])
def test_weight_planes_follow_an_optimizer_loop_outside_the_engine(policy):
    """The reference-shaped loop of utils/executor.py:42-63 - model(**batch), loss.backward(), optimizer.step() - steps FusedAdam
    without TrainEngine._finish.  The raw Adam kernel bumps no tensor version, so the arena's bf16 planes must go stale by the
    optimizer's own doing (FusedAdam.step -> ParamArena.mark_step) and be honoured by every reader under every planes policy
    (round 3: under "ln" / "all" such a loop kept training on the weights of the first split).  Each step's loss
    must equal the same loop's without any pre-split operand."""
    from openeat_amd import hip, planes
    from openeat_amd.optim import FusedAdam
    old = (hip.GEMM_PRECISION, planes.POLICY, planes.MIN_SPLIT_ELEMS, planes.HYB_MIN_ROWS)
    b = batch_of(seed=12)
    losses = {}
    try:
        for pol in ("0", policy):
            hip.GEMM_PRECISION, planes.POLICY, planes.MIN_SPLIT_ELEMS, planes.HYB_MIN_ROWS = 6, pol, 0, 0
            planes.clear_all()
            m = tiny(seed=23).to(DEV).train()
            ar = A.ParamArena(m).activate()
            opt = FusedAdam(ar, lr=2e-2, max_grad_norm=5.0)                  # a large rate: stale weights show in the next loss
            out = []
            for _ in range(4):
                opt.zero_grad()
                loss, _ = m(**b)
                loss.backward()
                ops.join_side_stream()
                opt.step()
                out.append(float(loss))
            if pol != "0":
                w = m.encoder.encoders[0].feed_forward.w_1.weight
                pl = planes.arena_weight(w)                                   # a reader after the last step: split again
                assert pl is not None
                off = (pl.ptr - ar.planes.data_ptr()) // 2
                p = torch.stack([ar.planes[n, off:off + w.numel()].view_as(w).float() for n in range(3)])
                torch.cuda.synchronize()
                assert torch.equal(p[0] + p[1] + p[2], w.detach())
            losses[pol] = out
            ar.deactivate()
    finally:
        hip.GEMM_PRECISION, planes.POLICY, planes.MIN_SPLIT_ELEMS, planes.HYB_MIN_ROWS = old
        planes.clear_all()
    assert losses["0"][0] > losses["0"][-1]                                   # the loop does train
    np.testing.assert_allclose(losses[policy], losses["0"], rtol=5e-4)


def test_layernorm_backward_as_a_gemm_prologue_equals_the_separate_launches():
    """ops._PENDING_LN / _PENDING_LNF: at d = 256 and >= 4096 rows in precision 6 the pre-norm forks of an encoder layer run as prologues
    of the row-block kernels next to them - forward in front of the consuming feed-forward / q-k-v projection / pointwise_conv1, backward
    in front of the previous block's first input-gradient kernel (oe_rowgemm6's and oe_ffn_fwd / _bwd's ln / lnf arguments).  Same model, same batch, same
    dropout masks with the switch off and on: the gradient arena must agree to rounding (the two kernels contract a*b+c differently:
    one unit in the last place per element), the fused launches are counted, nothing stays parked."""
    from openeat_amd import hip
    old = (hip.GEMM_PRECISION, ops.LN_BWD_FUSE, ops.LN_FWD_FUSE, ops.LN_EPI_FUSE)
    hip.GEMM_PRECISION = 6
    grads, losses, launches, fwd_launches, epi_launches = [], [], [], [], []
    B, T = 11, 1530                                              # T' = 381: 11 x 381 = 4191 encoder rows
    try:
        for fuse in (False, True):
            ops.LN_BWD_FUSE = ops.LN_FWD_FUSE = ops.LN_EPI_FUSE = fuse
            torch.manual_seed(5)
            ops.manual_seed(17)                                  # the same dropout streams in both runs
            m = ASRModel(80, 40, encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=256, attention_heads=4,
                         linear_units=512, dropout_rate=0.1, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3).to(DEV).train()
            e = TrainEngine(m, lr=1e-3, grad_clip=5.0, static_shapes=True)
            try:
                b = batch_of(B=B, T=T, L=9, seed=4)
                n0, f0, e0 = ops.LN_BWD_FUSED_LAUNCHES, ops.LN_FWD_FUSED_LAUNCHES, ops.LN_EPI_FUSED_LAUNCHES
                e.arena.zero_grad()
                loss, _ = e._fwd_bwd(b)
                torch.cuda.synchronize()
                assert not ops._PENDING_LN
                assert not ops._PENDING_LNF
                launches.append(ops.LN_BWD_FUSED_LAUNCHES - n0)
                fwd_launches.append(ops.LN_FWD_FUSED_LAUNCHES - f0)
                epi_launches.append(ops.LN_EPI_FUSED_LAUNCHES - e0)
                grads.append(e.arena.grad.detach().clone())
                losses.append(float(loss))
            finally:
                e.arena.deactivate()
                ops.set_seed_device_counter(None)
    finally:
        hip.GEMM_PRECISION, ops.LN_BWD_FUSE, ops.LN_FWD_FUSE, ops.LN_EPI_FUSE = old
    assert fwd_launches == [0, 8]                                # layer 0's macaron norm, the pair in front of layer 1's, per layer: norm_mha, norm_conv, norm_ff
    assert epi_launches == [0, 2]                                # the conv module's own norm, behind pointwise_conv2's input gradient
    assert launches == [0, 8]                                    # two encoder layers x (attention, conv module, both feed-forwards)
    assert abs(losses[0] - losses[1]) <= 2e-6 * abs(losses[0])    # (forward: the same norms computed in other kernels)
    scale = float(grads[0].abs().max())
    assert float((grads[1] - grads[0]).abs().max()) <= 2e-5 * scale
    assert float((grads[1] - grads[0]).norm()) <= 1e-5 * float(grads[0].norm())


def test_inference_with_the_layernorm_prologues_equals_the_separate_launches():
    """The decode entry points run the same fused forward (no_grad, eval): encoder output and CTC greedy ids with ops.LN_FWD_FUSE on
    and off at a size where the prologues are active (>= 4096 encoder rows, d = 256, precision 6), eagerly and from a captured graph."""
    from openeat_amd import hip
    old = (hip.GEMM_PRECISION, ops.LN_FWD_FUSE)
    hip.GEMM_PRECISION = 6
    torch.manual_seed(6)
    m = ASRModel(80, 40, encoder_num_blocks=3, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=256, attention_heads=4,
                 linear_units=512, dropout_rate=0.1, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3).to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    B, T = 11, 1530
    feats = torch.randn(B, T, 80, generator=g).to(DEV)
    nfr = torch.tensor([T, T - 40, T, T - 200, T, T, T - 8, T, T, T - 333, T], dtype=torch.int32, device=DEV)
    outs, ids, fused = [], [], []
    try:
        with torch.no_grad():
            for fuse in (False, True):
                ops.LN_FWD_FUSE = fuse
                f0 = ops.LN_FWD_FUSED_LAUNCHES
                enc, mask, _ = m._encode(feats, nfr)
                torch.cuda.synchronize()
                assert not ops._PENDING_LNF
                fused.append(ops.LN_FWD_FUSED_LAUNCHES - f0)
                outs.append(enc.clone())
                ids.append(m.ctc_greedy_search(feats, nfr))
            # ... and from a captured graph (what the batched decode paths replay)
            ops.LN_FWD_FUSE = True
            static = feats.clone()
            gph = torch.cuda.CUDAGraph()
            from openeat_amd import planes
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                m._encode(static, nfr)
                torch.cuda.synchronize()
                with planes.capture_scope(), torch.cuda.graph(gph, stream=s):
                    enc_g, _, _ = m._encode(static, nfr)
            torch.cuda.synchronize()
            enc_g.fill_(float("nan"))
            gph.replay()
            torch.cuda.synchronize()
    finally:
        hip.GEMM_PRECISION, ops.LN_FWD_FUSE = old
    assert fused[0] == 0 and fused[1] >= 3 * 3
    scale = float(outs[0].abs().max())
    assert float((outs[1] - outs[0]).abs().max()) <= 2e-5 * scale
    assert ids[0] == ids[1]
    assert float((enc_g - outs[1]).abs().max()) <= 2e-5 * scale
