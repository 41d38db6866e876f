#!/usr/bin/env python3
"""Headline benchmark: training throughput of the 12-layer Conformer (BASELINE.json
configs[1]) on synthetic 16 kHz audio, hot path end to end on the GPU:

    wav (B x 10 s) -> fbank + per-utterance norm -> Conv2dSubsampling4 -> 12 x Conformer ->
    CTC head + bi-directional attention decoder -> joint loss -> backward -> gradient
    all-reduce (N > 1) -> clip + Adam.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0.  A "step" is one optimizer step on one batch of
B utterances per GPU; `value` = audio frames (10 ms) processed per second by the
whole job, inputs resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "audio-frames/sec/GPU (train) + RTF (attention-rescoring decode), 12L Conformer"
V = 3246
MODEL_CONF = dict(encoder_num_blocks=12, decoder_num_blocks=3, r_decoder_num_blocks=3, d_model=256, attention_heads=4,
                  linear_units=1024, dropout_rate=0.1, input_layer="conv2d", pos_enc_layer_type="rel_pos",
                  activation_type="swish", macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, causal=False,
                  ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3, length_normalized_loss=False)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md, dense bf16 MFMA (no sparsity)
TERMS = {0: 1, 1: 1, 3: 3, 6: 6}     # bf16 MFMAs issued per algorithmic product in each arithmetic mode
KERNEL_NAMES = {0: "gemm_f32_kernel (oe_gemm_f32, v_mfma_f32_32x32x2_f32)",
                6: "gemm_pl_kernel + gemm_dma_kernel + gemm_bf16_kernel + gemm_tn_planes/grouped_kernel + ffn6_kernel + rowgemm6/rowgemm6p/rowtile6_kernel <terms=6> "
                   "(the kernels behind oe_gemm_f32 / oe_gemm_tn_grouped / oe_ffn_fwd / oe_ffn_bwd / oe_rowgemm6 precision 6: three exact bf16 "
                   "pieces per operand, hh+hm+mh+mm+hl+lh on v_mfma_f32_32x32x16_bf16)",
                1: "gemm_dma_kernel + gemm_bf16_kernel + gemm_tn_planes/grouped_kernel + ffn_fwd_kernel <terms=1> (the kernels behind "
                   "oe_gemm_f32 / oe_gemm_tn_grouped / oe_ffn_fwd precision 1, v_mfma_f32_32x32x16_bf16)",
                3: "gemm_dma_kernel + gemm_bf16_kernel + gemm_tn_planes/grouped_kernel + ffn_fwd_kernel <terms=3> (the kernels behind "
                   "oe_gemm_f32 / oe_gemm_tn_grouped / oe_ffn_fwd precision 3: hi*hi+hi*lo+lo*hi, v_mfma_f32_32x32x16_bf16)"}
DTYPE_NAMES = {0: "f32", 1: "bf16 (MFMA inputs; fp32 storage, accumulate, softmax, norms, losses, optimizer)",
               6: "f32 (storage, accumulate, softmax, norms, losses, optimizer all fp32; every matrix product is the fp32 product to within "
                  "one fp32 rounding, computed as six exact bf16 x bf16 MFMA products of three-piece operand splits h+m+l = x, "
                  "dropped terms < 2^-24 |a||b| - tests hold it to the exact-fp32 mode's tolerances)",
               3: "bf16x3 (matrix products as hi*hi+hi*lo+lo*hi on bf16 MFMA: ~2^-17 relative error per product, fp32 accumulate - "
                  "narrower than the reference's fp32 products, wider than bf16; fp32 storage and fp32 everywhere else)"}


def synth_batch(B, seconds, L, seed, device):
    g = torch.Generator().manual_seed(seed)
    wav = (torch.rand(B, int(16000 * seconds), generator=g) - 0.5)
    tgt = torch.randint(2, V - 1, (B, L), generator=g, dtype=torch.int32)
    tlen = torch.full((B,), L, dtype=torch.int32)
    return wav.to(device), tgt.to(device), tlen.to(device)


def cpu_baseline(B, seconds, L, steps):
    """The oracle (CPU restatement of the reference, oracle/) timed on this box's host cores on a
    bounded sample of the same workload: same model, B utterances of `seconds` s, fwd+bwd+clip+Adam."""
    from oracle import asr as O
    from oracle import fbank as FB
    from openeat_amd.models.asr_model import ASRModel
    torch.manual_seed(777)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OE_CPU_BASELINE_THREADS", "16"))))   # the GPU box grants a 16-CPU share
    torch.set_num_threads(cores)
    model = ASRModel(80, V, **MODEL_CONF)                       # only used as a parameter container / initialiser
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    cfg = O.Config(input_size=80, vocab_size=V, **MODEL_CONF)
    params = [v for v in sd.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3)
    wav, tgt, tlen = synth_batch(B, seconds, L, 0, "cpu")
    frames = 0
    times = []
    warm = 2                                                     # 2 warm-up + 5 timed steps, median
    for it in range(steps + warm):
        t0 = time.perf_counter()
        feats = torch.stack([FB.utt_normalize(FB.fbank(w)) for w in wav])
        flen = torch.full((B,), feats.shape[1], dtype=torch.int32)
        loss, _ = O.forward(sd, cfg, feats, flen, tgt, tlen, training=True)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
        dt = time.perf_counter() - t0
        log(f"  cpu step {it}: {dt:.1f} s")
        if it >= warm:
            times.append(dt)
    med = sorted(times)[len(times) // 2]
    return {"value": B * feats.shape[1] / med, "unit": "audio-frames/s", "cores": cores, "kind": "port",
            "sample": f"B={B} x {seconds:g} s utterances (the GPU run's batch), fbank+fwd+bwd+clip+Adam, dropout 0.1: {warm} warm-up + {steps} "
                      f"timed steps, median step {med:.2f} s (BASELINE.md section 3 protocol); threads = this box's CPU share, capped at 16"}


def decode_cpu_baseline(model, lm, n_utt, seconds, beam, lm_weight):
    """The oracle's attention rescoring (oracle/asr.py: the reference's one-utterance algorithm, asr_model.py:418-534, incl. the
    pure-Python prefix recursion of :359-396) timed on this box's host cores on a bounded sample: n_utt utterances of the decode
    workload, fbank + per-utterance norm + encoder + prefix beam + bi-decoder + LM + scoring, one utterance at a time as the
    reference decodes.  RTF = wall seconds / audio seconds."""
    from oracle import asr as O
    from oracle import fbank as FB
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OE_CPU_BASELINE_THREADS", "16"))))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = O.Config(input_size=80, vocab_size=V, **MODEL_CONF)
    lm_sd = {k: v.detach().cpu().clone() for k, v in lm.state_dict().items()}
    lm_cfg = O.Config(vocab_size=V, macaron_style=False, use_cnn_module=False, pos_enc_layer_type="abs_pos", encoder_num_blocks=6,
                      d_model=256, attention_heads=4, linear_units=1024)
    g = torch.Generator().manual_seed(123)
    wav = (torch.rand(n_utt, int(16000 * seconds), generator=g) - 0.5)          # the first utterances of the GPU run's batch
    times, lens = [], []
    with torch.no_grad():
        for b in range(n_utt):
            t0 = time.perf_counter()
            feats = FB.utt_normalize(FB.fbank(wav[b])).unsqueeze(0)
            flen = torch.tensor([feats.shape[1]], dtype=torch.int32)
            hyp, _, _ = O.attention_rescoring(sd, cfg, feats, flen, beam, 0.5, 0.3, lm=(lm_sd, lm_cfg), lm_weight=lm_weight)
            times.append(time.perf_counter() - t0)
            lens.append(len(hyp))
            log(f"  cpu decode utterance {b}: {times[-1]:.1f} s, {lens[-1]} tokens")
    wall = sum(times[1:]) if n_utt > 1 else times[0]                             # the first utterance warms the thread pool up
    n = max(n_utt - 1, 1)
    return {"rtf": wall / (n * seconds), "unit": "wall s / audio s", "cores": cores, "kind": "port",
            "sample": f"{n} utterances x {seconds:g} s (after 1 warm-up utterance) of the GPU run's batch, one at a time as the reference "
                      f"decodes (asr_model.py:444 asserts batch 1): fbank + norm + encoder + Python prefix beam {beam} + bi-decoder "
                      "+ 6-layer LM + scoring, ctc 0.5 / reverse 0.3 / lm " + f"{lm_weight}; threads = this box's CPU share, capped at 16",
            "mean_best_len": sum(lens) / len(lens)}


def decode_rtf(model, fb, utt_norm, n_utt, seconds, beam, dev, reps=3, lm=None, lm_weight=0.0):
    """BASELINE.json configs[3]: attention-rescoring decode of n_utt synthetic utterances on one GPU
    (ctc_weight 0.5, reverse_weight 0.3 as in examples/aishell/run.sh:69-72; LM = the build-defined 6-layer d=256
    Transformer LM, lm_weight 0.3: the reference's own LanguageModel cannot be constructed).
    RTF = wall seconds / audio seconds, front end included."""
    model.eval()
    if lm is not None:
        lm.eval()
    # the training engine's stream forks (right decoder / CTC head / positional projections beside the main chain) pay for
    # 992-row launches; the decode batch is 640 hypotheses x 200+ tokens of chip-filling launches: one stream
    from openeat_amd import ops as _ops
    forks = (_ops.PARALLEL_DECODERS, _ops.POS_PROJ_AHEAD)
    _ops.PARALLEL_DECODERS = _ops.POS_PROJ_AHEAD = False
    g = torch.Generator().manual_seed(123)
    wav = (torch.rand(n_utt, int(16000 * seconds), generator=g) - 0.5).to(dev)
    times = []
    hyps = None
    for it in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feats, nfr = fb(wav)
        utt_norm(feats, nfr)
        hyps = model.attention_rescoring_batch(feats, nfr, beam, ctc_weight=0.5, reverse_weight=0.3, lm=lm, lm_weight=lm_weight,
                                               use_graphs=os.environ.get("OE_BENCH_DECODE_GRAPHS", "1") == "1")      # both stages replayed from per-shape HIP graphs after the first call
        torch.cuda.synchronize()
        if it > 0:
            times.append(time.perf_counter() - t0)
    model.train()
    _ops.PARALLEL_DECODERS, _ops.POS_PROJ_AHEAD = forks
    best = min(times)
    return {"rtf": best / (n_utt * seconds), "wall_s": best, "utterances": n_utt, "seconds_each": seconds, "beam": beam,
            "hip_graph": os.environ.get("OE_BENCH_DECODE_GRAPHS", "1") == "1",      # both stages replayed from per-shape graphs (the library default is eager)
            "ctc_weight": 0.5, "reverse_weight": 0.3,
            "lm": None if lm is None else "6-layer d=256 h=4 ff=1024 Transformer LM (seeded init), shallow fusion", "lm_weight": lm_weight,
            "mean_best_len": sum(len(h) for h in hyps) / len(hyps),
            "mean_nbest_len": getattr(model, "last_nbest_mean_len", None),
            "note": "seeded random-init weights: the n-best lists and the rescored pick are whatever an untrained model emits "
                    "(an untrained CTC head emits far more tokens than speech has - the prefix recursion and the decoder "
                    "batch both scale with mean_nbest_len)"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--target-len", type=int, default=30)
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-graph-multi", dest="graph_multi", action="store_false",
                    help="N > 1: do not try the HIP-graph replay of forward+backward, time eager steps only")
    ap.add_argument("--force-graph", action="store_true", help="time the HIP-graph replay even if the eager step calibrated faster")
    ap.add_argument("--no-single-graph", action="store_true", help="N > 1: do not try the one-graph + one-all-reduce form beside the chain of five graphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-table", action="store_true", help="log the per-problem GEMM timing table of one step to stderr")
    ap.add_argument("--dropout", type=float, default=None, help="tuning aid: override the config's dropout 0.1 (the reported line is only valid at the default)")
    ap.add_argument("--no-decode", action="store_true", help="skip the attention-rescoring RTF measurement")
    ap.add_argument("--serial-decoders", action="store_true", help="tuning: keep the right-to-left decoder on the main stream")
    ap.add_argument("--single-stream", action="store_true",
                    help="profiling: every kernel on one stream (no decoder / CTC / weight-gradient streams), so that a profiler's "
                         "per-kernel durations are those of the kernel alone, as the live roofline measurement takes them")
    ap.add_argument("--decode-utts", type=int, default=64)
    ap.add_argument("--cpu-batch", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-decode-utts", type=int, default=16, help="utterances of the decode workload the CPU baseline rescoring times (the first warms up)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the extra ms/step measurements in arithmetic modes 0 (fp32) and 1 (bf16)")
    args = ap.parse_args()
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner at its first collective):
    # keep the real stdout for the line alone and send everything else that reaches descriptor 1 to stderr.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from openeat_amd import ddp, hip
    hip.GEMM_PRECISION = int(os.environ.get("OE_GEMM_PRECISION", "6"))      # default: the reference-precision 6-term bf16 split
    rank, local, world = ddp.init_from_env()
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (openeat_amd has no CPU path)"
    local = local % torch.cuda.device_count()        # rehearsals may put several ranks on one GPU (gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    hip.lib()

    from openeat_amd.engine import TrainEngine
    from openeat_amd.frontend import Fbank, utt_normalize_
    from openeat_amd.models.asr_model import ASRModel

    log(f"world={world} device={torch.cuda.get_device_name(dev)}")
    torch.manual_seed(777)
    conf = dict(MODEL_CONF)
    if args.dropout is not None:
        conf["dropout_rate"] = args.dropout
        log(f"NOTE: dropout overridden to {args.dropout} - not the BASELINE config, tuning only")
    model = ASRModel(80, V, **conf).to(dev).train()
    engine = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, async_wgrad=not args.single_stream,
                         parallel_decoders=not (args.serial_decoders or args.single_stream))
    if args.single_stream:
        # profile runs (`--single-stream --no-graph`): the eager step launches what the captured step holds, one kernel after
        # the other - weight gradients collected in groups of 48, the small ones of a group as one grouped launch
        from openeat_amd import ops as _ops0
        _ops0.WGRAD_DEFER, _ops0.WGRAD_FLUSH_INLINE = 48, True
    fb = Fbank(80, device=dev)
    wav, tgt, tlen = synth_batch(args.batch, args.seconds, args.target_len, seed=rank, device=dev)
    T = fb.num_frames(wav.shape[1])
    feats = torch.empty(args.batch, T, 80, device=dev)
    flen = torch.full((args.batch,), T, dtype=torch.int32, device=dev)

    class WithFrontend(torch.nn.Module):
        """fbank + per-utterance normalisation in front of the model: the whole hot path is one step."""
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, wav, targets, targets_length):
            fb(wav, out=feats)
            utt_normalize_(feats, flen)
            return self.m(feats, flen, targets, targets_length)

    engine.model = WithFrontend(model)
    batch = {"wav": wav, "targets": tgt, "targets_length": tlen}
    # Step launch mode: both are tried during warm-up and the faster one is timed (the decision is the same on every
    # rank: MAX over ranks of each timing).  Graph replay has no host work; the eager step is host-bound on some boxes
    # but overlaps more: at N = 1 the side-stream weight gradients, at N > 1 the gradient all-reduce, which the autograd
    # hooks start inside backward (with a graph the collectives run after the replayed forward+backward instead).
    log("first eager step ...")
    l0 = engine.step(batch)[0]
    torch.cuda.synchronize()
    log(f"first eager step done, loss={float(l0):.4f}")
    eager = lambda: engine.step(batch)

    def agree(ms):                                   # same number on every rank
        if not torch.distributed.is_initialized():
            return ms
        t = torch.tensor([ms], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t)

    def trial(f, n, give_up_ms=None):
        f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        f()
        torch.cuda.synchronize()
        first = agree((time.perf_counter() - t) * 1e3)
        if give_up_ms is not None and first > give_up_ms:       # hopeless (e.g. a host-staged collective backend): stop here
            return first
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return agree((time.perf_counter() - t) / n * 1e3)

    use_graph, graph_form = False, None
    # N > 1: the step is captured as a chain of five graphs cut at the encoder output and the quarter points of the encoder
    # stack (TrainEngine._capture_segments); the all-reduce of the arena tail that a graph has finished is issued right
    # after its replay and runs beside the next graph, clip + Adam follow the last collective - against an eager step that
    # overlaps the same way from autograd hooks but is host-bound (~1100 launches, 23-28 ms on the boxes seen).  Both are
    # tried and the faster is timed; the capture runs in thread-local error mode (RCCL's watchdog thread polls events
    # meanwhile) and any capture error falls back to eager.  Rehearsed with 2 ranks over gloo on one GPU
    # (profiles/r02_multi_gpu_rehearsal.md); RCCL itself needs > 1 GPU.
    if not args.no_graph and (world == 1 or args.graph_multi):
        n_trial = 10 if not torch.distributed.is_initialized() else 5
        t_eager = trial(eager, n_trial)              # before the capture: the graph's private memory pool changes allocator state
        try:
            engine.capture(batch, warmup=max(1, args.warmup))
            log("HIP graph captured")
            t_graph = trial(engine.replay, n_trial, give_up_ms=3.0 * t_eager)
        except Exception as e:                        # capture trouble must never cost the run: stay eager
            log(f"graph capture/replay failed ({type(e).__name__}: {e}); staying eager")
            t_graph = float("inf")
        graph_form = "one graph" if not engine.reducer.active else "five graphs, all-reduces of the finished arena tails between the replays"
        # Several ranks: two more forms of the captured step are tried beside the chain of five graphs, and the fastest is timed.
        # The chain hides the collectives but loses ~0.35 ms of overlap at each of its four cuts (one rank over RCCL: 20.4 against
        # 19.05 ms/step for one graph); over xGMI the 125 MB all-reduce may be cheaper than that - measured here, not assumed:
        #   two graphs: one cut below the first quarter of the encoder; the all-reduce of everything above it (~80 % of the arena)
        #               runs beside the second graph, a small one follows it;
        #   one graph:  the whole step, then ONE all-reduce of the gradient arena, clip + Adam.
        if engine.reducer.active and engine.segmented and t_graph != float("inf") and not args.no_single_graph:
            n_enc = len(model.encoder.encoders)
            forms = {"five": (None, True, True, graph_form),
                     "two": ([max(1, n_enc // 4)], False, True, "two graphs cut below encoder layer %d, the all-reduce of the upper arena beside the second" % max(1, n_enc // 4)),
                     "one": (None, True, False, "one graph, then one all-reduce of the gradient arena, clip + Adam")}

            def capture_form(name):
                cuts, heads, seg, _ = forms[name]
                engine.set_overlap_cuts(cuts, heads)           # (drops the current graph and its pool)
                engine.segmented = seg
                engine.capture(batch, warmup=1)

            times = {"five": t_graph}
            current = "five"
            for name in ("two", "one"):
                current = None                                # (capture_form drops whatever graph the engine held)
                try:
                    capture_form(name)
                    current = name
                    times[name] = trial(engine.replay, n_trial, give_up_ms=3.0 * t_eager)
                except Exception as e:
                    log(f"capture/replay of the '{name}' form failed ({type(e).__name__}: {e})")
                    times[name] = float("inf")
            best = min(times, key=times.get)
            log("warm-up calibration of the captured forms: " + ", ".join(f"{k} {v:.2f}" for k, v in times.items()) + f" ms/step -> {best}")
            if best != current:
                try:
                    capture_form(best)
                except Exception as e:
                    log(f"re-capture of the '{best}' form failed ({type(e).__name__}: {e}); staying eager")
                    times[best] = float("inf")
                    engine.drop_graph()
            t_graph, graph_form = times[best], forms[best][3]
        use_graph = (args.force_graph and t_graph != float("inf")) or t_graph <= t_eager
        log(f"warm-up calibration: eager {t_eager:.2f} ms/step, graph replay {t_graph:.2f} ms/step -> timing {'graph' if use_graph else 'eager'}")
        if not use_graph:
            engine.drop_graph()
            if engine.reducer.active:
                engine.set_overlap_cuts()             # the eager step's hooks at their default places again
    run = (lambda: engine.replay()) if use_graph else eager
    if not use_graph:
        for _ in range(args.warmup):
            run()

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(out[0])
    log(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * T / (elapsed / args.steps)

    # ---- roofline of the dominant kernel (fp32 MFMA GEMM), measured live with HIP events on the launch stream
    roof = None
    if rank == 0:
        hip.PROFILE = []
    # the bracketed steps run on ONE stream: with the decoder / CTC / weight-gradient streams active, kernels of different
    # streams share the chip and an event pair would time that sharing, not the kernel (rocprofv3 serialises them too)
    from openeat_amd import ops as _ops
    saved_streams = (_ops.PARALLEL_DECODERS, _ops.ASYNC_WGRAD, engine.parallel)
    _ops.PARALLEL_DECODERS, _ops.ASYNC_WGRAD, engine.parallel = False, False, False
    # ... and with the captured step's launch structure: weight gradients collected in groups of 48, the small ones of a
    # group as one grouped launch, flushed in line
    saved_defer = (_ops.WGRAD_DEFER, _ops.WGRAD_FLUSH_INLINE)
    _ops.WGRAD_DEFER, _ops.WGRAD_FLUSH_INLINE = 48, True
    for _ in range(2):
        # park the GPU behind a ~0.15 s spin kernel while the host enqueues the whole eager step: the launches then run
        # back to back and an event pair measures the kernel, not the host's time between record() and launch
        torch.cuda._sleep(int(3.5e8))
        engine.step(batch)                                      # eager steps on EVERY rank (they contain the collective);
    torch.cuda.synchronize()                                    # rank 0 brackets each GEMM launch with events
    _ops.PARALLEL_DECODERS, _ops.ASYNC_WGRAD, engine.parallel = saved_streams
    _ops.WGRAD_DEFER, _ops.WGRAD_FLUSH_INLINE = saved_defer
    if rank == 0:
        recs, hip.PROFILE = hip.PROFILE, None
        recs = recs[len(recs) // 2:]                            # second step only
        flops = sum(r[2] for r in recs)
        # an event pair itself occupies the queue for a few microseconds: measure empty pairs the same way (GPU parked,
        # pairs back to back) and take that out of every launch interval
        torch.cuda._sleep(int(1.0e8))
        empty = []
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            empty.append((e0, e1))
        torch.cuda.synchronize()
        pair_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
        secs = sum(max(r[0].elapsed_time(r[1]) - pair_ms, 0.0) for r in recs) * 1e-3
        ach = flops / secs / 1e12
        if args.gemm_table:                                      # per-problem breakdown of the step's GEMM launches (stderr)
            by = {}
            for r in recs:
                c = by.setdefault(r[3], [0, 0.0, r[2]])
                c[0] += 1
                c[1] += r[0].elapsed_time(r[1]) * 1e3
            log("GEMM problems of one step (m, n, k, a_kmajor, b_kmajor, gather, split_k): launches, us each, ms total, TFLOP/s")
            for key, (cnt, us, fl) in sorted(by.items(), key=lambda kv: -kv[1][1]):
                log(f"  {str(key):44s} x{cnt:3d} {us / cnt:9.1f} us {us / 1e3:7.3f} ms {fl * cnt / us / 1e6:7.1f}")
        prec = hip.GEMM_PRECISION
        # peak for ALGORITHMIC flops: the dense bf16 MFMA rate divided by the MFMAs a mode issues per product (6 in the
        # reference-precision mode: 2500 / 6 = 416.7 TFLOP/s, against 157.3 on the fp32-input MFMA)
        terms = TERMS[prec]
        peak = PEAK_FP32_MFMA_TFLOPS if prec == 0 else PEAK_BF16_MFMA_TFLOPS / terms
        mfma_flops = flops * terms
        # HBM bytes per launch: PMC counters cannot be read from inside this process - the figure is the one of the
        # committed rocprofv3 --pmc passes of this same command (tools/final_profiles.sh), and the line says so
        traffic, traffic_src = None, None
        for name in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    pmc = json.load(f)
                if pmc.get("precision") == prec:
                    traffic = pmc["gemm"]["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/{name} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, gfx950 corrections applied; not measured in this run)"
                    break
            except (OSError, KeyError, ValueError):
                pass
        roof = {"kernel": KERNEL_NAMES[prec], "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": len(recs),
                "avg_launch_us": secs / max(len(recs), 1) * 1e6, "gemm_ms_per_step": secs * 1e3,
                "algorithmic_gflop_per_step": flops / 1e9, "mfma_issue_tflops": mfma_flops / secs / 1e12,
                "mfma_terms_per_product": terms, "peak_dense_bf16_mfma": PEAK_BF16_MFMA_TFLOPS,
                "frac_of_dense_bf16_issue": mfma_flops / secs / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                "event_pair_overhead_us": pair_ms * 1e3}

    # ---- the same step in the other arithmetic modes (one GPU only): mode 0 = fp32-input MFMA, mode 3 = three-term bf16
    # split (~2^-17 per product), mode 1 = plain bf16 MFMA inputs.  A fresh HIP graph per mode, 2 + 5 replays.
    other = {}
    if world == 1 and not args.no_other_modes and not args.no_graph:
        main_prec = hip.GEMM_PRECISION
        for mode, key in ((0, "dtype_f32_mfma"), (3, "dtype_bf16x3"), (1, "dtype_bf16")):
            if mode == main_prec:
                continue
            try:
                engine.drop_graph()
                hip.GEMM_PRECISION = mode
                engine.capture(batch, warmup=1)
                for _ in range(2):
                    engine.replay()
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(5):
                    o2 = engine.replay()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t) / 5 * 1e3
                other[key] = {"dtype": DTYPE_NAMES[mode].split(" ")[0], "precision_mode": mode, "ms_per_step": ms,
                              "value": args.batch * T / (ms * 1e-3), "steps": 5, "loss": float(o2[0])}
                log(f"mode {mode}: {ms:.2f} ms/step")
            except Exception as e:
                other[key] = {"precision_mode": mode, "error": f"{type(e).__name__}: {e}"}
            finally:
                hip.GEMM_PRECISION = main_prec
        engine.drop_graph()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline (oracle on host cores) ...")
        cpu = cpu_baseline(args.cpu_batch, args.seconds, args.target_len, args.cpu_steps)
        log(f"cpu baseline done: {cpu['value']:.0f} frames/s on {cpu['cores']} threads")

    dec = None
    if rank == 0 and world == 1 and not args.no_decode:
        log("decode RTF (attention rescoring, 64 utterances) ...")
        engine.arena.enabled = False
        torch.manual_seed(4)                                     # configs[3]: "weights = seeded init", not the model trained above
        dec_model = ASRModel(80, V, **MODEL_CONF).to(dev)
        from openeat_amd.models.language_model import LanguageModel
        lm = LanguageModel(V, encoder_num_blocks=6, d_model=256, attention_heads=4, linear_units=1024).to(dev)
        dec = decode_rtf(dec_model, fb, utt_normalize_, args.decode_utts, args.seconds, 10, dev, lm=lm, lm_weight=0.3)
        if not args.no_cpu_baseline:
            log("decode cpu baseline (oracle rescoring on host cores) ...")
            dec["cpu_baseline"] = decode_cpu_baseline(dec_model, lm, args.cpu_decode_utts, args.seconds, 10, 0.3)
            log(f"decode cpu baseline done: RTF {dec['cpu_baseline']['rtf']:.3f} on {dec['cpu_baseline']['cores']} threads")
        del dec_model, lm
        engine.arena.enabled = True
        log(f"decode done: RTF {dec['rtf']:.5f} ({dec['wall_s'] * 1e3:.0f} ms for {dec['utterances']} x {args.seconds:g} s)")

    if rank == 0:
        line = {"metric": METRIC, "value": value, "unit": "audio-frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": DTYPE_NAMES[hip.GEMM_PRECISION], "data": "synthetic",
                "config": {"workload": "configs[1]: 12L Conformer d=256 (12+3+3, h=4, ff=1024, K=15, V=3246), "
                                       f"B={args.batch}/GPU x {args.seconds:g} s 16 kHz wav (T={T} frames), L={args.target_len}, "
                                       "CTC+attention joint loss, fbank+fwd+bwd+clip+Adam, dropout 0.1",
                           "global_batch": world * args.batch, "parallelism": f"dp{world}", "hip_graph": use_graph,
                           **({"graph_form": graph_form} if use_graph and engine.reducer.active else {}),
                           **({"ddp_forced_with_one_rank": "OE_DDP_FORCE=1: RCCL process group, hooks and all-reduces (over a group of one) "
                                                           "inside the timed step"} if ddp.forced() else {})},
                **({"invalid": f"dropout overridden to {args.dropout}"} if args.dropout is not None else {}),
                "loss": loss, "roofline": roof, "cpu_baseline": cpu, **other, "decode": dec}
        real_stdout.write(json.dumps(line) + "\n")
        real_stdout.flush()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
