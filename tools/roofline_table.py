#!/usr/bin/env python3
"""Per-kernel-class roofline table for config 2 from the rocprofv3 kernel stats of
`bench.py --no-graph --single-stream` (profiles/<tag>_kernel_stats_p3.csv): algorithmic work per optimizer step (SURVEY 8d /
DESIGN section 4 formulas) / measured kernel time per step, against the MI355X peaks (8 TB/s HBM, 2.5 PFLOP/s dense bf16).

usage: roofline_table.py <kernel_stats.csv> <steps in trace> > profiles/<tag>_roofline_by_class.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
_adam = [int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"]]
if _adam:
    steps = float(_adam[0])                      # optimizer steps in the trace = launches of the Adam kernel
import json
import os
PREC = int(sys.argv[3]) if len(sys.argv) > 3 else 6
TERMS = {0: 1, 1: 1, 3: 3, 6: 6}[PREC]
_bench = json.load(open(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), os.path.basename(sys.argv[1]).split("_")[0] + f"_bench_p{PREC}.json")))
GEMM_FLOP = _bench["roofline"]["algorithmic_gflop_per_step"] * 1e9
GEMM_LAUNCHES = _bench["roofline"]["launches_per_step"]
ms = lambda pred: sum(int(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / 1e6 / steps
B, T, T1, F1, Tp, d, H, ff, K, V, L1, C = 32, 998, 498, 39, 248, 256, 4, 1024, 15, 3246, 31, 256
M = B * Tp                       # 7936 encoder rows
Md = B * L1                      # 992 decoder rows
f4 = 4
act = M * d * f4                 # one encoder activation: 8.1 MB
actd = Md * d * f4
y1 = B * T1 * F1 * C * f4        # conv1 output: 636 MB
attn_enc = 12 * (4 + 14) * B * H * Tp * Tp * (d // H)                    # fwd 4 + bwd 14 flops per (q, k, feature)
attn_dec = 6 * (4 + 14) * B * H * (L1 * L1 + L1 * Tp) * (d // H)         # self + source attention of 2 x 3 decoder layers
ln_enc, ln_dec = 12 * 6 + 1, 6 * 3 + 2                                   # 5 block norms + conv-module norm per layer, after_norm
# round 3, second session: a layer's norm_final + the norm behind it are ONE launch (11 pairs that write both outputs, one - in
# front of after_norm - that writes the second only), the conv module's norm rides in the depthwise-conv forward
# round 4: most encoder norms run as prologues / epilogues of the row-block GEMM kernels (their time is in the GEMM class): only the
# launches that still exist are counted here - by their call counts in the trace (decoder norms: ln_dec of them, 992 rows each)
calls = lambda pred: sum(int(r["Calls"]) for r in rows if pred(r["Name"])) / steps
n_fs, n_fp = calls(lambda n: "layernorm_fwd_kernel" in n and "false>" in n), calls(lambda n: "layernorm_fwd_kernel" in n and "true>" in n)
n_bs, n_bp = calls(lambda n: "layernorm_bwd_kernel" in n and "false>" in n), calls(lambda n: "layernorm_bwd_kernel" in n and "true>" in n)
ln_fwd_bytes = max(0.0, n_fs - ln_dec) * 2 * act + max(0.0, n_fp - 1) * 3 * act + min(n_fp, 1) * 2 * act + min(n_fs, ln_dec) * 2 * actd
ln_bwd_bytes = max(0.0, n_bs - ln_dec) * 4 * act + n_bp * 4 * act + min(n_bs, ln_dec) * 4 * actd
classes = [
    ("GEMM kernels (`gemm_pl_kernel`, `gemm_dma_kernel`, `gemm_bf16_kernel`, `gemm_tn_*`, `ffn6_kernel`, `rowgemm6*_kernel`, `rowtile6_kernel`)", "mfma", GEMM_FLOP, lambda n: ("gemm_" in n and "kernel" in n) or "ffn_fwd_kernel" in n or "ffn6_kernel" in n or ("rowgemm6" in n or "rowtile6" in n),   # gemm_tn_grouped_kernel included
     f"2*m*n*k of the step's {GEMM_LAUNCHES} launches (counted live by bench.py; conv2 forward / input / weight gradients included)"),
    ("attention (`attn_planes_q`, `attn_planes_k`; `attn_qtile`, `attn_ktile_bwd` for short axes)", "mfma", attn_enc + attn_dec, lambda n: n.startswith("void attn_") or n.startswith("attn_"),
     "(4 fwd + 14 bwd) * B*H*T1*T2*dk, encoder self-attention + decoder self/source attention"),
    ("CTC (`ctc_rows`, `ctc_alphabeta`, `ctc_labels`)", "hbm", 2 * M * V * f4 + 4 * M * (2 * 30 + 1) * f4, lambda n: "ctc_" in n and "greedy" not in n,
     "logits read once + gradient written once + alpha/beta"),
    ("LayerNorm forward (the launches that exist: norm pairs, decoder norms; the other encoder norms are prologues of GEMM-class kernels)", "hbm", ln_fwd_bytes, lambda n: "layernorm_fwd" in n, f"{n_fs:.0f} single + {n_fp:.0f} pair launches per step: read x + write y (a pair: + its first norm's output)"),
    ("LayerNorm backward (+ parameter reduce; the launches that exist)", "hbm", ln_bwd_bytes, lambda n: "layernorm_bwd" in n or "ln_param_reduce" in n,
     f"{n_bs:.0f} single + {n_bp:.0f} pair launches per step: read dy, x, residual gradient + write dx"),
    ("depthwise conv + GLU + its LayerNorm, forward", "hbm", 12 * 4 * act, lambda n: "dwconv_glu_fwd" in n, "read (B*T', 2d) + write (B*T', d) twice (conv output, normalised + activated)"),
    ("depthwise conv + GLU backward (+ reduce)", "hbm", 12 * 5 * act, lambda n: "dwconv_glu_bwd" in n or "dwconv_param_reduce" in n, "read a, dy + write da"),
    ("conv1 forward (`conv1_fwd`)", "hbm", y1 * 3 // 2 + B * T * 80 * f4, lambda n: "conv1_fwd" in n, "write the NHWC activation as three bf16 planes (6 bytes per element; no fp32 copy)"),
    ("conv1 weight gradient", "hbm", y1 + B * T * 80 * f4, lambda n: "conv1_wgrad" in n, "read dy1 + x"),
    ("fbank + per-utterance norm", "hbm", B * 160000 * f4 + 3 * B * T * 80 * f4, lambda n: "fbank_kernel" in n or "utt_norm" in n, "read wav, write/normalise features"),
    ("label-smoothing loss rows", "hbm", 2 * 2 * Md * V * f4, lambda n: "lsm_" in n, "logits read + gradient written, two decoders"),
    ("clip + Adam (`sumsq_partial`, `adam_kernel`)", "hbm", 31.3e6 * f4 * 8, lambda n: "adam_kernel" in n or "sumsq" in n, "g read twice; p, m, v read and written"),
]
peak = {"hbm": 8.0e12, "mfma": 2.5e15 / TERMS}      # algorithmic flops: the dense bf16 MFMA rate / MFMAs issued per product
unit = {"hbm": ("TB/s", 1e12), "mfma": ("TFLOP/s", 1e12)}
tot = sum(int(r["TotalDurationNs"]) for r in rows if "spin_kernel" not in r["Name"]) / 1e6 / steps
print(f"# Roofline by kernel class, config 2 (B=32 x 10 s), precision {PREC}\n")
print("Kernel time: rocprofv3 `--kernel-trace --stats` of `bench.py --no-graph --single-stream` (every kernel alone on one stream), "
      f"per optimizer step; all kernels {tot:.2f} ms/step.  Work: algorithmic bytes / flops (DESIGN section 4).  Peaks: HBM 8 TB/s, "
      f"dense bf16 MFMA 2.5 PFLOP/s / {TERMS} = {2500 / TERMS:.0f} TFLOP/s of algorithmic flops (precision {PREC} issues {TERMS} MFMAs per product; "
      "the attention class's short-axis (decoder) kernels run on the fp32-input MFMA in this mode and are priced against the same figure).\n")
print("| class | bound | algorithmic work / step | kernel ms / step | achieved | % of peak | work counted |\n|---|---|---|---|---|---|---|")
seen = 0.0
for name, bound, work, pred, note in classes:
    t = ms(pred)
    if t <= 0:
        continue
    seen += t
    u, s = unit[bound]
    w = f"{work / 1e9:.1f} GFLOP" if bound == "mfma" else f"{work / 1e6:.0f} MB"
    print(f"| {name} | {bound} | {w} | {t:.3f} | {work / (t * 1e-3) / s:.2f} {u} | {100 * work / (t * 1e-3) / peak[bound]:.1f} | {note} |")
print(f"\nThese classes cover {seen:.2f} of the {tot:.2f} ms of kernel time per step; the rest is glue (dropout of gradients, relative-position "
      "prepare/backward, embeddings, layout swaps, token bookkeeping).")
