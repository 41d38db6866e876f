#!/usr/bin/env python3
"""Un-profiled timeline of the captured training step (config 2): wall-clock stamps (oe_stamp, one-thread kernels) captured
into the graph at phase boundaries of the main stream - forward per three encoder layers, heads, backward per three layers,
the tail - read back after replays.  A kernel trace serialises what the graph overlaps; these stamps do not.  (GPU box.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openeat_amd import hip, ops  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.frontend import Fbank, utt_normalize_  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

hip.GEMM_PRECISION = int(os.environ.get("OE_GEMM_PRECISION", "3"))
dev = torch.device("cuda:0")
torch.manual_seed(777)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
eng = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=True)
fb = Fbank(80, device=dev)
wav, tgt, tlen = bench.synth_batch(32, 10.0, 30, seed=0, device=dev)
T = fb.num_frames(wav.shape[1])
feats = torch.empty(32, T, 80, device=dev)
flen = torch.full((32,), T, dtype=torch.int32, device=dev)


class WithFrontend(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, wav, targets, targets_length):
        fb(wav, out=feats)
        utt_normalize_(feats, flen)
        ops.stamp("fwd: fbank + normalisation done")
        return self.m(feats, flen, targets, targets_length)


eng.model = WithFrontend(model)
batch = {"wav": wav, "targets": tgt, "targets_length": tlen}
eng.step(batch)
torch.cuda.synchronize()
ops.STAMPS = {"buf": torch.zeros(64, dtype=torch.int64, device=dev), "tags": []}
eng.capture(batch, warmup=1)
tags = list(ops.STAMPS["tags"])
buf = ops.STAMPS["buf"]
first = len(tags) - 1 - tags[::-1].index("step starts")          # capture()'s eager warm-up step recorded a copy first
ops.STAMPS = None                       # nothing more is recorded; the graph keeps its stamp launches
runs = []
for _ in range(12):
    eng.replay()
    torch.cuda.synchronize()
    runs.append(buf[first: len(tags)].cpu().clone())
runs = torch.stack(runs[2:]).double()
t = (runs - runs[:, :1]) / 100.0        # 100 MHz ticks -> microseconds since the step's first stamp
med = t.median(0).values
tags = tags[first:]
order = sorted(range(len(tags)), key=lambda i: float(med[i]))
prev = 0.0
print(f"captured step, {len(tags)} stamps, median of {runs.shape[0]} replays (microseconds since the step's start; delta to the previous stamp)")
for i in order:
    print(f"  {med[i]:9.1f}  +{med[i] - prev:8.1f}   {tags[i]}")
    prev = float(med[i])
