#!/usr/bin/env python3
"""Where in backward do the groups of deferred weight gradients get flushed to the side stream?  One eager config-2 step with
ops.WGRAD_DEFER = 48: every flush with its size and the autograd Function whose backward was running.  (GPU box.)"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from openeat_amd import hip, ops  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
torch.manual_seed(777)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
engine = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, async_wgrad=True, parallel_decoders=True)
feats = torch.randn(32, 998, 80, device=dev)
flen = torch.full((32,), 998, dtype=torch.int32, device=dev)
_, tgt, tlen = bench.synth_batch(32, 10.0, 30, seed=0, device=dev)
batch = {"features": feats, "features_length": flen, "targets": tgt, "targets_length": tlen}
engine.step(batch)
ops.WGRAD_DEFER = 48
orig = ops.flush_wgrads
seen = [0]


def flush():
    n = len(ops._deferred)
    if n:
        who = "?"
        for fr in reversed(traceback.extract_stack()[:-1]):
            if fr.name == "backward" or fr.name in ("join_side_stream", "_fwd_bwd_body", "step"):
                who = f"{fr.name} ({os.path.basename(fr.filename)}:{fr.lineno})"
                break
        names = []
        for fn, ts, st, d in ops._deferred:
            if d is not None:
                names.append(tuple(d["out"].shape))
        seen[0] += n
        print(f"flush of {n:3d} (total so far {seen[0]:3d}) from {who}; last outputs {names[-3:]}")
    return orig()


ops.flush_wgrads = flush
engine.step(batch)
torch.cuda.synchronize()
print("deferred weight gradients in the step:", seen[0])
