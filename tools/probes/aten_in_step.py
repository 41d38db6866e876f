#!/usr/bin/env python3
"""Which aten ops still launch kernels inside the training step, and from where?  One eager config-2 step under torch.profiler with
stacks: every aten op that launched a device kernel / memcpy, grouped by op and innermost repo frame.  (GPU box.)"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from openeat_amd import hip  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.frontend import Fbank, utt_normalize_  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
torch.manual_seed(777)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
engine = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, async_wgrad=True, parallel_decoders=True)
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
        return self.m(feats, flen, targets, targets_length)


engine.model = WithFrontend(model)
batch = {"wav": wav, "targets": tgt, "targets_length": tlen}
for _ in range(3):
    engine.step(batch)
torch.cuda.synchronize()
# host-side census of the copies that go to hipMemcpyAsync (same dtype, both sides dense): call sites
import traceback  # noqa: E402
sites = collections.Counter()
_copy, _clone = torch.Tensor.copy_, torch.Tensor.clone


def _site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "openeat_amd" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
    return "?"


def copy_(self, src, *a, **k):
    if self.is_cuda and isinstance(src, torch.Tensor) and src.is_cuda and self.dtype == src.dtype and self.is_contiguous() and src.is_contiguous():
        sites[("copy_", _site(), self.numel() * self.element_size())] += 1
    return _copy(self, src, *a, **k)


def clone(self, *a, **k):
    if self.is_cuda and self.is_contiguous():
        sites[("clone", _site(), self.numel() * self.element_size())] += 1
    return _clone(self, *a, **k)


torch.Tensor.copy_, torch.Tensor.clone = copy_, clone
engine.step(batch)
torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.clone = _copy, _clone
print(f"{sum(sites.values())} dense device-to-device copies requested from Python in one step")
for k, n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d} x {k[0]:6s} {k[2]:10d} B  {k[1]}")
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    engine.step(batch)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
groups = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
        continue
    if any(c.name.startswith("aten::") and c.kernels for c in ev.cpu_children):
        continue                               # count the innermost aten op that owns the kernels
    site = "?"
    for fr in ev.stack or []:
        if "openeat_amd" in fr or "bench.py" in fr:
            site = fr.replace(root + "/", "")
            break
    groups[(ev.name, site)] += len(ev.kernels)
    dur[(ev.name, site)] += sum(k.duration for k in ev.kernels)
tot = sum(groups.values())
print(f"{tot} device launches from aten ops in one step, {sum(dur.values()):.0f} us")
for k, n in sorted(groups.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{n:4d} launches {dur[k]:8.1f} us  {k[0]:28s} {k[1][:130]}")
