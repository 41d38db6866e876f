#!/usr/bin/env python3
"""Which python lines of the training step launch aten kernels (copies, fills, adds - the glue between the HIP kernels)?
One eager step at the bench shape under torch.profiler with stacks; prints aten ops that ran a device kernel, grouped by
the innermost openeat_amd frame.  (GPU box.)"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openeat_amd import hip  # noqa: E402
from openeat_amd.engine import TrainEngine  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

hip.GEMM_PRECISION = 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).train()
eng = TrainEngine(model, lr=1e-3, grad_clip=5.0, static_shapes=True, parallel_decoders=False, async_wgrad=False)
B, T, L = 32, 998, 30
batch = dict(features=torch.randn(B, T, 80, device=dev), features_length=torch.full((B,), T, dtype=torch.int32, device=dev),
             targets=torch.randint(2, bench.V - 1, (B, L), dtype=torch.int32, device=dev),
             targets_length=torch.full((B,), L, dtype=torch.int32, device=dev))
for _ in range(2):
    eng.step(batch)
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SKIP = ("aten::view", "aten::_unsafe_view", "aten::as_strided", "aten::detach", "aten::alias", "aten::reshape", "aten::t", "aten::transpose",
        "aten::select", "aten::slice", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::permute", "aten::empty", "aten::empty_like",
        "aten::empty_strided", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::is_same_size", "aten::unbind", "aten::split",
        "aten::new_empty", "aten::narrow", "aten::_reshape_alias", "aten::stride", "aten::size", "aten::sym_size", "aten::unfold")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name not in SKIP:
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            if t is not None and t.is_cuda:
                fr = [f for f in traceback.extract_stack() if "openeat_amd" in f.filename or f.filename.endswith("bench.py")]
                where = f"{os.path.relpath(fr[-1].filename, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}:{fr[-1].lineno} {fr[-1].line}" if fr else "(autograd engine)"
                k = (name, where[:150])
                self.by[k][0] += 1
                self.by[k][1] += t.numel()
        return out


spy = Spy()
with spy:
    eng.step(batch)
torch.cuda.synchronize()
print("aten calls on CUDA tensors in one eager step (dispatch level; count, total elements of the result):")
for (name, where), (n, el) in sorted(spy.by.items(), key=lambda kv: -kv[1][1]):
    print(f"x{n:4d} {el:12d} el  {name:26s} {where}")
