#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): time the reference's own training step against the oracle's on the same
host cores, same parameters, same batch - BASELINE.md section 3 asks the restatement that stands in for the reference
on the GPU box (bench.py's cpu_baseline, kind "port") to run within ~+-20 % of the reference's CPU speed here.

    python tools/ref_vs_oracle_cpu.py [--batch 8] [--steps 3] > profiles/r02_ref_vs_oracle_cpu.json

Both arms: 12-layer Conformer (configs[1]), B utterances x 998 frames x 80 mel, 30 tokens each, train mode with dropout
0.1, forward + backward + clip_grad_norm_(5.0) + Adam; arms interleaved step by step; median of the timed steps.
The reference is imported as tests/golden/make_fixtures.py does (typeguard stubbed in memory); nothing of it is stored.
"""
import argparse
import json
import os
import statistics
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    assert os.path.isdir(REF), "this script only runs where /root/reference exists"
    tg = types.ModuleType("typeguard")
    tg.check_argument_types = lambda: True
    sys.modules["typeguard"] = tg
    # the reference's `openeat` has no __init__.py (a namespace package): the repo's alias package of the same name would
    # win the import, so the reference is imported first, with the repo root (and this script's directory) off sys.path
    sys.path[:] = [REF] + [p for p in sys.path if os.path.abspath(p or ".") not in (ROOT, os.path.join(ROOT, "tools"))]
    from openeat.models.asr_model import ASRModel as RefModel      # the reference itself
    assert RefModel.__module__ == "openeat.models.asr_model" and sys.modules[RefModel.__module__].__file__.startswith(REF)
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "openeat" or k.startswith("openeat.")]:
        sys.modules["_ref_" + k] = sys.modules.pop(k)                 # keep the repo's alias package importable afterwards
    sys.path.insert(0, ROOT)
    from bench import MODEL_CONF, V
    from oracle import asr as O

    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    torch.manual_seed(777)
    ref = RefModel(80, V, **MODEL_CONF).train()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in ref.state_dict().items()}
    cfg = O.Config(input_size=80, vocab_size=V, **MODEL_CONF)
    o_params = [v for v in sd.values() if v.requires_grad]
    r_opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    o_opt = torch.optim.Adam(o_params, lr=1e-3)
    g = torch.Generator().manual_seed(0)
    B = args.batch
    feats = torch.randn(B, 998, 80, generator=g)
    flen = torch.full((B,), 998, dtype=torch.int32)
    tgt = torch.randint(2, V - 1, (B, 30), generator=g, dtype=torch.int32)
    tlen = torch.full((B,), 30, dtype=torch.int32)

    def ref_step():
        r_opt.zero_grad()
        loss, _ = ref(feats, flen, tgt, tlen)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
        r_opt.step()
        return float(loss)

    def oracle_step():
        o_opt.zero_grad()
        loss, _ = O.forward(sd, cfg, feats, flen, tgt, tlen, training=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(o_params, 5.0)
        o_opt.step()
        return float(loss)

    t_ref, t_orc = [], []
    for it in range(args.warmup + args.steps):
        t0 = time.perf_counter(); lr_ = ref_step(); t1 = time.perf_counter(); lo_ = oracle_step(); t2 = time.perf_counter()
        print(f"step {it}: reference {t1 - t0:.2f} s (loss {lr_:.3f}), oracle {t2 - t1:.2f} s (loss {lo_:.3f})", file=sys.stderr, flush=True)
        if it >= args.warmup:
            t_ref.append(t1 - t0)
            t_orc.append(t2 - t1)
    mr, mo = statistics.median(t_ref), statistics.median(t_orc)
    print(json.dumps({"what": "reference (/root/reference, imported) vs oracle/ training step on the same host cores",
                      "model": "configs[1] 12L Conformer d=256 12+3+3 V=3246, dropout 0.1, fwd+bwd+clip+Adam",
                      "batch": f"{B} x 998 frames x 80 mel, 30 tokens", "cores": cores, "timed_steps": args.steps, "warmup": args.warmup,
                      "reference_s_per_step": mr, "oracle_s_per_step": mo, "reference_frames_per_s": B * 998 / mr,
                      "oracle_frames_per_s": B * 998 / mo, "oracle_over_reference_speed": mr / mo,
                      "within_20_percent": bool(0.8 <= mr / mo <= 1.25)}, indent=1))


if __name__ == "__main__":
    main()
