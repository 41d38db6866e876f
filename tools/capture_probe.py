"""Round 3's red GPU test, made deterministic (run once on the GPU box; output -> profiles/r04_capture_probe.txt).

  python tools/capture_probe.py

For ISOLATE_CAPTURES in (False, True): eager decode (registers planes), graph call (eager + capture), then every planes
buffer eager code owns is released and the freed blocks are filled with 0xFF (tests/test_gpu_capture_isolation.py), then the
REPLAY of another batch is compared with its eager result.  Also counted: how many operands the captured launches took from
planes that were made OUTSIDE the capture (registry / weight-cache hits while the stream was capturing)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from openeat_amd import hip, planes  # noqa: E402
from test_gpu_capture_isolation import _case, _model, release_and_poison  # noqa: E402


def run(isolate: bool):
    hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY = 6, 0, "all"
    hip.lib().oe_gemm_pl_config(0, -1, -1, -1)
    planes.ISOLATE_CAPTURES = isolate
    planes.clear_all()
    foreign = {"registry": 0, "weights": 0}
    lookup0, weight0 = planes.lookup, planes.weight

    def lookup(t):
        pl = lookup0(t)
        if pl is not None and torch.cuda.is_current_stream_capturing():
            alloc_in_capture = getattr(pl.t, "_oe_in_capture", False)
            foreign["registry"] += 0 if alloc_in_capture else 1
        return pl

    alloc0 = planes.alloc

    def alloc(rows, cols, device):
        pl = alloc0(rows, cols, device)
        pl.t._oe_in_capture = torch.cuda.is_current_stream_capturing()
        return pl

    def weight(w):
        pl = weight0(w)
        if pl is not None and torch.cuda.is_current_stream_capturing() and not getattr(pl.t, "_oe_in_capture", False):
            foreign["weights"] += 1
        return pl

    planes.lookup, planes.alloc, planes.weight = lookup, alloc, weight
    try:
        model = _model()
        a, b = _case(41, [97, 83, 64, 41, 23]), _case(42, [97, 90, 97, 60, 97])
        kw = dict(ctc_weight=0.5, reverse_weight=0.3)
        with torch.no_grad():
            want_a = model.attention_rescoring_batch(*a, 4, use_graphs=False, **kw)
            want_b = model.attention_rescoring_batch(*b, 4, use_graphs=False, **kw)
            got_a = model.attention_rescoring_batch(*a, 4, use_graphs=True, **kw)
            poison = release_and_poison(planes)
            got_b = model.attention_rescoring_batch(*b, 4, use_graphs=True, **kw)
        print(f"ISOLATE_CAPTURES={isolate}: operands a capture took from planes made outside it: registry {foreign['registry']}, "
              f"cached weight splits {foreign['weights']}; poison blocks {len(poison)}")
        print(f"  first call (eager + capture) equals eager: {got_a == want_a}")
        print(f"  REPLAY equals eager: {got_b == want_b}")
        print(f"    eager : {[len(h) for h in want_b]} tokens per utterance")
        print(f"    replay: {[len(h) for h in got_b]} tokens per utterance")
    finally:
        planes.lookup, planes.alloc, planes.weight = lookup0, alloc0, weight0
        planes.ISOLATE_CAPTURES = True
        planes.clear_all()


if __name__ == "__main__":
    run(False)
    run(True)
