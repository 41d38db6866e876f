#!/usr/bin/env python3
"""Where the attention-rescoring decode of bench.py (64 x 10 s, beam 10) spends its wall time (GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from openeat_amd import hip, ops  # noqa: E402
from openeat_amd.frontend import Fbank, utt_normalize_  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).eval()
fb = Fbank(80, device=dev)
wav = (torch.rand(64, 160000) - 0.5).to(dev)


def T(label, f, acc):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f()
    torch.cuda.synchronize()
    acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t) * 1e3
    return r


for it in range(3):
    acc = {}
    with torch.no_grad():
        feats, nfr = T("fbank+norm", lambda: (lambda fn: (utt_normalize_(fn[0], fn[1]), fn)[1])(fb(wav)), acc)
        enc, mask, _ = T("encoder", lambda: model._encode(feats, nfr), acc)
        tp, ti = T("ctc projection + fused log-softmax top-k", lambda: ops.topk_rows(model.ctc.logits(enc), 10, log_softmax=True), acc)
        lens_dev = mask.squeeze(1).sum(1).to(torch.int32)
        nb_dev = T("device prefix beam x64 (one wave each, incl. D2H of the n-best)", lambda: hip.ctc_prefix_beam_device(tp, ti, lens_dev, 10), acc)
        tpc, tic, lens = T("d2h", lambda: (tp.cpu(), ti.cpu(), mask.squeeze(1).sum(1).cpu().tolist()), acc)
        nb = T("host prefix beam x64 (one by one)", lambda: [hip.ctc_prefix_beam_host(tpc[b, : lens[b]], tic[b, : lens[b]], 10) for b in range(64)], acc)
        nb2 = T("host prefix beam x64 (batch, threads)", lambda: hip.ctc_prefix_beam_host_batch(tpc, tic, lens, 10), acc)
        assert nb == nb2
        assert [[p for p, _ in u] for u in nb_dev] == [[p for p, _ in u] for u in nb2]
        mean_len = sum(len(p) for u in nb for p, _ in u) / 640.0
        T("whole attention_rescoring_batch", lambda: model.attention_rescoring_batch(feats, nfr, 10, ctc_weight=0.5, reverse_weight=0.3), acc)
    if it == 2:
        for k, v in acc.items():
            print(f"{k:72s} {v:8.2f} ms")
        print(f"{'mean n-best length (untrained weights: hypotheses of nearly every frame)':72s} {mean_len:8.2f} tokens")
