#!/usr/bin/env python3
"""Batched attention-rescoring decode (64 x 10 s, beam 10) eagerly and from the per-shape HIP graphs, per stage (GPU box).
LM=1 adds the 6-layer Transformer LM.  Short hypotheses (a trained model) make decode launch-bound and the graphs halve it;
the bench's untrained model emits 210-token hypotheses and is GEMM-bound either way."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from openeat_amd import hip
from openeat_amd.frontend import Fbank, utt_normalize_
from openeat_amd.models.asr_model import ASRModel
import openeat_amd.models.asr_model as M
hip.GEMM_PRECISION = 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ASRModel(80, bench.V, **bench.MODEL_CONF).to(dev).eval()
fb = Fbank(80, device=dev)
from openeat_amd.models.language_model import LanguageModel
lm = LanguageModel(bench.V, encoder_num_blocks=6, d_model=256, attention_heads=4, linear_units=1024).to(dev).eval() if os.environ.get("LM") else None
wav = (torch.rand(64, 160000) - 0.5).to(dev)
feats, nfr = fb(wav); utt_normalize_(feats, nfr)
orig = M._graph_call
def spy(cache, key, fn, args):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = orig(cache, key, fn, args)
    torch.cuda.synchronize()
    print(f"   {key[0]} {'captured' if cache.get(key) else cache.get(key)}: {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
    return out
M._graph_call = spy
with torch.no_grad():
    for g in (False, True, True, True):
        torch.cuda.synchronize(); t = time.perf_counter()
        h = model.attention_rescoring_batch(feats, nfr, 10, ctc_weight=0.5, reverse_weight=0.3, use_graphs=g, lm=lm, lm_weight=0.3)
        torch.cuda.synchronize()
        print(f"graphs={g}: {(time.perf_counter()-t)*1e3:.1f} ms", flush=True)
