#!/usr/bin/env python3
"""fbank + per-utterance normalisation at config-2 size in a loop (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openeat_amd.frontend import Fbank, utt_normalize_  # noqa: E402

fb = Fbank(80, device="cuda")
wav = torch.rand(32, 160000, device="cuda") - 0.5
for _ in range(20):
    feats, nfr = fb(wav)
    utt_normalize_(feats, nfr)
torch.cuda.synchronize()
