"""Feature-level augmentation on the device: SpecAugment and spec-substitute of
/root/reference/openeat/dataset/feature_processor.py:10-64 (applied by CollateFunc, dataset.py:203-209, after the
per-utterance normalisation).

The reference draws its parameters with python's ``random`` per utterance; the draws are reproduced here on the host
in exactly the same order (so ``random.seed(s)`` gives the same masks as the reference), and the data movement happens
on the GPU on the whole padded batch at once.
"""
import random
from typing import Sequence

import torch

from openeat_amd import hip


def feature_dither_(feats: torch.Tensor, nframes: Sequence[int], feature_dither: float) -> torch.Tensor:
    """dataset.py:197-201: a = random.uniform(0, feature_dither) once per batch, then x += (U[0,1) - 0.5) * a.
    `a` is drawn with python's random exactly as the reference does; the per-element uniforms come from the device
    generator (the reference's come from numpy's global state), so this is distribution parity."""
    if feature_dither == 0.0:
        return feats
    a = random.uniform(0, feature_dither)
    B, Tmax, F = feats.shape
    nf = torch.as_tensor([int(n) for n in nframes], dtype=torch.int32).to(feats.device)
    hip.call("oe_feature_dither", feats, nf, B, Tmax, F, float(a), random.getrandbits(63))
    return feats


def spec_substitute_(feats: torch.Tensor, nframes: Sequence[int], max_t: int = 20, num_t_sub: int = 3) -> torch.Tensor:
    """feature_processor.py:45-64, in place on (B, Tmax, F): per utterance num_t_sub times
    start = randint(0, T-1); length = randint(1, max_t); end = min(T, start+length); pos = randint(0, start);
    y[start:end] = y[start-pos:end-pos]."""
    B, Tmax, F = feats.shape
    subs = []
    for b in range(B):
        T = int(nframes[b])
        for _ in range(num_t_sub):
            start = random.randint(0, T - 1)
            length = random.randint(1, max_t)
            end = min(T, start + length)
            pos = random.randint(0, start)
            subs.append((start, end, pos))
    if not subs:
        return feats
    sd = torch.tensor(subs, dtype=torch.int32).view(B, num_t_sub, 3).to(feats.device)
    hip.call("oe_spec_substitute", feats, B, Tmax, F, sd, num_t_sub, max(int(max_t), 1))
    return feats


def spec_augment_(feats: torch.Tensor, nframes: Sequence[int], num_t_mask: int = 2, num_f_mask: int = 2, max_t: int = 50,
                  max_f: int = 10) -> torch.Tensor:
    """feature_processor.py:10-43, in place on (B, Tmax, F): per utterance first the time masks
    (start = randint(0, T-1); length = randint(1, max_t)), then the frequency masks (same with F, max_f); value 0."""
    B, Tmax, F = feats.shape
    tm, fm = [], []
    for b in range(B):
        T = int(nframes[b])
        for _ in range(num_t_mask):
            start = random.randint(0, T - 1)
            length = random.randint(1, max_t)
            tm.append((start, min(T, start + length)))
        for _ in range(num_f_mask):
            start = random.randint(0, F - 1)
            length = random.randint(1, max_f)
            fm.append((start, min(F, start + length)))
    dev = feats.device
    nf = torch.as_tensor([int(n) for n in nframes], dtype=torch.int32).to(dev)
    tmd = torch.tensor(tm, dtype=torch.int32).view(B, num_t_mask, 2).to(dev) if tm else None
    fmd = torch.tensor(fm, dtype=torch.int32).view(B, num_f_mask, 2).to(dev) if fm else None
    hip.call("oe_spec_augment", feats, nf, B, Tmax, F, tmd, num_t_mask, fmd, num_f_mask, 0.0)
    return feats
