"""Waveform-level augmentation: the speed draw and speed perturbation of
/root/reference/openeat/dataset/audio_processor.py:5-35, the latter on the device for a whole padded batch."""
import math
import random
from typing import Optional, Sequence

import torch

from openeat_amd import hip


def _speed_generator(speeds=None):
    """audio_processor.py:5-19 (python `random`, same call order: one randrange or one random() per call).
    With the default [0.9, 1.1, 0.1] the reference's randrange(9, 10) always yields 9, i.e. speed 0.9 (kept)."""
    if speeds is None:
        speeds = [0.9, 1.1, 0.1]
    speeds = [float(s) for s in speeds]
    if len(speeds) > 1:
        assert speeds[1] > speeds[0], 'speeds is wrong !'
        if speeds[2] != 0:
            lo = int(speeds[0] / speeds[2])
            speed = random.randrange(lo, lo + 1)
            speed *= speeds[2]
        else:
            speed = speeds[0] + random.random() * (speeds[1] - speeds[0])
    else:
        speed = speeds[0]
    return speed


def perturbed_length(n: int, speed: float) -> int:
    """Samples after `speed`: floor(n / speed + 0.5)."""
    return int(math.floor(n / speed + 0.5))


def speed_perturb_batch(wav: torch.Tensor, nsamples: Sequence[int], speeds: Sequence[float], out: Optional[torch.Tensor] = None):
    """audio_processor.py:20-35 for a padded batch on the device: utterance b is read speeds[b] times faster and
    resampled to the same rate (oe_speed_perturb: windowed sinc; sox is not available - distribution parity).
    wav (B, N) float32 CUDA; returns (out (B, Nmax_out), nsamples_out list)."""
    if wav.device.type != "cuda":
        raise TypeError("speed_perturb_batch: CUDA tensor required (openeat_amd has no CPU fallback)")
    assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.stride(1) == 1
    B = wav.shape[0]
    assert len(nsamples) == B and len(speeds) == B
    n_out = [perturbed_length(int(n), float(s)) if float(s) != 1.0 else int(n) for n, s in zip(nsamples, speeds)]
    nmax = max(max(n_out), 1)
    if out is None:
        out = torch.empty(B, nmax, dtype=torch.float32, device=wav.device)
    assert out.shape[0] == B and out.shape[1] >= nmax and out.stride(1) == 1
    dev = wav.device
    ni = torch.tensor([int(n) for n in nsamples], dtype=torch.int32).to(dev)
    no = torch.tensor(n_out, dtype=torch.int32).to(dev)
    sp = torch.tensor([float(s) for s in speeds], dtype=torch.float32).to(dev)
    ld_in = wav.stride(0) if B > 1 else max(wav.shape[1], 1)      # a lone row may carry any (even zero) batch stride
    hip.call("oe_speed_perturb", wav, ld_in, ni, sp, B, out.shape[1], out, out.stride(0), no)
    return out, n_out


def _speed_perturb(waveform: torch.Tensor, sample_rate: int, speed: Optional[float] = None) -> torch.Tensor:
    """The reference's per-utterance signature (audio_processor.py:20): waveform (1, N) -> (1, N')."""
    if speed is None or speed == 1.0:
        return waveform
    out, n = speed_perturb_batch(waveform.reshape(1, -1).contiguous(), [waveform.numel()], [speed])
    return out[:, :n[0]]
