"""Kaldi-compatible log-mel filterbank + CMVN (oracle; test-only).

PARITY UNPINNED.  The reference calls ``torchaudio.compliance.kaldi.fbank``
(``/root/reference/openeat/dataset/dataset.py:93-100``: num_mel_bins,
frame_length=25, frame_shift=10, dither, energy_floor=0.0, sample_frequency;
every other argument at torchaudio's default).  torchaudio is a third-party
dependency that is neither vendored in the reference nor installed in this
image, and the reference pins no version (only a conda env name
``torch1.9_cuda11.1`` => torchaudio 0.9.x, ``examples/aishell/run.sh:10``).
This file restates the published Kaldi algorithm with those parameters:

  wav * 2^15 (dataset.py:75) -> frames of 25 ms / hop 10 ms, snip_edges ->
  [dither] -> subtract frame mean -> pre-emphasis 0.97 (first sample uses
  itself as predecessor) -> povey window (symmetric hann ** 0.85) -> zero-pad
  to 512 -> |rFFT|^2 -> 80 triangular mel filters (20 Hz .. Nyquist, mel =
  1127 ln(1 + f/700), built on the first N/2 bins; the Nyquist bin gets weight
  0) -> log(max(., FLT_EPSILON)).

The tests cross-check it against ``transformers.audio_utils`` (an independent
Kaldi-compatible implementation present in this image); that is a sanity
anchor, not a pin against torchaudio.

Per-utterance normalisation follows
``/root/reference/openeat/dataset/feature_processor.py:5-8`` (population std,
no epsilon) and global CMVN ``/root/reference/openeat/modules/cmvn.py:35-46``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

FLT_EPS = 1.1920928955078125e-07


def _next_pow2(n: int) -> int:
    return 1 if n == 0 else 2 ** (n - 1).bit_length()


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def povey_window(n: int) -> torch.Tensor:
    return torch.hann_window(n, periodic=False, dtype=torch.float32).pow(0.85)


def mel_banks(num_bins: int, padded: int, sample_rate: float, low: float = 20.0, high: float = 0.0) -> torch.Tensor:
    """(num_bins, padded//2 + 1) float32; last column zero."""
    nfft_bins = padded // 2
    nyq = 0.5 * sample_rate
    if high <= 0.0:
        high += nyq
    bin_w = sample_rate / padded
    mlo, mhi = float(mel_scale(low)), float(mel_scale(high))
    delta = (mhi - mlo) / (num_bins + 1)
    b = torch.arange(num_bins, dtype=torch.float32).unsqueeze(1)
    left = mlo + b * delta
    center = mlo + (b + 1.0) * delta
    right = mlo + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + bin_w * torch.arange(nfft_bins, dtype=torch.float32) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(w, (0, 1), value=0.0)


def num_frames(n_samples: int, win: int = 400, hop: int = 160) -> int:
    return 0 if n_samples < win else 1 + (n_samples - win) // hop


def fbank(wav: torch.Tensor, num_mel_bins: int = 80, sample_rate: float = 16000.0,
          frame_length_ms: float = 25.0, frame_shift_ms: float = 10.0, dither: float = 0.0,
          preemph: float = 0.97, scale: float = 32768.0) -> torch.Tensor:
    """wav (N,) float in [-1, 1) -> (T, num_mel_bins) float32."""
    x = wav.to(torch.float32) * scale
    hop = int(sample_rate * frame_shift_ms * 0.001)
    win = int(sample_rate * frame_length_ms * 0.001)
    padded = _next_pow2(win)
    T = num_frames(x.numel(), win, hop)
    if T == 0:
        return torch.empty(0, num_mel_bins)
    fr = x.as_strided((T, win), (hop, 1)).clone()
    if dither != 0.0:
        fr = fr + torch.randn_like(fr) * dither
    fr = fr - fr.mean(dim=1, keepdim=True)
    prev = torch.cat([fr[:, :1], fr[:, :-1]], dim=1)
    fr = fr - preemph * prev
    fr = fr * povey_window(win).unsqueeze(0)
    fr = torch.nn.functional.pad(fr, (0, padded - win))
    power = torch.fft.rfft(fr).abs().pow(2.0)
    mel = power @ mel_banks(num_mel_bins, padded, sample_rate).T
    return torch.max(mel, torch.tensor(FLT_EPS)).log()


def utt_normalize(feat: torch.Tensor) -> torch.Tensor:
    """feature_processor.py:5-8 on a (T, F) matrix (numpy mean/std, ddof=0)."""
    a = feat.numpy()
    return torch.from_numpy((a - np.mean(a, axis=0)) / np.std(a, axis=0))


def global_cmvn(x: torch.Tensor, mean: torch.Tensor, istd: torch.Tensor) -> torch.Tensor:
    return (x - mean) * istd
