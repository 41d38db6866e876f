"""On-device acoustic front end: kaldi fbank (+ global CMVN) and per-utterance
normalisation - the part of the reference's CPU data plane that the hot path
moves onto the GPU (/root/reference/openeat/dataset/dataset.py:75,93-100;
feature_processor.py:5-8)."""
import math

import numpy as np
import torch

from openeat_amd import hip

FLT_EPS = 1.1920928955078125e-07


class Fbank:
    """fbank(wav (B,N) in [-1,1), nsamples) -> (features (B,T,n_mel), nframes (B))."""

    def __init__(self, num_mel_bins: int = 80, sample_rate: float = 16000.0, frame_length_ms: float = 25.0,
                 frame_shift_ms: float = 10.0, low_freq: float = 20.0, high_freq: float = 0.0, preemph: float = 0.97,
                 scale: float = 32768.0, device="cuda"):
        self.n_mel = num_mel_bins
        self.win = int(sample_rate * frame_length_ms * 0.001)
        self.hop = int(sample_rate * frame_shift_ms * 0.001)
        self.nfft = max(128, 1 << (self.win - 1).bit_length())    # kaldi: round_to_power_of_two (the kernel's smallest FFT is 128)
        if self.nfft > 1024:
            raise NotImplementedError("the fbank kernel takes windows of up to 1024 samples (64 ms @ 16 kHz)")
        self.preemph, self.scale = preemph, scale
        # tables, built once on the host with the kaldi/torchaudio float32 formulas
        window = torch.hann_window(self.win, periodic=False, dtype=torch.float32).pow(0.85)
        k = np.arange(self.nfft // 2, dtype=np.float64)
        tw = np.stack([np.cos(2 * np.pi * k / self.nfft), -np.sin(2 * np.pi * k / self.nfft)], 1).astype(np.float32)
        nyq = 0.5 * sample_rate
        hi = high_freq + nyq if high_freq <= 0 else high_freq
        mlo, mhi = 1127.0 * math.log(1.0 + low_freq / 700.0), 1127.0 * math.log(1.0 + hi / 700.0)
        delta = (mhi - mlo) / (num_mel_bins + 1)
        bins = torch.arange(num_mel_bins, dtype=torch.float32).unsqueeze(1)
        left, center, right = mlo + bins * delta, mlo + (bins + 1.0) * delta, mlo + (bins + 2.0) * delta
        mel = (1127.0 * (1.0 + (sample_rate / self.nfft) * torch.arange(self.nfft // 2, dtype=torch.float32) / 700.0).log()).unsqueeze(0)
        w = torch.max(torch.zeros(1), torch.min((mel - left) / (center - left), (right - mel) / (right - center)))
        starts, offs, vals = [], [0], []
        for m in range(num_mel_bins):
            nz = torch.nonzero(w[m] > 0).flatten()
            lo, hi_ = (int(nz[0]), int(nz[-1]) + 1) if nz.numel() else (0, 0)
            starts.append(lo)
            vals.append(w[m, lo:hi_])
            offs.append(offs[-1] + hi_ - lo)
        self.window = window.to(device)
        self.twiddle = torch.from_numpy(tw).to(device).contiguous()
        self.mel_start = torch.tensor(starts, dtype=torch.int32, device=device)
        self.mel_off = torch.tensor(offs, dtype=torch.int32, device=device)
        self.mel_w = torch.cat(vals).to(device).contiguous()

    def num_frames(self, n: int) -> int:
        return 0 if n < self.win else 1 + (n - self.win) // self.hop

    def __call__(self, wav: torch.Tensor, nsamples: torch.Tensor = None, cmvn=None, out: torch.Tensor = None,
                 dither: float = 0.0, seed: int = 0):
        """dither: kaldi.fbank's waveform dither (dataset.py:98), N(0, dither^2) per window sample, keyed by `seed`."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2, "fbank needs a float32 CUDA (B,N) tensor"
        wav = wav.contiguous()
        B, N = wav.shape
        T = self.num_frames(N)
        if out is None:
            out = torch.empty(B, T, self.n_mel, device=wav.device)
        ns = None if nsamples is None else nsamples.to(torch.int32).contiguous()
        mean, istd = (None, None) if cmvn is None else cmvn
        if dither != 0.0:
            hip.call("oe_fbank_dither", wav, ns, B, N, T, self.win, self.hop, self.n_mel, self.scale, self.preemph, self.window,
                     self.twiddle, self.mel_start, self.mel_off, self.mel_w, FLT_EPS, mean, istd, float(dither), int(seed), out)
        else:
            hip.call("oe_fbank", wav, ns, B, N, T, self.win, self.hop, self.n_mel, self.scale, self.preemph, self.window,
                     self.twiddle, self.mel_start, self.mel_off, self.mel_w, FLT_EPS, mean, istd, out)
        if ns is None:
            nframes = torch.full((B,), T, dtype=torch.int32, device=wav.device)
        else:
            nframes = torch.where(ns < self.win, torch.zeros_like(ns), 1 + (ns - self.win) // self.hop).to(torch.int32)
        return out, nframes


def utt_normalize_(feats: torch.Tensor, nframes: torch.Tensor = None) -> torch.Tensor:
    """In-place per-utterance mean/std normalisation (feature_processor.py:5-8)."""
    B, T, F = feats.shape
    nf = None if nframes is None else nframes.to(torch.int32).contiguous()
    hip.call("oe_utt_normalize", feats, nf, B, T, F)
    return feats
