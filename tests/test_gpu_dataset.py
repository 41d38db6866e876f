"""GPU: speed perturbation kernel against its float64 restatement, and the device collate (wav files -> speed ->
fbank -> normalisation -> spec-substitute -> SpecAugment -> padded batch dict) against the oracle pipeline run per
utterance in the reference's order of python-`random` draws (dataset.py:39-119, 186-240)."""
import os
import random
import sys
import wave

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

from oracle import augment as OA  # noqa: E402
from oracle import batching as OB  # noqa: E402
from oracle import fbank as FB  # noqa: E402

DEV = "cuda"


def test_speed_perturb_kernel_matches_float64_restatement():
    from openeat_amd.dataset.audio_processor import _speed_perturb, speed_perturb_batch
    rng = np.random.default_rng(3)
    lens = [5000, 4321, 3000, 777, 5000]
    speeds = [0.9, 1.0, 1.1, 0.77, 1.3]
    wav = np.zeros((5, 5000), dtype=np.float32)
    for b, n in enumerate(lens):
        wav[b, :n] = rng.uniform(-0.5, 0.5, n)
    out, n_out = speed_perturb_batch(torch.from_numpy(wav).to(DEV), lens, speeds)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    assert out.shape[1] == max(n_out)
    for b, (n, s) in enumerate(zip(lens, speeds)):
        ref = OB.speed_perturb(wav[b, :n], s)
        assert n_out[b] == len(ref)
        np.testing.assert_allclose(out[b, :n_out[b]], ref, rtol=0, atol=2e-5)          # |x| <= 0.5, fp32 taps vs float64
        assert np.all(out[b, n_out[b]:] == 0)                                          # batch padding zeroed
    assert np.array_equal(out[1, :lens[1]], wav[1, :lens[1]])                          # speed 1 is a copy
    one = _speed_perturb(torch.from_numpy(wav[:1]).to(DEV), 16000, 0.9)
    assert one.shape == (1, n_out[0]) and torch.equal(one.cpu(), torch.from_numpy(out[:1, :n_out[0]]))


def _write_wavs(tmp_path, lens, seed=0):
    rng = np.random.default_rng(seed)
    paths, sigs = [], []
    for i, n in enumerate(lens):
        x = (rng.uniform(-0.4, 0.4, n) * 32768).astype("<i2")
        p = str(tmp_path / f"u{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
        paths.append(p)
        sigs.append(x.astype(np.float32) / 32768.0)
    return paths, sigs


@pytest.mark.parametrize("speed_rate", [0.0, 1.0])
def test_device_collate_matches_the_per_utterance_pipeline(tmp_path, speed_rate):
    from openeat_amd.dataset.dataset import audio_collate_func
    lens = [16000, 9000, 12345, 20000]
    paths, sigs = _write_wavs(tmp_path, lens)
    labels = [[3, 4, 5], [7], [8, 9], [2, 2, 2, 6]]
    batch = [(f"k{i}", paths[i], labels[i], 1.0) for i in range(4)]
    conf = dict(mel_bins=80, wav_dither=0.0, speed_perturb_rate=speed_rate, speeds=[0.9, 1.1, 0.1])
    coll = audio_collate_func(spec_aug=True, spec_aug_conf=dict(num_t_mask=2, num_f_mask=2, max_t=20, max_f=10), spec_sub=True,
                              spec_sub_conf=dict(max_t=10, num_t_sub=2), data_type="wav", feature_extraction_conf=conf, device=DEV)
    random.seed(7)
    keys, out = coll([batch])                       # DataLoader(batch_size=1) hands the pre-formed batch in a list
    torch.cuda.synchronize()
    # the same thing one utterance at a time on the CPU, python-random draws in the reference's order
    random.seed(7)
    feats = []
    for i in range(4):
        x = sigs[i]
        speed = 1.0
        if random.random() < speed_rate:
            speed = OB.speed_generator(conf["speeds"])
        if speed != 1.0:
            x = OB.speed_perturb(x, speed).astype(np.float32)
        feats.append(FB.utt_normalize(FB.fbank(torch.from_numpy(x))).numpy())
    order = np.argsort([f.shape[0] for f in feats])[::-1]
    feats = [feats[i] for i in order]
    feats = [OA.spec_substitute(f, max_t=10, num_t_sub=2) for f in feats]
    feats = [OA.spec_augmentation(f, num_t_mask=2, num_f_mask=2, max_t=20, max_f=10) for f in feats]
    assert keys == [f"k{i}" for i in order]
    assert out["features_length"].tolist() == [f.shape[0] for f in feats]
    assert out["features"].shape == (4, feats[0].shape[0], 80) and out["features"].is_cuda
    got = out["features"].cpu().numpy()
    tol = dict(rtol=2e-3, atol=5e-3) if speed_rate == 0.0 else dict(rtol=5e-3, atol=2e-2)   # normalised log-mel; resampler taps fp32
    for r, f in enumerate(feats):
        np.testing.assert_allclose(got[r, :f.shape[0]], f, **tol)
        assert np.all(got[r, f.shape[0]:] == 0)
        assert np.array_equal(got[r, :f.shape[0]] == 0, f == 0)                      # the masks sit on the same cells
    want_t = [labels[i] for i in order]
    assert out["targets_length"].tolist() == [len(t) for t in want_t]
    tg = out["targets"].cpu()
    assert tg.dtype == torch.int32 and tg.shape == (4, 4)
    for r, t in enumerate(want_t):
        assert tg[r, :len(t)].tolist() == t and bool((tg[r, len(t):] == -1).all())


def test_collated_batch_trains_a_step(tmp_path):
    """model(**batch) with the collate's dict, as executor.py:47 calls it."""
    from openeat_amd.dataset.dataset import audio_collate_func
    from openeat_amd.models.asr_model import ASRModel
    paths, _ = _write_wavs(tmp_path, [16000, 12000, 14000], seed=1)
    batch = [(f"k{i}", paths[i], [3 + i, 4, 5], 1.0) for i in range(3)]
    coll = audio_collate_func(data_type="wav", feature_extraction_conf=dict(mel_bins=80, wav_dither=0.0, speed_perturb_rate=0.0), device=DEV)
    _, b = coll(batch)
    torch.manual_seed(0)
    m = ASRModel(80, 30, encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
                 linear_units=64, reverse_weight=0.3, dropout_rate=0.0).to(DEV)
    loss, acc = m(**b)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_manifest_to_trained_epoch_as_train_py_wires_it(tmp_path):
    """The reference's training wiring (bin/train.py:96-116,154-240) on the drop-in modules: manifest -> AudioDataset (dynamic
    batches) -> sampler -> DataLoader(batch_size=1, collate_fn) -> Executor.train / cv with WarmupLR -> CTC greedy search."""
    import logging
    from torch.utils.data import DataLoader
    from openeat.dataset.dataset import AudioDataset, audio_collate_func
    from openeat.dataset.sampler import DistributedBatchSampler
    from openeat.models.asr_model import ASRModel
    from openeat.utils.executor import Executor
    from openeat.utils.scheduler import WarmupLR
    rng = np.random.default_rng(0)
    chars = ["<blank>", "<unk>"] + [chr(0x4E00 + i) for i in range(20)] + ["<sos/eos>"]
    char_dict = {c: i for i, c in enumerate(chars)}
    lines = []
    for i in range(14):
        sec = float(rng.uniform(0.6, 1.6))
        x = (rng.uniform(-0.3, 0.3, int(sec * 16000)) * 32768).astype("<i2")
        p = str(tmp_path / f"w{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
        text = "".join(chars[2 + int(c)] for c in rng.integers(0, 20, int(rng.integers(2, 6))))
        lines.append(f"utt:u{i}\tfeat:{p}\tfeat_shape:{len(x) / 16000:.2f}\ttext:{text}")
    man = tmp_path / "format.data"
    man.write_text("\n".join(lines) + "\n", encoding="utf-8")
    ds = AudioDataset(str(man), char_dict, batch_type="dynamic", max_frames_in_batch=450, sort=True, data_type="wav", min_length=10)
    assert 3 <= len(ds) <= 14 and sum(len(b) for b in ds.data) == 14
    coll = audio_collate_func(spec_aug=True, spec_aug_conf=dict(num_t_mask=1, num_f_mask=1, max_t=5, max_f=4), data_type="wav",
                              feature_extraction_conf=dict(mel_bins=80, wav_dither=0.0, speed_perturb_rate=0.5, speeds=[0.9, 1.1, 0.1]),
                              device=DEV)
    sampler = DistributedBatchSampler(len(ds), 1, 0, shuffle=True, seed=3, mode="reference")
    loader = DataLoader(ds, batch_size=1, sampler=sampler, collate_fn=coll, num_workers=0)
    torch.manual_seed(1)
    random.seed(1)
    model = ASRModel(80, len(chars), encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
                     linear_units=64, reverse_weight=0.3, dropout_rate=0.1).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    sched = WarmupLR(opt, warmup_steps=5)
    ex, log = Executor(), logging.getLogger("t")
    args = dict(grad_clip=5.0, accum_grad=2, log_interval=100)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    losses = []
    for epoch in range(4):
        sampler.set_epoch(epoch)
        losses.append(ex.train(log, model, opt, sched, loader, torch.device(DEV), args)[0])
    cv_loss, cv_acc = ex.cv(log, model, loader, torch.device(DEV), args)
    assert all(np.isfinite(l) for l in losses) and np.isfinite(cv_loss) and 0.0 <= cv_acc <= 1.0
    assert losses[-1] < losses[0]                                                 # four epochs on 14 utterances: it learns something
    assert ex.step == 4 * -(-len(ds) // 2) and any(not torch.equal(v, model.state_dict()[k]) for k, v in before.items())
    model.eval()
    keys, b = coll(ds[0])
    hyps = model.ctc_greedy_search(b["features"], b["features_length"])
    assert len(hyps) == len(keys) and all(isinstance(t, int) and 0 < t < len(chars) for h in hyps for t in h)


def test_fbank_waveform_dither():
    """kaldi.fbank(dither=wav_dither) (dataset.py:98): dither 0 is the plain kernel bit for bit; the noise is a function of the
    seed; on silence the log-mel statistics are those of white noise of that standard deviation through the oracle."""
    from openeat_amd.frontend import Fbank
    fb = Fbank(num_mel_bins=80, sample_rate=16000.0, device=DEV)
    rng = np.random.default_rng(5)
    wav = torch.from_numpy(rng.uniform(-0.3, 0.3, (3, 16000)).astype(np.float32)).to(DEV)
    ns = torch.tensor([16000, 12000, 8000], dtype=torch.int32, device=DEV)
    plain, _ = fb(wav, ns)
    zero, _ = fb(wav, ns, dither=0.0, seed=9)
    assert torch.equal(plain, zero)
    a, _ = fb(wav, ns, dither=1.0, seed=9)
    b, _ = fb(wav, ns, dither=1.0, seed=9)
    c, _ = fb(wav, ns, dither=1.0, seed=10)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float((a - plain).abs().max()) < 0.05            # one LSB of noise under a 0.3 full-scale signal barely moves the log-mels
    assert torch.equal(a[1, fb.num_frames(12000):], torch.zeros_like(a[1, fb.num_frames(12000):]))
    # silence + dither d  ==  white noise of sigma d / 32768 (stationary, so independent per-frame draws have the same law)
    d = 2.0
    sil = torch.zeros(4, 160000, device=DEV)
    got, _ = fb(sil, None, dither=d, seed=123)
    ref = FB.fbank(torch.from_numpy(rng.normal(0, d / 32768.0, 640000).astype(np.float32)))
    gm, rm = got.reshape(-1, 80).double().mean(0).cpu(), ref.double().mean(0)
    gs, rs = got.reshape(-1, 80).double().std(0).cpu(), ref.double().std(0)
    # per-bin standard error of the difference of the two means: ours has ~4000 independent frames, the oracle's 4000 frames
    # overlap by 60 % (counted as 1600); the narrow low mel bins have a log-energy spread of ~1, the wide high ones ~0.1
    se = rs * (1.0 / got.reshape(-1, 80).shape[0] + 1.0 / 1600) ** 0.5
    assert bool(((gm - rm).abs() < 4.5 * se + 0.003).all()), ((gm - rm).abs() / se).max()
    assert float((gs / rs - 1).abs().max()) < 0.1


def test_collate_resamples_to_resample_rate(tmp_path):
    """dataset.py:81-84: a file whose rate differs from resample_rate is resampled before the features.  A band-limited signal
    decimated by two and written at 8 kHz comes back to the 16 kHz original."""
    from openeat_amd.dataset.audio_processor import speed_perturb_batch
    from openeat_amd.dataset.dataset import audio_collate_func
    rng = np.random.default_rng(11)
    n = 32000
    spec = np.fft.rfft(rng.normal(0, 1, n))
    spec[int(3000 / 8000 * (n // 2)):] = 0                   # nothing above 3 kHz
    x16 = np.fft.irfft(spec, n)
    x16 = (0.3 * x16 / np.abs(x16).max()).astype(np.float32)
    x8 = np.ascontiguousarray(x16[::2])
    up, n_up = speed_perturb_batch(torch.from_numpy(x8[None]).to(DEV), [n // 2], [0.5])
    assert n_up == [n]
    err = (up[0].cpu().numpy() - x16)[200:-200]
    assert np.abs(err).max() < 2e-3, np.abs(err).max()
    p = str(tmp_path / "u8k.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000); w.writeframes((x8 * 32768).astype("<i2").tobytes())
    conf = dict(mel_bins=80, wav_dither=0.0, speed_perturb_rate=0.0, resample_rate=16000)
    coll = audio_collate_func(data_type="wav", feature_extraction_conf=conf, device=DEV)
    keys, out = coll([("k0", p, [3, 4], 1.0)])
    want = FB.utt_normalize(FB.fbank(torch.from_numpy(x16))).numpy()
    assert out["features_length"].tolist() == [want.shape[0]]
    got = out["features"][0].cpu().numpy()
    lo = slice(0, 50)                                        # mel bins below ~2.6 kHz, where the signal lives
    assert np.abs(got[5:-5, lo] - want[5:-5, lo]).max() < 0.1, np.abs(got[5:-5, lo] - want[5:-5, lo]).max()
