"""Manifest -> batches -> padded DEVICE batch, the data side of the hot path.

Mirrors /root/reference/openeat/dataset/dataset.py: `AudioDataset` (manifest parsing, length filters, speed copies,
'static' / 'dynamic' / 'shuffle' batch formation, :262-403) and `audio_collate_func` (:156-240), whose arithmetic -
speed perturbation, fbank, per-utterance normalisation, feature dither, spec-substitute, SpecAugment - runs on the GPU
on the whole padded batch instead of per utterance in numpy / torchaudio / sox worker processes.  The batch dict has the
reference's keys (`features`, `features_length`, `targets`, `targets_length`), so `model(**batch)` is unchanged.

MI355X-first additions (not in the reference): `bucket_batches` (frame-budget batches that count PADDED frames and
round the batch's length up to a multiple, so that few distinct shapes reach the kernels / HIP-graph cache) - see
also sampler.DistributedBatchSampler(mode="balanced").

Out of scope: sox-backed decoding of arbitrary audio containers (plain PCM16 / float32 RIFF wav is read with the
standard library), Kaldi compressed matrices, text tokenisation beyond characters / a caller-supplied tokenizer
(`_remove_punctuation` needs `zhon`, sentencepiece models are a storage format).
"""
import codecs
import logging
import random
import struct
import wave
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from openeat_amd import augment
from openeat_amd.dataset.audio_processor import _speed_generator, speed_perturb_batch
from openeat_amd.frontend import Fbank, utt_normalize_
from openeat_amd.utils.common import IGNORE_ID


# ------------------------------------------------------------------ manifest ----
def parse_manifest_line(line: str, char_dict: Dict[str, int], data_type: str = "kaldi", tokenizer: Optional[Callable] = None):
    """One line of the 4- or 7-field manifest written by tools/format_data.sh (dataset.py:328-353).
    Returns (key, path, num_frames, tokenid, feat_dim or None) or None for a malformed line."""
    arr = line.strip().split('\t')
    if len(arr) != 4 and len(arr) != 7:
        return None
    key = arr[0].split(':')[1]
    if len(arr) == 4:
        text = arr[3].split(':')[1]
        text = text.replace('<unk>', '#')
        tokens = tokenizer(text) if tokenizer is not None else [c for c in text if not c.isspace()]
        tokenid = [char_dict[w] if w in char_dict else char_dict['<unk>'] for w in tokens]
    else:
        tokenid = arr[5].split(':')[1]          # kept as the reference keeps it (a string; :343)
    path = ':'.join(arr[1].split(':')[1:])
    feat_dim = None
    if data_type == 'wav':
        num_frames = int(float(arr[2].split(':')[1]) * 1000 / 10)
    else:
        feat_info = arr[2].split(':')[1].split(',')
        feat_dim = int(feat_info[1].strip())
        num_frames = int(feat_info[0].strip())
    return key, path, num_frames, tokenid, feat_dim


class AudioDataset(torch.utils.data.Dataset):
    """dataset.py:262-403: same constructor, same `self.data` (a list of batches of (key, path, tokenid, speed))."""

    def __init__(self, data_file, char_dict, bpe_model=None, max_length=10240, min_length=0, token_max_length=200,
                 token_min_length=0, batch_type='static', batch_size=1, max_frames_in_batch=0, sort=False,
                 speed_perturb=False, speeds=[0.9, 1.1, 0.1], data_type="kaldi", tokenizer: Optional[Callable] = None):
        assert batch_type in ['static', 'dynamic', 'shuffle']
        if bpe_model is not None and tokenizer is None:
            raise NotImplementedError("sentencepiece models are out of scope: pass tokenizer=callable(text) -> tokens")
        self.batch_size = 1 if batch_type in ['static', 'dynamic'] else batch_size
        self.char_dict = char_dict
        self.vocab_size = len(char_dict)
        speed_list = [float(s) for s in np.arange(speeds[0], speeds[1], speeds[2])] if speed_perturb else [1.0]
        data = []
        with codecs.open(data_file, 'r', encoding='utf-8') as f:
            for line in f:
                rec = parse_manifest_line(line, char_dict, data_type, tokenizer)
                if rec is None:
                    continue
                key, path, num_frames, tokenid, feat_dim = rec
                if feat_dim is not None:
                    self.input_size = feat_dim
                if min_length < num_frames < max_length and token_min_length < len(tokenid) < token_max_length:
                    for speed in speed_list:
                        num_frames *= speed            # cumulative, as the reference (dataset.py:360-361)
                        data.append((key, path, num_frames, tokenid, speed))
        self.entries = data
        self.data = make_batches(data, batch_type, batch_size, max_frames_in_batch, sort)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx]


def make_batches(data: Sequence, batch_type='static', batch_size=1, max_frames_in_batch=0, sort=False) -> List:
    """dataset.py:363-397 on entries (key, path, num_frames, tokenid, speed)."""
    assert batch_type in ['static', 'dynamic', 'shuffle']
    if sort:
        data = sorted(data, key=lambda x: x[2])            # stable, as the reference's sorted()
    item = lambda e: (e[0], e[1], e[3], e[4])
    if batch_type == 'shuffle':
        return [list(item(e)) for e in data]
    if batch_type == 'static':
        return [[item(e) for e in data[c:c + batch_size]] for c in range(0, len(data), batch_size)]
    assert max_frames_in_batch > 0
    # a batch is closed by the first utterance that takes the running sum of frames past the budget
    batches, frames = [[]], 0
    for e in data:
        if frames + e[2] > max_frames_in_batch:
            batches.append([])
            frames = 0
        frames += e[2]
        batches[-1].append(item(e))
    return batches


def bucket_batches(data: Sequence, max_padded_frames: int, length_multiple: int = 32, max_utts: int = 0) -> List:
    """MI355X-first frame-budget batching: entries sorted by length, a batch holds as many consecutive utterances as
    keep  n_utts * round_up(longest, length_multiple)  <= max_padded_frames  (what the kernels actually process,
    padding included; the reference budgets the SUM of true lengths).  Batch lengths are multiples of
    `length_multiple` frames, so a corpus maps to a few dozen distinct (B, T) shapes."""
    data = sorted(data, key=lambda x: x[2])
    rup = lambda n: -(-int(np.ceil(n)) // length_multiple) * length_multiple
    batches, cur = [], []
    for e in data:
        n = len(cur) + 1
        if cur and (n * rup(e[2]) > max_padded_frames or (max_utts and n > max_utts)):
            batches.append(cur)
            cur = []
        cur.append((e[0], e[1], e[3], e[4]))
    if cur:
        batches.append(cur)
    return batches


# ------------------------------------------------------------------ file formats ----
def read_wav(path: str):
    """PCM16 / PCM32 / float32 RIFF wav -> (float32 array in [-1, 1), sample_rate); first channel only
    (torchaudio.load's normalised convention, dataset.py:63-74; a `path,start,end` segment is cut by the caller)."""
    with wave.open(path, 'rb') as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, dtype='<i4').astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"{path}: unsupported sample width {sw}")
    return x.reshape(-1, nch)[:, 0].copy(), sr


def read_kaldi_mat(path_offset: str) -> np.ndarray:
    """`file.ark:offset` -> float32 matrix (uncompressed binary FM / DM only; what kaldi_io.read_mat is used for at
    dataset.py:131)."""
    path, off = path_offset.rsplit(':', 1)
    with open(path, 'rb') as f:
        f.seek(int(off))
        if f.read(2) != b'\0B':
            raise ValueError(f"{path_offset}: not a binary Kaldi object")
        kind = f.read(3)
        if kind not in (b'FM ', b'DM '):
            raise ValueError(f"{path_offset}: unsupported Kaldi matrix type {kind!r} (compressed matrices are out of scope)")
        assert f.read(1) == b'\4'
        rows = struct.unpack('<i', f.read(4))[0]
        assert f.read(1) == b'\4'
        cols = struct.unpack('<i', f.read(4))[0]
        dt = '<f4' if kind == b'FM ' else '<f8'
        return np.frombuffer(f.read(rows * cols * np.dtype(dt).itemsize), dtype=dt).reshape(rows, cols).astype(np.float32)


# ------------------------------------------------------------------ collate ----
class audio_collate_func(object):
    """dataset.py:156-240 with the arithmetic on the device.  Order of operations and of python-`random` draws as the
    reference: per utterance [speed draw], then for the batch: normalisation, dither (one draw), every utterance's
    spec-substitute, every utterance's SpecAugment; utterances sorted by length, longest first (:115-119)."""

    def __init__(self, feature_dither=0.0, spec_aug=False, spec_aug_conf=None, spec_sub=False, spec_sub_conf=None,
                 data_type="kaldi", feature_extraction_conf=None, normalization=True, device="cuda",
                 wav_reader: Callable = read_wav, mat_reader: Callable = read_kaldi_mat):
        self.feature_dither = feature_dither
        self.spec_sub = spec_sub
        self.spec_aug = spec_aug
        self.spec_sub_conf = spec_sub_conf or {}
        self.spec_aug_conf = spec_aug_conf or {}
        self.data_type = data_type
        self.feature_extraction_conf = feature_extraction_conf or {}
        self.normalization = normalization
        self.device = torch.device(device)
        self.wav_reader, self.mat_reader = wav_reader, mat_reader
        self._fbank = None

    # -- dataset.py:39-119, the device version
    def _extract_feature(self, batch):
        conf = self.feature_extraction_conf
        rate, speeds = conf.get('speed_perturb_rate', 0.5), conf.get('speeds', None)
        keys, wavs, labels, spd = [], [], [], []
        sample_rate = None
        for x in batch:
            try:
                value = x[1].strip().split(",")
                assert len(value) == 1 or len(value) == 3
                w, sr = self.wav_reader(value[0])
                if len(value) == 3:
                    w = w[int(float(value[1]) * sr):int(float(value[2]) * sr)]
                speed = x[3]
                if random.random() < rate:
                    speed = _speed_generator(speeds)
                if 'resample_rate' in conf and conf['resample_rate'] != sr:
                    # dataset.py:77-90 resamples first (torchaudio Resample to resample_rate), then perturbs the speed at that
                    # rate; here both are ONE pass of the same windowed-sinc interpolator: read sr / resample_rate times
                    # faster, times the speed.  Not sample-exact with torchaudio's Resample (another kernel; the output
                    # length can differ by one sample): distribution-level parity, stated in DESIGN section 7.
                    speed = float(speed) * sr / conf['resample_rate']
                    sr = conf['resample_rate']
                assert sample_rate in (None, sr), "one sample rate per batch"
                sample_rate = sr
                keys.append(x[0]); wavs.append(w); labels.append(np.array(x[2])); spd.append(float(speed))
            except NotImplementedError:
                raise
            except Exception as e:      # the reference logs and drops the utterance (:108-111)
                logging.warning('read utterance {} error: {}'.format(x[0], e))
        if not wavs:
            return [], None, [], []
        n = [len(w) for w in wavs]
        host = torch.zeros(len(wavs), max(n), dtype=torch.float32)
        for i, w in enumerate(wavs):
            host[i, :n[i]] = torch.from_numpy(w)
        wav = host.to(self.device)
        if any(s != 1.0 for s in spd):
            wav, n = speed_perturb_batch(wav, n, spd)
        if self._fbank is None or self._fbank_rate != sample_rate:
            self._fbank = Fbank(num_mel_bins=conf.get('mel_bins', 80), sample_rate=float(sample_rate), device=self.device)
            self._fbank_rate = sample_rate
        fb = self._fbank
        nfr = [fb.num_frames(k) for k in n]
        dither = float(conf.get('wav_dither', 0.0))                    # dataset.py:98; the noise is keyed by a host-drawn seed
        feats, _ = fb(wav, torch.tensor(n, dtype=torch.int32, device=self.device), dither=dither,
                      seed=random.getrandbits(63) if dither != 0.0 else 0)
        order = np.argsort(nfr)[::-1]                                   # :115
        idx = torch.as_tensor(order.copy(), device=self.device)
        return [keys[i] for i in order], feats.index_select(0, idx), [nfr[i] for i in order], [labels[i] for i in order]

    # -- dataset.py:121-154
    def _load_feature(self, batch):
        keys, mats, labels = [], [], []
        for x in batch:
            try:
                m = self.mat_reader(x[1])
                keys.append(x[0]); mats.append(m); labels.append(np.array(x[2]))
            except Exception as e:
                logging.warning('read utterance {} error: {}'.format(x[0], e))
        if not mats:
            return [], None, [], []
        nfr = [m.shape[0] for m in mats]
        order = np.argsort(nfr)[::-1]
        host = torch.zeros(len(mats), max(nfr), mats[0].shape[1], dtype=torch.float32)
        for r, i in enumerate(order):
            host[r, :nfr[i]] = torch.from_numpy(mats[i])
        return [keys[i] for i in order], host.to(self.device), [nfr[i] for i in order], [labels[i] for i in order]

    def __call__(self, batch):
        if len(batch) == 1 and isinstance(batch[0], list):
            batch = batch[0]
        keys, feats, nfr, ys = self._extract_feature(batch) if self.data_type == 'wav' else self._load_feature(batch)
        train_flag = not (len(ys) > 0 and ys[0].ndim == 0)             # labels absent at inference
        if feats is None:
            return keys, {'features': torch.zeros(0), 'features_length': torch.zeros(0, dtype=torch.int32), 'targets': None,
                          'targets_length': None}
        nf_dev = torch.tensor(nfr, dtype=torch.int32, device=self.device)
        if self.normalization:
            utt_normalize_(feats, nf_dev)
        if self.feature_dither != 0.0:
            augment.feature_dither_(feats, nfr, self.feature_dither)
        if self.spec_sub:
            augment.spec_substitute_(feats, nfr, **self.spec_sub_conf)
        if self.spec_aug:
            augment.spec_augment_(feats, nfr, **self.spec_aug_conf)
        targets = targets_length = None
        if train_flag:
            L = max(len(y) for y in ys)
            t = np.full((len(ys), L), IGNORE_ID, dtype=np.int32)
            for i, y in enumerate(ys):
                t[i, :len(y)] = y
            targets = torch.from_numpy(t).to(self.device)
            targets_length = torch.tensor([len(y) for y in ys], dtype=torch.int32, device=self.device)
        return keys, {'features': feats, 'features_length': nf_dev, 'targets': targets, 'targets_length': targets_length}
