"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's data-side index work around the hot path:

* batch formation of AudioDataset (/root/reference/openeat/dataset/dataset.py:286-364): manifest lines -> filtered,
  speed-expanded, optionally length-sorted entries -> 'static' / 'dynamic' / 'shuffle' batches;
* the speed draw (/root/reference/openeat/dataset/audio_processor.py:5-19);
* speed perturbation itself (audio_processor.py:20-35) as a float64 windowed-sinc resampler - sox is outside the
  reference tree, so that part is the build's own definition (distribution parity).

PARITY UNPINNED for the batch formation: openeat.dataset.dataset cannot be imported here (kaldi_io, torchaudio and zhon
are not installed - SURVEY 8c) and the reference has no fixtures for it; this follows the source line by line and the
product (openeat_amd/dataset/dataset.py) is compared with it on synthetic manifests.  The rank split is pinned
separately against torch.utils.data.DistributedSampler (the class train_ddp.py:131-134 uses), which is importable.
Pure python / numpy; nothing here is ever imported by the product path."""
import math
import random

import numpy as np


def speed_list(speed_perturb, speeds):
    """dataset.py:322-325."""
    if speed_perturb:
        return [float(s) for s in np.arange(speeds[0], speeds[1], speeds[2])]
    return [1.0]


def expand_and_filter(items, max_length=10240, min_length=0, token_max_length=200, token_min_length=0, speed_perturb=False,
                      speeds=(0.9, 1.1, 0.1)):
    """dataset.py:326-362.  items: (key, path, num_frames, tokenid list).  Returns (key, path, num_frames, tokenid, speed).
    The reference multiplies num_frames by each speed IN TURN without resetting it (dataset.py:360-361), so the recorded
    length of the 2nd copy is frames * s1 * s2: reproduced."""
    data = []
    for key, path, num_frames, tokenid in items:
        length = num_frames
        token_length = len(tokenid)
        if min_length < length < max_length and token_min_length < token_length < token_max_length:
            for speed in speed_list(speed_perturb, speeds):
                num_frames *= speed
                data.append((key, path, num_frames, tokenid, speed))
    return data


def form_batches(data, batch_type="static", batch_size=1, max_frames_in_batch=0, sort=False):
    """dataset.py:363-397: list of batches, each a list of (key, path, tokenid, speed)."""
    assert batch_type in ["static", "dynamic", "shuffle"]
    if sort:
        data = sorted(data, key=lambda x: x[2])
    num_data = len(data)
    if batch_type == "dynamic":
        assert max_frames_in_batch > 0
        out = [[]]
        num_frames_in_batch = 0
        for i in range(num_data):
            length = data[i][2]
            num_frames_in_batch += length
            if num_frames_in_batch > max_frames_in_batch:
                out.append([])
                num_frames_in_batch = length
            out[-1].append((data[i][0], data[i][1], data[i][3], data[i][4]))
        return out
    if batch_type == "static":
        cur = 0
        out = []
        while cur < num_data:
            end = min(cur + batch_size, num_data)
            out.append([(data[i][0], data[i][1], data[i][3], data[i][4]) for i in range(cur, end)])
            cur = end
        return out
    return [[data[i][0], data[i][1], data[i][3], data[i][4]] for i in range(num_data)]


def speed_generator(speeds):
    """audio_processor.py:5-19, python `random` in the reference's call order."""
    if speeds is None:
        speeds = [0.9, 1.1, 0.1]
    speeds = [float(s) for s in speeds]
    if len(speeds) > 1:
        assert speeds[1] > speeds[0], "speeds is wrong !"
        if speeds[2] != 0:
            speed = random.randrange(int(speeds[0] / speeds[2]), int(speeds[0] / speeds[2]) + 1)
            speed *= speeds[2]
        else:
            speed = speeds[0] + random.random() * (speeds[1] - speeds[0])
    else:
        speed = speeds[0]
    return speed


def speed_perturb_len(n, speed):
    return int(math.floor(n / speed + 0.5))


def speed_perturb(x, speed, zeros=16):
    """out[i] = x(i * speed) through a Hann-windowed sinc, cutoff 0.95 * min(1, 1/speed), unit DC gain (float64)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    if speed == 1.0:
        return x.copy()
    s = float(np.float32(speed))
    no = speed_perturb_len(n, speed)
    fc = 0.95 * min(1.0, 1.0 / s)
    half = zeros / fc
    R = int(math.ceil(half))
    y = np.zeros(no)
    ks = np.arange(-R + 1, R + 1)
    for i in range(no):
        p = i * s
        c = int(math.floor(p))
        d = ks - (p - c)
        a = np.pi * fc * d
        w = np.where(np.abs(a) < 1e-6, 1.0, np.sin(a) / np.where(a == 0, 1.0, a)) * (0.5 + 0.5 * np.cos(np.pi * d / half))
        w = np.where(np.abs(d) >= half, 0.0, w)
        j = c + ks
        ok = (j >= 0) & (j < n)
        y[i] = np.dot(w[ok], x[j[ok]]) / w.sum()
    return y
