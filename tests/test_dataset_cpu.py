"""CPU: the data side of the hot path - manifest parsing and batch formation against the oracle
restatement of dataset.py:286-397, the rank split against torch's own DistributedSampler (the class train_ddp.py:131-134
uses), the balanced plan's invariants, a world-size-2 gloo check that both ranks derive the same plan, file readers."""
import math
import os
import random
import socket
import struct
import sys
import wave

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import batching as OB  # noqa: E402
from openeat_amd.dataset import dataset as D  # noqa: E402
from openeat_amd.dataset.audio_processor import _speed_generator, perturbed_length  # noqa: E402
from openeat_amd.dataset.sampler import DistributedBatchSampler, step_imbalance  # noqa: E402


def _manifest(tmp_path, n=57, seed=0, wav=False):
    rng = random.Random(seed)
    items, lines = [], []
    for i in range(n):
        frames = rng.randint(5, 1700)
        toks = [rng.randint(2, 40) for _ in range(rng.randint(0, 30))]
        key = f"utt{i:04d}"
        if wav:
            shape = f"{frames / 100:.2f}"
            frames = int(float(shape) * 1000 / 10)
            path = f"/data/w{i}.wav"
        else:
            shape, path = f"{frames},80", f"/data/f.ark:{i * 100 + 7}"
        tokstr = " ".join(map(str, toks))
        lines.append(f"utt:{key}\tfeat:{path}\tfeat_shape:{shape}\ttext:x\ttoken:x\ttokenid:{tokstr}\ttoken_shape:{len(toks)},41")
        items.append((key, path, frames, tokstr))
    lines.insert(3, "utt:broken\tfeat:/x")                      # malformed lines are skipped (dataset.py:330-331)
    p = tmp_path / "format.data"
    p.write_text("\n".join(lines) + "\n", encoding="utf-8")
    return str(p), items


@pytest.mark.parametrize("batch_type,sort,speed,wav", [("static", False, False, False), ("static", True, True, False),
                                                       ("dynamic", True, False, False), ("dynamic", False, True, True),
                                                       ("shuffle", True, False, True)])
def test_batch_formation_matches_the_restated_reference(tmp_path, batch_type, sort, speed, wav):
    path, items = _manifest(tmp_path, wav=wav)
    kw = dict(max_length=1500, min_length=10, token_max_length=60, token_min_length=1)
    ds = D.AudioDataset(path, {"<unk>": 1}, batch_type=batch_type, batch_size=7, max_frames_in_batch=4000, sort=sort,
                        speed_perturb=speed, data_type="wav" if wav else "kaldi", **kw)
    data = OB.expand_and_filter(items, speed_perturb=speed, **kw)
    want = OB.form_batches(data, batch_type, 7, 4000, sort)
    assert ds.data == want and len(ds) == len(want)
    assert ds[0] == want[0]
    if speed:                       # np.arange(0.9, 1.1, 0.1) has three elements; lengths compound (reference quirk kept)
        assert len(data) % 3 == 0 and data[1][2] == pytest.approx(data[0][2] * 1.0) and data[2][4] == pytest.approx(1.1)
    if not wav:
        assert ds.input_size == 80


def test_four_field_manifest_uses_the_char_dict(tmp_path):
    p = tmp_path / "m"
    p.write_text("utt:a\tfeat:/a.ark:5\tfeat_shape:100,80\ttext:你好<unk>吗\n", encoding="utf-8")
    ds = D.AudioDataset(str(p), {"<unk>": 1, "你": 5, "好": 6, "#": 9}, batch_type="static", batch_size=4)
    assert ds.data == [[("a", "/a.ark:5", [5, 6, 9, 1], 1.0)]]


def test_bucket_batches_respect_the_padded_budget():
    rng = random.Random(1)
    data = [(f"u{i}", "p", rng.randint(100, 1600), [1, 2], 1.0) for i in range(500)]
    lens = {e[0]: e[2] for e in data}
    bs = D.bucket_batches(data, max_padded_frames=12000, length_multiple=32)
    assert sorted(k for b in bs for k, *_ in b) == sorted(lens)                       # every utterance exactly once
    shapes = set()
    for b in bs:
        T = -(-max(lens[k] for k, *_ in b) // 32) * 32
        assert len(b) * T <= 12000 or len(b) == 1
        shapes.add(T)
    assert len(shapes) <= 48                                                          # 1600 / 32 = 50 possible lengths
    # padding waste is bounded by the rounding: true frames / padded frames
    true = sum(lens.values())
    padded = sum(len(b) * (-(-max(lens[k] for k, *_ in b) // 32) * 32) for b in bs)
    assert true / padded > 0.93


@pytest.mark.parametrize("n,world,shuffle", [(10, 2, True), (10, 4, True), (3, 8, True), (17, 8, False), (64, 8, True), (1, 4, True)])
def test_reference_mode_is_torch_distributed_sampler(n, world, shuffle):
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(n))
    for epoch in (0, 3):
        for rank in range(world):
            ref = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle, seed=0)
            ref.set_epoch(epoch)
            mine = DistributedBatchSampler(n, world, rank, shuffle=shuffle, seed=0, mode="reference")
            mine.set_epoch(epoch)
            assert list(mine) == list(ref) and len(mine) == len(ref)


def _sorted_dynamic_costs(seed=0, n=2000, budget=10000):
    rng = random.Random(seed)
    data = [(f"u{i}", "p", rng.randint(200, 1600), [1], 1.0) for i in range(n)]
    lens = {e[0]: e[2] for e in data}
    batches = D.make_batches(data, "dynamic", max_frames_in_batch=budget, sort=True)
    return [len(b) * max(lens[k] for k, *_ in b) for b in batches]                    # padded frames of each batch


def test_balanced_plan_covers_every_batch_once_and_evens_out_the_ranks():
    costs = _sorted_dynamic_costs()
    costs[:40] = [c * (0.3 + 0.02 * i) for i, c in enumerate(costs[:40])]            # short-utterance tail: uneven batches
    n, world = len(costs), 8
    ref = DistributedBatchSampler(n, world, 0, seed=5, mode="reference")
    bal = DistributedBatchSampler(n, world, 0, seed=5, mode="balanced", costs=costs)
    for epoch in (0, 1):
        bal.set_epoch(epoch)
        plan = bal.plan()
        flat = [i for step in plan for i in step]
        assert len(plan) == math.ceil(n / world) and all(len(s) == world for s in plan)
        assert set(flat) == set(range(n)) and len(flat) - n == (-n) % world           # each batch once + the padding repeats
        for r in range(world):                                                       # a rank's iterator is its column of the plan
            sr = DistributedBatchSampler(n, world, r, seed=5, mode="balanced", costs=costs)
            sr.set_epoch(epoch)
            assert list(sr) == [step[r] for step in plan] and len(sr) == len(plan)
    assert step_imbalance(bal.plan(), costs) < 0.25 * step_imbalance(ref.plan(), costs)
    e0, e1 = (bal.set_epoch(0), bal.plan())[1], (bal.set_epoch(1), bal.plan())[1]
    assert e0 != e1                                                                   # reshuffled per epoch


def _rank_worker(rank, world, port, costs, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = DistributedBatchSampler(len(costs), world, rank, seed=11, mode="balanced", costs=costs)
    s.set_epoch(2)
    mine = torch.tensor(list(s), dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)                                   # every rank sees what the others will train on
    plan = torch.tensor(s.plan(), dtype=torch.int64)             # [step][rank]
    for r in range(world):
        assert torch.equal(got[r], plan[:, r])                   # the ranks derived the same plan without talking
    assert len(set(torch.stack(got).flatten().tolist())) == len(costs)
    dist.destroy_process_group()
    q.put(rank)


def test_ranks_agree_on_the_plan_world2_gloo():
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    costs = _sorted_dynamic_costs(seed=2, n=300)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, costs, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def test_speed_draw_follows_the_reference_order():
    for speeds in (None, [0.9, 1.1, 0.1], [0.8, 1.2, 0], [1.05], [0.5, 2.0, 0.25]):
        random.seed(42)
        a = [_speed_generator(speeds) for _ in range(20)]
        random.seed(42)
        b = [OB.speed_generator(speeds) for _ in range(20)]
        assert a == b
    random.seed(0)
    assert {_speed_generator(None) for _ in range(50)} == {9 * 0.1}       # randrange(9, 10): the default always gives 0.9
    assert perturbed_length(160000, 0.9) == OB.speed_perturb_len(160000, 0.9) == 177778


def test_resampler_restatement_against_scipy_polyphase():
    """The oracle's windowed-sinc resampler vs scipy.signal.resample_poly on a band-limited signal (speed 0.9 = 10/9,
    1.1 ~ 10/11): two different designs of the same operation agree to ~1 % of the signal's RMS away from the edges."""
    from scipy.signal import resample_poly
    rng = np.random.default_rng(0)
    n = 4000
    t = np.arange(n)
    x = sum(rng.normal() * np.sin(2 * np.pi * f * t + rng.uniform(0, 6.28)) for f in rng.uniform(0.002, 0.15, 12))
    for speed, up, down in ((0.9, 10, 9), (1.1, 10, 11)):
        y = OB.speed_perturb(x, speed)
        z = resample_poly(x, up, down)
        m = min(len(y), len(z))
        assert abs(len(y) - len(z)) <= 1
        err = (y[200:m - 200] - z[200:m - 200])
        assert np.sqrt(np.mean(err ** 2)) < 0.012 * np.sqrt(np.mean(z ** 2))
    assert np.array_equal(OB.speed_perturb(x, 1.0), x)
    assert np.allclose(OB.speed_perturb(np.full(500, 3.0), 0.9)[40:-40], 3.0, atol=1e-9)     # unit DC gain


def test_wav_and_kaldi_matrix_readers(tmp_path):
    x = (np.sin(np.arange(1600) * 0.05) * 20000).astype("<i2")
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
    y, sr = D.read_wav(p)
    assert sr == 16000 and np.array_equal(y, x.astype(np.float32) / 32768.0)
    m = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.5
    ark = tmp_path / "f.ark"
    with open(ark, "wb") as f:
        f.write(b"utt1 ")
        off = f.tell()
        f.write(b"\0BFM \4" + struct.pack("<i", 3) + b"\4" + struct.pack("<i", 4) + m.tobytes())
    assert np.array_equal(D.read_kaldi_mat(f"{ark}:{off}"), m)


def test_collate_refuses_cpu():
    from openeat_amd.dataset.audio_processor import speed_perturb_batch
    with pytest.raises(TypeError, match="no CPU fallback"):
        speed_perturb_batch(torch.zeros(1, 100), [100], [0.9])
