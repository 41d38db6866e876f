#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_fixtures.py

The reference is imported as-is from /root/reference with an in-memory stub
for the absent ``typeguard`` package (``check_argument_types`` only; imported
at encoder.py:9, decoder.py:7, ctc.py:3, convolution.py:12, scheduler.py:6).
All fixtures are seeded, fp32, dropout 0 / eval, and small (<~1.5 MB each).
Each .npz holds inputs (``in/..``), parameters (``sd/..``, reference state-dict
key names), outputs (``out/..``) and gradients (``grad/..``).  Only data is
stored - no reference source text.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

tg = types.ModuleType("typeguard")
tg.check_argument_types = lambda: True
sys.modules["typeguard"] = tg
sys.path.insert(0, REF)

from openeat.models.asr_model import ASRModel  # noqa: E402
from openeat.modules.attention import MultiHeadedAttention, RelPositionMultiHeadedAttention  # noqa: E402
from openeat.modules.convolution import ConvolutionModule  # noqa: E402
from openeat.modules.ctc import CTC  # noqa: E402
from openeat.modules.decoder import BiTransformerDecoder  # noqa: E402
from openeat.modules.embedding import PositionalEncoding, RelPositionalEncoding  # noqa: E402
from openeat.modules.encoder import TransformerEncoder  # noqa: E402
from openeat.modules.label_smoothing_loss import LabelSmoothingLoss  # noqa: E402
from openeat.modules.subsampling import Conv2dSubsampling4  # noqa: E402
from openeat.modules.swish import Swish  # noqa: E402
from openeat.modules.cmvn import GlobalCMVN  # noqa: E402
from openeat.utils import common as rc  # noqa: E402
from openeat.utils import mask as rm  # noqa: E402
from openeat.utils.scheduler import WarmupLR  # noqa: E402
from openeat.utils.cmvn import load_cmvn  # noqa: E402
from openeat.dataset.feature_processor import _normalization  # noqa: E402


def npy(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def save(name, **groups):
    flat = {}
    for g, d in groups.items():
        for k, v in d.items():
            flat[f"{g}/{k}"] = npy(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"{name}: {os.path.getsize(path) / 1e3:.1f} kB, {len(flat)} arrays")


def sd_of(mod, prefix=""):
    return {prefix + k: v.clone() for k, v in mod.state_dict().items()}


def grads_of(mod, prefix=""):
    return {prefix + k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}


def draw_param(shape, g, scale):
    """Deterministic parameter draw shared with tests/conftest.py (F11 stores
    only the key order + seed and re-draws its 1.7 M parameters there)."""
    shape = tuple(shape)
    t = torch.randn(shape, generator=g)
    if len(shape) > 1:
        return t * (3.0 * scale / max(1.0, shape[-1] ** 0.5))
    return t * scale + 0.5


def randomize(mod, seed, scale=0.3):
    """Re-draw every parameter (incl. LayerNorm affine, biases) so that no
    term is hidden behind a zero/one initial value."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in mod.parameters():
            p.copy_(draw_param(p.shape, g, scale))


def ragged_mask(lens, T):
    return ~rm.make_pad_mask(torch.tensor(lens), T).unsqueeze(1)


def f1_subsampling():
    torch.manual_seed(101)
    m = Conv2dSubsampling4(80, 32, RelPositionalEncoding(32))
    randomize(m, 1)
    lens = [67, 50, 23]
    x = torch.randn(3, 67, 80, requires_grad=True)
    mask = ragged_mask(lens, 67)
    y, ymask, pos = m(x, mask)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    m2 = Conv2dSubsampling4(80, 32, PositionalEncoding(32))
    m2.load_state_dict(m.state_dict())
    y2, _, _ = m2(x.detach(), mask)
    save("f01_subsampling4", **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(m, "encoder.embed."),
                                "out": {"y": y, "mask": ymask, "pos": pos, "y_abs": y2},
                                "grad": {**grads_of(m, "encoder.embed."), "x": x.grad}})


def f2_relpos_mha():
    torch.manual_seed(102)
    m = RelPositionMultiHeadedAttention(4, 32, 0.0)
    randomize(m, 2)
    B, T = 3, 13
    x = torch.randn(B, T, 32, requires_grad=True)
    mask = ragged_mask([13, 9, 4], T)
    pos = RelPositionalEncoding(32)(x.detach())[1]
    y = m(x, x, x, mask, pos)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("f02_relpos_mha", **{"in": {"x": x, "mask": mask, "pos": pos, "w": w}, "sd": sd_of(m, "attn."),
                              "out": {"y": y}, "grad": {**grads_of(m, "attn."), "x": x.grad}})


def f3_mha():
    torch.manual_seed(103)
    m = MultiHeadedAttention(4, 32, 0.0)
    randomize(m, 3)
    B, T1, T2 = 3, 7, 11
    q = torch.randn(B, T1, 32, requires_grad=True)
    kv = torch.randn(B, T2, 32, requires_grad=True)
    mask_k = ragged_mask([11, 6, 3], T2)                      # (B,1,T2)
    y1 = m(q, kv, kv, mask_k)
    w1 = torch.randn_like(y1)
    (y1 * w1).sum().backward()
    g1 = {**grads_of(m, "attn."), "q": q.grad.clone(), "kv": kv.grad.clone()}
    m.zero_grad(); q.grad = None; kv.grad = None
    s = torch.randn(B, T1, 32, requires_grad=True)
    mask_full = ragged_mask([7, 5, 2], T1) & rm.subsequent_mask(T1).unsqueeze(0)   # (B,T1,T1)
    y2 = m(s, s, s, mask_full)
    w2 = torch.randn_like(y2)
    (y2 * w2).sum().backward()
    g2 = {**grads_of(m, "attn."), "s": s.grad.clone()}
    save("f03_mha", **{"in": {"q": q, "kv": kv, "mask_k": mask_k, "w1": w1, "s": s, "mask_full": mask_full, "w2": w2},
                       "sd": sd_of(m, "attn."), "out": {"y1": y1, "y2": y2}, "grad1": g1, "grad2": g2})


def f4_conv_module():
    for causal, name in ((False, "f04_conv_module"), (True, "f04_conv_module_causal")):
        torch.manual_seed(104)
        m = ConvolutionModule(32, 15, Swish(), causal)
        randomize(m, 4)
        B, T = 3, 21
        x = torch.randn(B, T, 32, requires_grad=True)
        mask = ragged_mask([21, 10, 3], T)
        y = m(x * 1.0, mask)     # x*1.0: the module fills its (transposed view of the) input in place
        w = torch.randn_like(y)
        (y * w).sum().backward()
        save(name, **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(m, "conv."),
                      "out": {"y": y}, "grad": {**grads_of(m, "conv."), "x": x.grad}})


def f24_conv_module_cache():
    """The causal conv module fed chunk by chunk: `cache` = the previous chunk's last lorder input frames
    (convolution.py:92-104).  Stored: the second chunk's output and gradients with the first chunk's tail as cache."""
    torch.manual_seed(124)
    m = ConvolutionModule(32, 15, Swish(), True)
    randomize(m, 24)
    B, T = 3, 19
    x = torch.randn(B, T, 32, requires_grad=True)
    cache = torch.randn(B, 32, 14)                    # (batch, channel, lorder)
    mask = ragged_mask([19, 11, 4], T)
    y = m(x * 1.0, mask, cache)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("f24_conv_module_cache", **{"in": {"x": x, "mask": mask, "cache": cache, "w": w}, "sd": sd_of(m, "conv."),
                                     "out": {"y": y}, "grad": {**grads_of(m, "conv."), "x": x.grad}})


def _encoder(conformer, cmvn, seed, blocks=2, d=32):
    torch.manual_seed(seed)
    gc = None
    if cmvn:
        gc = GlobalCMVN(torch.randn(80) * 2.0, torch.rand(80) + 0.5)
    if conformer:
        enc = TransformerEncoder(80, "conv2d", "rel_pos", d, 0.0, 4, 64, "swish", True, True, 15, False,
                                 False, 64, 0.1, num_blocks=blocks, global_cmvn=gc)
    else:
        enc = TransformerEncoder(80, "conv2d", "abs_pos", d, 0.0, 4, 64, "relu", False, False, 15, False,
                                 False, 64, 0.1, num_blocks=blocks, global_cmvn=gc)
    randomize(enc, seed)
    return enc


def f5_f6_encoder():
    for conformer, cmvn, name in ((True, False, "f06_encoder_conformer"), (True, True, "f06_encoder_conformer_cmvn"),
                                  (False, False, "f06_encoder_transformer")):
        enc = _encoder(conformer, cmvn, 106)
        lens = [67, 41, 30]
        x = torch.randn(3, 67, 80, requires_grad=True)
        mask = ragged_mask(lens, 67)
        y, ymask, pos = enc(x, mask)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        # one encoder layer in isolation (F5) on the same parameters
        xl = torch.randn(3, y.shape[1], 32, requires_grad=True)
        yl, _ = enc.encoders[0](xl, ymask, pos)
        wl = torch.randn_like(yl)
        gl = torch.autograd.grad((yl * wl).sum(), [xl] + list(enc.encoders[0].parameters()))
        gl_named = {"xl": gl[0]}
        for (k, _), g in zip(enc.encoders[0].named_parameters(), gl[1:]):
            gl_named["encoder.encoders.0." + k] = g
        save(name, **{"in": {"x": x, "mask": mask, "w": w, "xl": xl, "wl": wl}, "sd": sd_of(enc, "encoder."),
                      "out": {"y": y, "mask": ymask, "pos": pos, "yl": yl},
                      "grad": {**grads_of(enc, "encoder."), "x": x.grad}, "grad_layer": gl_named})


def f16_encoder_linear_input():
    """TransformerEncoder(input_layer='linear') (encoder.py:150-151, subsampling.py:23-62), both positional encodings."""
    for kind, name in (("abs_pos", "f16_encoder_linear_abs"), ("rel_pos", "f16_encoder_linear_rel")):
        torch.manual_seed(116)
        if kind == "rel_pos":
            enc = TransformerEncoder(24, "linear", "rel_pos", 32, 0.0, 4, 64, "swish", True, True, 15, False,
                                     False, 64, 0.1, num_blocks=1)
        else:
            enc = TransformerEncoder(24, "linear", "abs_pos", 32, 0.0, 4, 64, "relu", False, False, 15, False,
                                     False, 64, 0.1, num_blocks=1)
        randomize(enc, 16)
        lens = [19, 12, 5]
        x = torch.randn(3, 19, 24, requires_grad=True)
        mask = ragged_mask(lens, 19)
        y, ymask, pos = enc(x, mask)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        save(name, **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(enc, "encoder."),
                      "out": {"y": y, "mask": ymask, "pos": pos}, "grad": {**grads_of(enc, "encoder."), "x": x.grad}})


def f18_encoder_conv2d8():
    """TransformerEncoder(input_layer='conv2d8') (encoder.py:156-157, subsampling.py:185-253), ragged lengths."""
    torch.manual_seed(118)
    enc = TransformerEncoder(80, "conv2d8", "rel_pos", 32, 0.0, 4, 64, "swish", True, True, 15, False, False, 64, 0.1, num_blocks=1)
    randomize(enc, 18)
    lens = [131, 97, 60]
    x = torch.randn(3, 131, 80, requires_grad=True)
    mask = ragged_mask(lens, 131)
    y, ymask, pos = enc(x, mask)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("f18_encoder_conv2d8", **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(enc, "encoder."),
                                   "out": {"y": y, "mask": ymask, "pos": pos}, "grad": {**grads_of(enc, "encoder."), "x": x.grad}})


def f19_activations():
    """One Conformer block (macaron FFNs + conv module use the activation) for the other entries of the reference's
    activation table (utils/common.py:160-173): tanh, hardtanh, selu, gelu."""
    for act in ("tanh", "hardtanh", "selu", "gelu"):
        torch.manual_seed(119)
        enc = TransformerEncoder(24, "linear", "rel_pos", 32, 0.0, 4, 64, act, True, True, 15, False, False, 64, 0.1, num_blocks=1)
        randomize(enc, 19)
        x = torch.randn(3, 19, 24, requires_grad=True)
        mask = ragged_mask([19, 12, 5], 19)
        y, ymask, pos = enc(x, mask)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        save(f"f19_encoder_act_{act}", **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(enc, "encoder."),
                                           "out": {"y": y}, "grad": {**grads_of(enc, "encoder."), "x": x.grad}})


def f21_encoder_conv2d6():
    """TransformerEncoder(input_layer='conv2d6') (encoder.py:154-155, subsampling.py:119-182), ragged lengths."""
    torch.manual_seed(121)
    enc = TransformerEncoder(80, "conv2d6", "abs_pos", 32, 0.0, 4, 64, "relu", False, False, 15, False, False, 64, 0.1, num_blocks=1)
    randomize(enc, 21)
    lens = [131, 97, 60]
    x = torch.randn(3, 131, 80, requires_grad=True)
    mask = ragged_mask(lens, 131)
    y, ymask, pos = enc(x, mask)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    save("f21_encoder_conv2d6", **{"in": {"x": x, "mask": mask, "w": w}, "sd": sd_of(enc, "encoder."),
                                   "out": {"y": y, "mask": ymask, "pos": pos}, "grad": {**grads_of(enc, "encoder."), "x": x.grad}})


def f17_spec_augment():
    """feature_processor.py:10-64 in CollateFunc's order (dataset.py:203-209) with a fixed python-random seed."""
    import random
    from openeat.dataset.feature_processor import _spec_augmentation, _spec_substitute
    rng = np.random.RandomState(17)
    xs = [rng.randn(t, 80).astype(np.float32) for t in (57, 40, 23)]
    sub_conf = dict(max_t=20, num_t_sub=3)
    aug_conf = dict(num_t_mask=2, num_f_mask=2, max_t=50, max_f=10)         # conf/train.yaml:51-56
    random.seed(17)
    ys = [_spec_substitute(x, **sub_conf) for x in xs]
    ys = [_spec_augmentation(y, **aug_conf) for y in ys]
    random.seed(18)
    zs = [_spec_augmentation(x, num_t_mask=3, num_f_mask=1, max_t=10, max_f=30) for x in xs]
    save("f17_spec_augment", **{"in": {f"x{i}": x for i, x in enumerate(xs)}, "out": {f"y{i}": y for i, y in enumerate(ys)},
                                "out_aug_only": {f"z{i}": z for i, z in enumerate(zs)}})


def f7_ctc():
    torch.manual_seed(107)
    V, D, B, T = 20, 16, 5, 12
    m = CTC(V, D)
    randomize(m, 7)
    hs = torch.randn(B, T, D, requires_grad=True)
    hlens = torch.tensor([12, 9, 3, 12, 7])
    # utt 2: 3 frames but 4 labels -> infeasible; utt 3: zero-length target; utt 4: repeats need blanks
    ys = torch.tensor([[3, 4, 4, 5, 1], [7, 7, 2, -1, -1], [2, 3, 4, 5, -1], [-1, -1, -1, -1, -1], [6, 6, 6, -1, -1]],
                      dtype=torch.int32)
    ylens = torch.tensor([5, 3, 4, 0, 3], dtype=torch.int32)
    logits = m.ctc_lo(hs)
    logits.retain_grad()
    logp = logits.transpose(0, 1).log_softmax(2)
    loss = m.ctc_loss(logp, ys, hlens, ylens) / B
    loss.backward()
    per_utt = torch.nn.functional.ctc_loss(logp.detach(), ys, hlens, ylens, reduction="none", zero_infinity=True)
    loss2 = m(hs.detach(), hlens, ys, ylens)
    assert torch.allclose(loss, loss2)
    save("f07_ctc", **{"in": {"hs": hs, "hlens": hlens, "ys": ys, "ylens": ylens}, "sd": sd_of(m, "ctc."),
                       "out": {"loss": loss, "per_utt": per_utt, "logits": logits},
                       "grad": {**grads_of(m, "ctc."), "hs": hs.grad, "logits": logits.grad}})


def f14_ctc_length_normalized():
    """CTC(length_normalized_loss=True): CTCLoss(reduction='mean') = mean_b(nll_b / max(len_b, 1)), then / B again
    (ctc.py:24-25,43-44)."""
    torch.manual_seed(114)
    V, D, B, T = 20, 16, 5, 12
    m = CTC(V, D, length_normalized_loss=True)
    randomize(m, 14)
    hs = torch.randn(B, T, D, requires_grad=True)
    hlens = torch.tensor([12, 9, 3, 12, 7])
    ys = torch.tensor([[3, 4, 4, 5, 1], [7, 7, 2, -1, -1], [2, 3, 4, 5, -1], [-1, -1, -1, -1, -1], [6, 6, 6, -1, -1]],
                      dtype=torch.int32)
    ylens = torch.tensor([5, 3, 4, 0, 3], dtype=torch.int32)
    logits = m.ctc_lo(hs)
    logits.retain_grad()
    loss = m.ctc_loss(logits.transpose(0, 1).log_softmax(2), ys, hlens, ylens) / B
    loss.backward()
    loss2 = m(hs.detach(), hlens, ys, ylens)
    assert torch.allclose(loss, loss2)
    save("f14_ctc_lennorm", **{"in": {"hs": hs, "hlens": hlens, "ys": ys, "ylens": ylens}, "sd": sd_of(m, "ctc."),
                               "out": {"loss": loss, "logits": logits},
                               "grad": {**grads_of(m, "ctc."), "hs": hs.grad, "logits": logits.grad}})


def f8_lsm():
    torch.manual_seed(108)
    V, B, L = 23, 4, 6
    x = torch.randn(B, L, V, requires_grad=True)
    tgt = torch.randint(0, V, (B, L))
    tgt[1, 4:] = -1
    tgt[3, 2:] = -1
    out = {}
    grads = {}
    for nl, sm, tag in ((False, 0.1, "b"), (True, 0.1, "l"), (False, 0.0, "ce")):
        crit = LabelSmoothingLoss(V, -1, sm, nl)
        loss = crit(x, tgt)
        (g,) = torch.autograd.grad(loss, x)
        out["loss_" + tag] = loss
        grads["x_" + tag] = g
    out["acc"] = rc.th_accuracy(x.view(-1, V), tgt, -1)
    save("f08_lsm", **{"in": {"x": x, "tgt": tgt}, "out": out, "grad": grads})


def f9_decoder():
    torch.manual_seed(109)
    V, D, B, L, T = 30, 32, 3, 7, 11
    dec = BiTransformerDecoder(V, D, 0.0, 4, 64, False, 64, 0.1, num_blocks=2, r_num_blocks=1)
    randomize(dec, 9)
    mem = torch.randn(B, T, D, requires_grad=True)
    mem_mask = ragged_mask([11, 8, 5], T)
    ys_lens = torch.tensor([6, 4, 2])
    ys = torch.full((B, L - 1), -1, dtype=torch.long)
    for b in range(B):
        ys[b, : ys_lens[b]] = torch.randint(1, V - 1, (int(ys_lens[b]),))
    ys_in, ys_out = rc.add_sos_eos(ys, V - 1, V - 1, -1)
    r_ys = rc.reverse_pad_list(ys, ys_lens, -1.0)
    r_in, r_out = rc.add_sos_eos(r_ys, V - 1, V - 1, -1)
    tgt_mask = (~rm.make_pad_mask(ys_lens + 1, ys_in.size(1)).unsqueeze(1)) & rm.subsequent_mask(ys_in.size(1)).unsqueeze(0)
    l_x, r_x, pre = dec(mem, mem_mask, ys_in, r_in, tgt_mask)
    wl, wr = torch.randn_like(l_x), torch.randn_like(r_x)
    ((l_x * wl).sum() + (r_x * wr).sum()).backward()
    # incremental decoding of the first row-batch with the output cache
    dec.eval()
    cache = None
    steps = []
    with torch.no_grad():
        for i in range(1, 5):
            hyp = ys_in[:, :i]
            hm = rm.subsequent_mask(i).unsqueeze(0).repeat(B, 1, 1)
            p, cache, pre_y = dec.forward_one_step(hyp, hm, mem, mem_mask, cache)
            steps.append(p)
    save("f09_decoder", **{"in": {"mem": mem, "mem_mask": mem_mask, "ys": ys, "ys_lens": ys_lens, "ys_in": ys_in,
                                  "ys_out": ys_out, "r_in": r_in, "r_out": r_out, "tgt_mask": tgt_mask, "wl": wl, "wr": wr},
                           "sd": sd_of(dec, "decoder."),
                           "out": {"l_x": l_x, "r_x": r_x, "pre": pre, "steps": torch.stack(steps)},
                           "grad": {**grads_of(dec, "decoder."), "mem": mem.grad}})


def f10_helpers():
    ys = torch.tensor([[1, 2, 3, 4, 5], [4, 5, 6, -1, -1], [7, 8, 9, -1, -1], [-1, -1, -1, -1, -1]], dtype=torch.int32)
    lens = torch.tensor([5, 3, 3, 0])
    ys_in, ys_out = rc.add_sos_eos(ys, 10, 11, -1)
    rev = rc.reverse_pad_list(ys, lens, -1.0)
    pm = rm.make_pad_mask(torch.tensor([5, 3, 2]))
    pm8 = rm.make_pad_mask(torch.tensor([5, 3, 2]), 8)
    sm = rm.subsequent_mask(5)
    paths = [[0, 0, 3, 3, 0, 3, 4, 4, 0, 0, 49, 49], [5, 5, 5], [0, 0, 0], [], [1, 0, 1, 1, 0, 2]]
    collapsed = [rc.remove_duplicates_and_blank(p) for p in paths]
    la = [rc.log_add([-1.0, -2.5, -float("inf")]), rc.log_add([-float("inf"), -float("inf")]), rc.log_add([0.3])]
    with open(os.path.join(HERE, "f10_helpers.json"), "w") as f:
        json.dump({"ys": ys.tolist(), "lens": lens.tolist(), "ys_in": ys_in.tolist(), "ys_out": ys_out.tolist(),
                   "rev": rev.tolist(), "pad_mask": pm.int().tolist(), "pad_mask8": pm8.int().tolist(),
                   "subsequent": sm.int().tolist(), "paths": paths, "collapsed": collapsed,
                   "log_add": [float(v) if v != -float("inf") else "-inf" for v in la]}, f)
    print("f10_helpers.json")


def _e2e(name, kwargs, lens, tlens, V, seed, beam=4, store_sd=True, force_token=None):
    torch.manual_seed(777)   # the reference's own seed, bin/train.py:47
    model = ASRModel(80, V, **kwargs)
    randomize(model, seed, scale=0.5)
    if force_token is not None:
        # a left decoder that always answers `force_token`, and targets that hold it at every other position:
        # th_accuracy (utils/common.py:135-157) then is a known non-trivial fraction instead of the 0.0 of random weights
        with torch.no_grad():
            model.decoder.left_decoder.output_layer.bias[force_token] += 40.0
    model.eval()
    B, T = len(lens), max(lens)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(B, T, 80, generator=g)
    flen = torch.tensor(lens, dtype=torch.int32)
    Lm = max(tlens)
    tgt = torch.full((B, Lm), -1, dtype=torch.int32)
    for b in range(B):
        tgt[b, : tlens[b]] = torch.randint(2, V - 1, (tlens[b],), generator=g, dtype=torch.int32)
        if force_token is not None:
            tgt[b, 0: tlens[b]: 2] = force_token
        feats[b, lens[b]:] = 0.0
    tlen = torch.tensor(tlens, dtype=torch.int32)
    loss, acc = model(feats, flen, tgt, tlen)
    loss.backward()
    grads = grads_of(model)
    with torch.no_grad():
        masks = ~rm.make_pad_mask(flen, T).unsqueeze(1)
        enc, enc_mask, _ = model.encoder(feats, masks)
        ctc_logits = model.ctc.ctc_lo(enc)
        greedy = model.ctc_greedy_search(feats, flen)
        nbest, _ = model._ctc_prefix_beam_search(feats[:1, : lens[0]], flen[:1], beam)
        token2char = {i: str(i) for i in range(V)}
        rw = kwargs.get("reverse_weight", 0.0)
        hyp, _, _ = model.attention_rescoring(feats[:1, : lens[0]], flen[:1], beam, ctc_weight=0.5,
                                              reverse_weight=rw, token2char=token2char)
        rec = model.recognize(feats[:2], flen[:2], beam_size=3)
    gnorm = {k: float(v.norm()) for k, v in grads.items()}
    if not store_sd:
        keep = ("ctc.ctc_lo.bias", "encoder.after_norm.weight", "encoder.embed.conv.0.weight",
                "decoder.left_decoder.after_norm.bias", "encoder.encoders.0.self_attn.linear_q.bias")
        grads = {k: v for k, v in grads.items() if k in keep}
    meta = {"kwargs": kwargs, "V": V, "beam": beam, "greedy": greedy, "grad_norm": gnorm, "seed": seed, "scale": 0.5,
            "param_order": [[k, list(p.shape)] for k, p in model.named_parameters()],
            "nbest": [[list(p), float(s)] for p, s in nbest], "rescored": list(hyp),
            "recognize": rec.tolist(), "loss": float(loss), "acc": None if acc is None else float(acc)}
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f)
    save(name, **{"in": {"feats": feats, "flen": flen, "tgt": tgt, "tlen": tlen}, "sd": sd_of(model) if store_sd else {},
                  "out": dict({"loss": loss, "enc": enc, "enc_mask": enc_mask, "ctc_logits": ctc_logits}, **({} if acc is None else {"acc": acc})),
                  "grad": grads})


def f11_f12_e2e():
    # F11 = BASELINE.json configs[0] at fixture scale: 4-enc/2-dec d=128 transformer, 4 ragged utterances
    _e2e("f11_config1_transformer",
         dict(encoder_num_blocks=4, decoder_num_blocks=2, r_decoder_num_blocks=0, d_model=128, attention_heads=4,
              linear_units=512, dropout_rate=0.0, activation_type="relu", macaron_style=False, use_cnn_module=False,
              pos_enc_layer_type="abs_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.0),
         lens=[131, 118, 99, 79], tlens=[9, 8, 6, 5], V=50, seed=11, store_sd=False)
    # F12 = tiny Conformer 2+1+1 with the bi-decoder
    _e2e("f12_tiny_conformer",
         dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True,
              cnn_module_kernel=15, pos_enc_layer_type="rel_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3),
         lens=[95, 70, 43], tlens=[7, 5, 3], V=40, seed=12)


def f15_e2e_length_normalized():
    _e2e("f15_tiny_conformer_lennorm",
         dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True,
              cnn_module_kernel=15, pos_enc_layer_type="rel_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3,
              length_normalized_loss=True),
         lens=[95, 70, 43], tlens=[7, 5, 3], V=40, seed=15)


def f20_e2e_adapters():
    """Tiny Conformer with adapters in every encoder and decoder layer (modules/adapter.py; asr_model.py:56-58)."""
    _e2e("f20_tiny_conformer_adapters",
         dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True,
              cnn_module_kernel=15, pos_enc_layer_type="rel_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3,
              encoder_use_adapter=True, decoder_use_adapter=True, down_size=16, scalar=0.25),
         lens=[95, 70, 43], tlens=[7, 5, 3], V=40, seed=20)


def f23_e2e_nonzero_accuracy():
    """Tiny Conformer end to end with an attention-decoder accuracy that is neither 0 nor 1."""
    _e2e("f23_tiny_conformer_acc",
         dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True,
              cnn_module_kernel=15, pos_enc_layer_type="rel_pos", ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3),
         lens=[95, 70, 43, 88], tlens=[7, 5, 3, 6], V=40, seed=23, force_token=5)


def f25_e2e_ctc_only():
    """ctc_weight = 1.0: the branch of asr_model.py:148-157 that skips the decoder - loss = loss_ctc, acc = None, no gradient
    reaches any decoder parameter (their entries are absent from `grad` / `grad_norm`)."""
    _e2e("f25_tiny_conformer_ctc_only",
         dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, activation_type="swish", macaron_style=True, use_cnn_module=True,
              cnn_module_kernel=15, pos_enc_layer_type="rel_pos", ctc_weight=1.0, lsm_weight=0.1, reverse_weight=0.3),
         lens=[95, 70, 43, 61], tlens=[7, 5, 3, 6], V=40, seed=25)


def f22_api_signatures():
    """Public surface of the reference's hot-path classes: for every class the methods it defines (own or inherited from
    another reference class) and their parameter names / defaults, as inspect.signature reports them.  Data only."""
    import inspect
    from openeat.modules.encoder import Encoder
    from openeat.modules.decoder import Decoder, TransformerDecoder
    from openeat.modules.encoder_layer import EncoderLayer
    from openeat.modules.decoder_layer import DecoderLayer
    from openeat.modules.positionwise_feed_forward import PositionwiseFeedForward
    from openeat.modules.subsampling import Conv2dSubsampling6, Conv2dSubsampling8, LinearNoSubsampling
    classes = [ASRModel, TransformerEncoder, Encoder, BiTransformerDecoder, TransformerDecoder, Decoder, MultiHeadedAttention,
               RelPositionMultiHeadedAttention, EncoderLayer, DecoderLayer, ConvolutionModule, PositionwiseFeedForward, CTC,
               LabelSmoothingLoss, Conv2dSubsampling4, Conv2dSubsampling6, Conv2dSubsampling8, LinearNoSubsampling,
               PositionalEncoding, RelPositionalEncoding, GlobalCMVN, Swish]
    out = {}
    for c in classes:
        methods = {}
        for name, fn in inspect.getmembers(c, predicate=inspect.isfunction):
            if not fn.__module__.startswith("openeat."):
                continue                                   # torch.nn.Module's own methods
            if name.startswith("_") and name != "__init__":
                continue
            params = []
            for pn, pv in inspect.signature(fn).parameters.items():
                d = pv.default
                params.append([pn, None if d is inspect.Parameter.empty else repr(d)])
            methods[name] = params
        out[c.__module__ + "." + c.__name__] = methods
    with open(os.path.join(HERE, "f22_api_signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("f22_api_signatures:", sum(len(m) for m in out.values()), "methods of", len(out), "classes")


def f13_misc():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sch = WarmupLR(opt, warmup_steps=25)
    lrs = []
    for _ in range(60):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sch.step()
    stats = {"mean_stat": [10.0, -4.0, 7.5], "var_stat": [60.0, 30.0, 40.0], "frame_num": 5}
    cm = os.path.join(HERE, "f13_cmvn_stats.json")
    with open(cm, "w") as f:
        json.dump(stats, f)
    mean, istd = load_cmvn(cm, True)
    rng = np.random.RandomState(13)
    feat = rng.randn(17, 5).astype(np.float32) * 3 + 1
    norm = _normalization(feat)
    save("f13_misc", **{"out": {"lrs": np.array(lrs), "cmvn_mean": mean, "cmvn_istd": istd, "feat": feat, "feat_norm": norm}})


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1:                       # regenerate only the named groups, e.g. `make_fixtures.py f14_ctc_length_normalized`
        for fn in sys.argv[1:]:
            globals()[fn]()
    else:
        f1_subsampling(); f2_relpos_mha(); f3_mha(); f4_conv_module(); f5_f6_encoder(); f7_ctc(); f8_lsm(); f9_decoder()
        f10_helpers(); f11_f12_e2e(); f13_misc(); f14_ctc_length_normalized(); f15_e2e_length_normalized()
        f16_encoder_linear_input(); f17_spec_augment(); f18_encoder_conv2d8(); f19_activations(); f20_e2e_adapters(); f21_encoder_conv2d6()
        f22_api_signatures(); f23_e2e_nonzero_accuracy(); f24_conv_module_cache(); f25_e2e_ctc_only()
