"""CPU: the oracle restatement against golden vectors produced by the reference
(tests/golden/make_fixtures.py).  fp32, dropout off; tolerance 1e-5 abs /
1e-4 rel unless noted (both sides are fp32 aten kernels, summation order may
differ)."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, load_golden_json, redraw_state_dict
from oracle import asr as O
from oracle import ctc_np, fbank as FB

TOL = dict(rtol=1e-4, atol=2e-5)


def req(sd):
    return {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}


def check_grads(sd, ref, tol=TOL, skip=()):
    for k, g in ref.items():
        if k in sd and k not in skip:
            assert sd[k].grad is not None, k
            torch.testing.assert_close(sd[k].grad, g, msg=lambda m, k=k: f"{k}: {m}", **tol)


def test_f01_subsampling4():
    g = load_golden("f01_subsampling4")
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    cfg = O.Config(d_model=32, pos_enc_layer_type="rel_pos")
    y, m, pos = O.subsample4(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], **TOL)
    assert torch.equal(m, g["out"]["mask"])
    torch.testing.assert_close(pos, g["out"]["pos"], **TOL)
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(x.grad, g["grad"]["x"], **TOL)
    cfg2 = O.Config(d_model=32, pos_enc_layer_type="abs_pos")
    y2, _, _ = O.subsample4(g["sd"], cfg2, g["in"]["x"], g["in"]["mask"])
    torch.testing.assert_close(y2, g["out"]["y_abs"], **TOL)


def test_f02_relpos_mha():
    g = load_golden("f02_relpos_mha")
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y = O.relpos_mha(sd, "attn", 4, x, g["in"]["mask"], g["in"]["pos"])
    torch.testing.assert_close(y, g["out"]["y"], **TOL)
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(x.grad, g["grad"]["x"], **TOL)


def test_f03_mha_both_mask_shapes():
    g = load_golden("f03_mha")
    sd = req(g["sd"])
    q = g["in"]["q"].clone().requires_grad_()
    kv = g["in"]["kv"].clone().requires_grad_()
    y1 = O.mha(sd, "attn", 4, q, kv, kv, g["in"]["mask_k"])
    torch.testing.assert_close(y1, g["out"]["y1"], **TOL)
    (y1 * g["in"]["w1"]).sum().backward()
    check_grads(sd, g["grad1"])
    torch.testing.assert_close(q.grad, g["grad1"]["q"], **TOL)
    torch.testing.assert_close(kv.grad, g["grad1"]["kv"], **TOL)
    sd = req(g["sd"])
    s = g["in"]["s"].clone().requires_grad_()
    y2 = O.mha(sd, "attn", 4, s, s, s, g["in"]["mask_full"])
    torch.testing.assert_close(y2, g["out"]["y2"], **TOL)
    (y2 * g["in"]["w2"]).sum().backward()
    check_grads(sd, g["grad2"])
    torch.testing.assert_close(s.grad, g["grad2"]["s"], **TOL)


@pytest.mark.parametrize("name,causal", [("f04_conv_module", False), ("f04_conv_module_causal", True)])
def test_f04_conv_module(name, causal):
    g = load_golden(name)
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    cfg = O.Config(d_model=32, cnn_module_kernel=15, causal=causal)
    y = O.conv_module(sd, "conv", cfg, x, g["in"]["mask"], O._act("swish"))
    torch.testing.assert_close(y, g["out"]["y"], **TOL)
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(x.grad, g["grad"]["x"], **TOL)


def test_f24_conv_module_with_streaming_cache():
    """convolution.py:92-104: the previous chunk's last lorder input frames stand where the causal zero padding would."""
    g = load_golden("f24_conv_module_cache")
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    cfg = O.Config(d_model=32, cnn_module_kernel=15, causal=True)
    y = O.conv_module(sd, "conv", cfg, x, g["in"]["mask"], O._act("swish"), cache=g["in"]["cache"])
    torch.testing.assert_close(y, g["out"]["y"], **TOL)
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(x.grad, g["grad"]["x"], **TOL)


ENC_CFGS = {
    "f06_encoder_conformer": dict(pos_enc_layer_type="rel_pos", activation_type="swish", macaron_style=True,
                                  use_cnn_module=True, has_cmvn=False),
    "f06_encoder_conformer_cmvn": dict(pos_enc_layer_type="rel_pos", activation_type="swish", macaron_style=True,
                                       use_cnn_module=True, has_cmvn=True),
    "f06_encoder_transformer": dict(pos_enc_layer_type="abs_pos", activation_type="relu", macaron_style=False,
                                    use_cnn_module=False, has_cmvn=False),
}


LINEAR_IN = {
    "f16_encoder_linear_abs": dict(pos_enc_layer_type="abs_pos", activation_type="relu", macaron_style=False, use_cnn_module=False),
    "f16_encoder_linear_rel": dict(pos_enc_layer_type="rel_pos", activation_type="swish", macaron_style=True, use_cnn_module=True),
}


@pytest.mark.parametrize("name", list(LINEAR_IN))
def test_f16_encoder_linear_input_layer(name):
    """TransformerEncoder(input_layer='linear'): subsampling.py:23-62 in front of one encoder layer."""
    g = load_golden(name)
    cfg = O.Config(input_size=24, d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, encoder_num_blocks=1,
                   input_layer="linear", **LINEAR_IN[name])
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y, m, pos = O.encoder(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], rtol=2e-4, atol=5e-5)
    assert torch.equal(m, g["out"]["mask"])
    torch.testing.assert_close(pos, g["out"]["pos"])
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"], tol=dict(rtol=1e-3, atol=2e-4))
    torch.testing.assert_close(x.grad, g["grad"]["x"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("act", ["tanh", "hardtanh", "selu", "gelu"])
def test_f19_other_activations(act):
    g = load_golden(f"f19_encoder_act_{act}")
    cfg = O.Config(input_size=24, d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, encoder_num_blocks=1,
                   input_layer="linear", pos_enc_layer_type="rel_pos", activation_type=act, macaron_style=True, use_cnn_module=True)
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y, _, _ = O.encoder(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], rtol=2e-4, atol=5e-5)
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"], tol=dict(rtol=1e-3, atol=2e-4))
    torch.testing.assert_close(x.grad, g["grad"]["x"], rtol=1e-3, atol=2e-4)


def test_f21_encoder_conv2d6_input_layer():
    """TransformerEncoder(input_layer='conv2d6'): subsampling.py:119-182 (3x3 stride 2, then 5x5 stride 3)."""
    g = load_golden("f21_encoder_conv2d6")
    cfg = O.Config(input_size=80, d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, encoder_num_blocks=1,
                   input_layer="conv2d6", pos_enc_layer_type="abs_pos", activation_type="relu", macaron_style=False, use_cnn_module=False)
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y, m, pos = O.encoder(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], rtol=2e-4, atol=5e-5)
    assert torch.equal(m, g["out"]["mask"]) and y.shape[1] == 21
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"], tol=dict(rtol=1e-3, atol=2e-4))
    torch.testing.assert_close(x.grad, g["grad"]["x"], rtol=1e-3, atol=2e-4)


def test_f18_encoder_conv2d8_input_layer():
    """TransformerEncoder(input_layer='conv2d8'): subsampling.py:185-253 (three 3x3 stride-2 convs, 1/8 frame rate)."""
    g = load_golden("f18_encoder_conv2d8")
    cfg = O.Config(input_size=80, d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, encoder_num_blocks=1,
                   input_layer="conv2d8", pos_enc_layer_type="rel_pos", activation_type="swish", macaron_style=True, use_cnn_module=True)
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y, m, pos = O.encoder(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], rtol=2e-4, atol=5e-5)
    assert torch.equal(m, g["out"]["mask"]) and y.shape[1] == 15
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"], tol=dict(rtol=1e-3, atol=2e-4))
    torch.testing.assert_close(x.grad, g["grad"]["x"], rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("name", list(ENC_CFGS))
def test_f05_f06_encoder(name):
    g = load_golden(name)
    cfg = O.Config(d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0, encoder_num_blocks=2,
                   **ENC_CFGS[name])
    sd = req(g["sd"])
    x = g["in"]["x"].clone().requires_grad_()
    y, m, pos = O.encoder(sd, cfg, x, g["in"]["mask"])
    torch.testing.assert_close(y, g["out"]["y"], rtol=2e-4, atol=5e-5)
    assert torch.equal(m, g["out"]["mask"])
    (y * g["in"]["w"]).sum().backward()
    check_grads(sd, g["grad"], tol=dict(rtol=1e-3, atol=2e-4))
    torch.testing.assert_close(x.grad, g["grad"]["x"], rtol=1e-3, atol=2e-4)
    # single layer (F5)
    sd = req(g["sd"])
    xl = g["in"]["xl"].clone().requires_grad_()
    yl = O.encoder_layer(sd, "encoder.encoders.0", cfg, xl, g["out"]["mask"], g["out"]["pos"])
    torch.testing.assert_close(yl, g["out"]["yl"], **TOL)
    (yl * g["in"]["wl"]).sum().backward()
    torch.testing.assert_close(xl.grad, g["grad_layer"]["xl"], rtol=5e-4, atol=5e-5)
    for k, v in g["grad_layer"].items():
        if k != "xl":
            torch.testing.assert_close(sd[k].grad, v, rtol=5e-4, atol=5e-5, msg=lambda m, k=k: f"{k}: {m}")


def test_f07_ctc_loss_and_grad():
    g = load_golden("f07_ctc")
    cfg = O.Config(vocab_size=20, d_model=16)
    sd = req(g["sd"])
    hs = g["in"]["hs"].clone().requires_grad_()
    loss = O.ctc_loss(sd, cfg, hs, g["in"]["hlens"], g["in"]["ys"], g["in"]["ylens"])
    torch.testing.assert_close(loss, g["out"]["loss"], **TOL)
    loss.backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(hs.grad, g["grad"]["hs"], **TOL)
    # independent alpha/beta restatement against the reference's numbers
    nll, dlog = ctc_np.ctc_nll_and_grad(g["out"]["logits"].numpy(), g["in"]["hlens"].numpy(),
                                        g["in"]["ys"].numpy(), g["in"]["ylens"].numpy())
    np.testing.assert_allclose(nll, g["out"]["per_utt"].numpy(), rtol=1e-5, atol=1e-5)
    assert nll[2] == 0.0 and float(g["out"]["per_utt"][2]) == 0.0       # infeasible utterance -> 0
    B = hs.shape[0]
    np.testing.assert_allclose(dlog / B, g["grad"]["logits"].numpy(), rtol=1e-4, atol=2e-6)
    assert np.all(g["grad"]["logits"].numpy()[2] == 0.0)               # and no gradient
    assert np.all(g["grad"]["logits"].numpy()[1, 9:] == 0.0)           # padded frames: exactly 0


def test_f14_ctc_length_normalized():
    """CTC(length_normalized_loss=True): CTCLoss(reduction='mean') then / B (ctc.py:24-25,43-44)."""
    g = load_golden("f14_ctc_lennorm")
    cfg = O.Config(vocab_size=20, d_model=16, length_normalized_loss=True)
    sd = req(g["sd"])
    hs = g["in"]["hs"].clone().requires_grad_()
    loss = O.ctc_loss(sd, cfg, hs, g["in"]["hlens"], g["in"]["ys"], g["in"]["ylens"])
    torch.testing.assert_close(loss, g["out"]["loss"], **TOL)
    loss.backward()
    check_grads(sd, g["grad"])
    torch.testing.assert_close(hs.grad, g["grad"]["hs"], **TOL)
    # the independent alpha/beta restatement with per-utterance weights 1 / max(len, 1) and the two / B
    nll, dlog = ctc_np.ctc_nll_and_grad(g["out"]["logits"].numpy(), g["in"]["hlens"].numpy(),
                                        g["in"]["ys"].numpy(), g["in"]["ylens"].numpy())
    B = hs.shape[0]
    w = 1.0 / np.maximum(g["in"]["ylens"].numpy(), 1)
    np.testing.assert_allclose((nll * w).sum() / (B * B), float(g["out"]["loss"]), rtol=1e-5)
    np.testing.assert_allclose(dlog * w[:, None, None] / (B * B), g["grad"]["logits"].numpy(), rtol=1e-4, atol=2e-6)


def test_f08_label_smoothing_and_accuracy():
    g = load_golden("f08_lsm")
    x, tgt = g["in"]["x"], g["in"]["tgt"]
    for nl, sm, tag in ((False, 0.1, "b"), (True, 0.1, "l"), (False, 0.0, "ce")):
        cfg = O.Config(vocab_size=23, lsm_weight=sm, length_normalized_loss=nl)
        xx = x.clone().requires_grad_()
        loss = O.label_smoothing_loss(cfg, xx, tgt)
        torch.testing.assert_close(loss, g["out"]["loss_" + tag], **TOL)
        loss.backward()
        torch.testing.assert_close(xx.grad, g["grad"]["x_" + tag], **TOL)
    torch.testing.assert_close(O.token_accuracy(x.view(-1, 23), tgt, -1), g["out"]["acc"])


def test_f09_bidecoder_and_one_step():
    g = load_golden("f09_decoder")
    cfg = O.Config(vocab_size=30, d_model=32, attention_heads=4, linear_units=64, dropout_rate=0.0,
                   decoder_num_blocks=2, r_decoder_num_blocks=1)
    i = g["in"]
    ys_in, ys_out = O.with_sos_eos(i["ys"], 29, 29, -1)
    assert torch.equal(ys_in, i["ys_in"]) and torch.equal(ys_out, i["ys_out"])
    r_in, r_out = O.with_sos_eos(O.reversed_targets(i["ys"], i["ys_lens"], -1), 29, 29, -1)
    assert torch.equal(r_in, i["r_in"]) and torch.equal(r_out, i["r_out"])
    sd = req(g["sd"])
    mem = i["mem"].clone().requires_grad_()
    l_x, r_x, pre = O.bi_decoder(sd, cfg, mem, i["mem_mask"], ys_in, r_in, i["tgt_mask"])
    torch.testing.assert_close(l_x, g["out"]["l_x"], **TOL)
    torch.testing.assert_close(r_x, g["out"]["r_x"], **TOL)
    torch.testing.assert_close(pre, g["out"]["pre"], **TOL)
    ((l_x * i["wl"]).sum() + (r_x * i["wr"]).sum()).backward()
    check_grads(sd, g["grad"], tol=dict(rtol=5e-4, atol=5e-5))
    torch.testing.assert_close(mem.grad, g["grad"]["mem"], rtol=5e-4, atol=5e-5)
    cache = None
    with torch.no_grad():
        for step in range(1, 5):
            hm = O.causal_mask(step).unsqueeze(0).repeat(3, 1, 1)
            p, cache, _ = O.decoder_one_step(g["sd"], cfg, ys_in[:, :step], hm, i["mem"], i["mem_mask"], cache)
            torch.testing.assert_close(p, g["out"]["steps"][step - 1], **TOL)


def test_f10_helper_tables():
    j = load_golden_json("f10_helpers")
    ys = torch.tensor(j["ys"], dtype=torch.int32)
    ys_in, ys_out = O.with_sos_eos(ys, 10, 11, -1)
    assert ys_in.tolist() == j["ys_in"] and ys_out.tolist() == j["ys_out"]
    assert O.reversed_targets(ys, torch.tensor(j["lens"]), -1).tolist() == j["rev"]
    assert O.pad_mask(torch.tensor([5, 3, 2])).int().tolist() == j["pad_mask"]
    assert O.pad_mask(torch.tensor([5, 3, 2]), 8).int().tolist() == j["pad_mask8"]
    assert O.causal_mask(5).int().tolist() == j["subsequent"]
    assert [O.collapse_ctc_path(p) for p in j["paths"]] == j["collapsed"]
    la = [O.log_sum_exp([-1.0, -2.5, -float("inf")]), O.log_sum_exp([-float("inf")] * 2), O.log_sum_exp([0.3])]
    for a, b in zip(la, j["log_add"]):
        b = -float("inf") if b == "-inf" else b
        assert a == b or abs(a - b) < 1e-12


E2E = {"f11_config1_transformer": True, "f12_tiny_conformer": False, "f15_tiny_conformer_lennorm": False,
       "f20_tiny_conformer_adapters": False, "f23_tiny_conformer_acc": False, "f25_tiny_conformer_ctc_only": False}


@pytest.mark.parametrize("name", list(E2E))
def test_f11_f12_end_to_end(name):
    g = load_golden(name)
    meta = load_golden_json(name)
    sd = redraw_state_dict(meta) if E2E[name] else g["sd"]
    cfg = O.Config(input_size=80, vocab_size=meta["V"], **meta["kwargs"])
    sdr = req(sd)
    i = g["in"]
    loss, acc = O.forward(sdr, cfg, i["feats"], i["flen"], i["tgt"], i["tlen"])
    torch.testing.assert_close(loss, g["out"]["loss"], rtol=2e-4, atol=2e-4)
    if meta["acc"] is None:                                 # ctc_weight = 1.0: asr_model.py:148-157 skips the decoder
        assert acc is None
    else:
        torch.testing.assert_close(acc, g["out"]["acc"])
    loss.backward()
    for k, n in meta["grad_norm"].items():
        got = float(sdr[k].grad.norm())
        assert abs(got - n) <= 2e-3 * max(1.0, abs(n)), (k, got, n)
    check_grads(sdr, g["grad"], tol=dict(rtol=2e-3, atol=2e-4))
    with torch.no_grad():
        masks = (~O.pad_mask(i["flen"], i["feats"].size(1))).unsqueeze(1)
        enc, enc_mask, _ = O.encoder(sd, cfg, i["feats"], masks)
        torch.testing.assert_close(enc, g["out"]["enc"], rtol=1e-3, atol=2e-4)
        assert torch.equal(enc_mask, g["out"]["enc_mask"])
        torch.testing.assert_close(O.ctc_logits(sd, enc), g["out"]["ctc_logits"], rtol=1e-3, atol=5e-4)
        assert O.ctc_greedy_search(sd, cfg, i["feats"], i["flen"]) == meta["greedy"]      # bit-exact ids
        n0 = int(i["flen"][0])
        nbest, _ = O.ctc_prefix_beam_search(sd, cfg, i["feats"][:1, :n0], i["flen"][:1], meta["beam"])
        assert [list(p) for p, _ in nbest] == [p for p, _ in meta["nbest"]]
        for (_, s), (_, r) in zip(nbest, meta["nbest"]):
            assert abs(s - r) < 1e-3 * max(1.0, abs(r))
        best, _, _ = O.attention_rescoring(sd, cfg, i["feats"][:1, :n0], i["flen"][:1], meta["beam"], 0.5,
                                           meta["kwargs"].get("reverse_weight", 0.0))
        assert list(best) == meta["rescored"]


def test_f13_cmvn_and_utt_norm():
    g = load_golden("f13_misc")
    torch.testing.assert_close(FB.utt_normalize(g["out"]["feat"]), g["out"]["feat_norm"], rtol=1e-6, atol=1e-6)


def test_fbank_cross_check_transformers():
    """Sanity anchor only (fbank parity is UNPINNED, see oracle/fbank.py): an
    independent kaldi-compatible implementation shipped with transformers."""
    au = pytest.importorskip("transformers.audio_utils")
    torch.manual_seed(5)
    wav = (torch.rand(16000) - 0.5) * 0.8
    mine = FB.fbank(wav)
    mel = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=80, min_frequency=20, max_frequency=8000,
                             sampling_rate=16000, norm=None, mel_scale="kaldi", triangularize_in_mel_space=True)
    win = au.window_function(400, "povey", periodic=False)
    theirs = au.spectrogram((wav.numpy() * 32768.0).astype(np.float64), win, frame_length=400, hop_length=160,
                            fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=mel,
                            log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T
    assert mine.shape == (98, 80) and theirs.shape == (98, 80)
    np.testing.assert_allclose(mine.numpy(), theirs, rtol=2e-4, atol=2e-3)


def test_f17_spec_augment_and_substitute():
    """oracle/augment.py against the reference's own output for the same python-random seed."""
    import random
    from oracle import augment as A
    g = load_golden("f17_spec_augment")
    xs = [g["in"][f"x{i}"].numpy() for i in range(3)]
    random.seed(17)
    ys = A.collate_augment(xs, dict(max_t=20, num_t_sub=3), dict(num_t_mask=2, num_f_mask=2, max_t=50, max_f=10))
    for i, y in enumerate(ys):
        assert np.array_equal(y, g["out"][f"y{i}"].numpy())
    random.seed(18)
    zs = A.collate_augment(xs, None, dict(num_t_mask=3, num_f_mask=1, max_t=10, max_f=30))
    for i, z in enumerate(zs):
        assert np.array_equal(z, g["out_aug_only"][f"z{i}"].numpy())
    assert any((y == 0).all(axis=1).any() for y in ys)              # some frame really is masked
