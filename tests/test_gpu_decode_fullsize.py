"""GPU: the decode path at BASELINE.json configs[3] SIZE - 64 utterances x 10 s (T' = 248 encoder frames), V = 3246, beam
10, the 12 + 3 + 3 d = 256 Conformer with the 6-layer Transformer LM - checked for equality, not only timed:

* the device prefix recursion (oe_ctc_prefix_beam) against the host recursion (oe_ctc_prefix_beam_host_batch, the bit-exact
  restatement of /root/reference/openeat/models/asr_model.py:359-396 that the goldens pin) on a CTC posterior with a
  realistic blank prior: n-best order exact, scores to 1e-9;
* attention_rescoring_batch from cached HIP graphs == eager == the reference's one-utterance algorithm
  (asr_model.py:418-534 as ASRModel.attention_rescoring) applied to every utterance, with the LM term.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from openeat_amd import hip, ops  # noqa: E402
from openeat_amd.models.asr_model import ASRModel  # noqa: E402
from openeat_amd.models.language_model import LanguageModel  # noqa: E402

DEV = "cuda"
V = 3246
CONF = dict(encoder_num_blocks=12, decoder_num_blocks=3, r_decoder_num_blocks=3, d_model=256, attention_heads=4,
            linear_units=1024, dropout_rate=0.1, input_layer="conv2d", pos_enc_layer_type="rel_pos", activation_type="swish",
            macaron_style=True, use_cnn_module=True, cnn_module_kernel=15, causal=False, ctc_weight=0.3, lsm_weight=0.1,
            reverse_weight=0.3, length_normalized_loss=False)


def test_device_prefix_beam_at_config4_size():
    B, T, beam = 64, 248, 10
    g = torch.Generator().manual_seed(64248)
    # a CTC-like posterior: ~70 % of the frames are confidently blank, the rest carry a peaked token that persists for a
    # few frames (repeats), with competing runners-up (merges and near-ties inside the beam)
    logits = torch.randn(B, T, V, generator=g) * 1.5
    blank = torch.rand(B, T, generator=g) < 0.7
    logits[:, :, 0] += torch.where(blank, torch.tensor(9.0), torch.tensor(-2.0))
    tok = torch.randint(1, V, (B, T), generator=g)
    tok[:, 1::2] = tok[:, 0::2][:, : tok[:, 1::2].shape[1]]           # every token lasts two frames
    peak = torch.where(blank, torch.tensor(4.0), torch.tensor(8.0))
    logits.scatter_add_(2, tok.unsqueeze(2), peak.unsqueeze(2))
    rival = torch.randint(1, V, (B, T), generator=g)
    logits.scatter_add_(2, rival.unsqueeze(2), (peak - 0.5).unsqueeze(2))
    lens = torch.randint(120, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[:8] = T
    top_p, top_i = ops.topk_rows(logits.to(DEV), beam, log_softmax=True)
    want = hip.ctc_prefix_beam_host_batch(top_p.cpu(), top_i.cpu(), lens.tolist(), beam)
    got = hip.ctc_prefix_beam_device(top_p, top_i, lens.to(DEV), beam)
    assert len(got) == B
    n_tok = 0
    for b in range(B):
        assert [p for p, _ in got[b]] == [p for p, _ in want[b]], (b, got[b][:2], want[b][:2])
        for (_, s1), (_, s2) in zip(got[b], want[b]):
            assert s1 == s2 or abs(s1 - s2) < 1e-9 * max(1.0, abs(s2)), (b, s1, s2)
        n_tok += len(want[b][0][0])
    assert 10 * B < n_tok < 80 * B            # hypotheses of speech-like length (~35 tokens), not the 210 of an untrained head


def test_batched_rescoring_at_config4_size_graphs_eager_and_per_utterance():
    torch.manual_seed(4)                      # bench.py's decode model: seeded init
    model = ASRModel(80, V, **CONF).to(DEV).eval()
    lm = LanguageModel(V, encoder_num_blocks=6, d_model=256, attention_heads=4, linear_units=1024).to(DEV).eval()
    with torch.no_grad():
        model.ctc.ctc_lo.bias[0] += 2.5       # a blank prior: an untrained CTC head otherwise emits ~210 tokens per utterance
    B, T, beam = 64, 998, 10
    g = torch.Generator().manual_seed(123)
    feats = torch.randn(B, T, 80, generator=g).to(DEV)
    flen = torch.full((B,), T, dtype=torch.int32, device=DEV)
    kw = dict(ctc_weight=0.5, reverse_weight=0.3, lm=lm, lm_weight=0.3)
    with torch.no_grad():
        eager = model.attention_rescoring_batch(feats, flen, beam, use_graphs=False, **kw)
        first = model.attention_rescoring_batch(feats, flen, beam, use_graphs=True, **kw)      # runs eagerly, captures
        replay = model.attention_rescoring_batch(feats, flen, beam, use_graphs=True, **kw)     # both stages from graphs
        assert first == eager and replay == eager
        recs = model._decode_graphs
        assert any(k[0] == "s1" and v is not None for k, v in recs.items()) and any(k[0] == "s2" and v is not None for k, v in recs.items())
        tok2chr = {}
        single = [list(model.attention_rescoring(feats[b:b + 1].contiguous(), flen[b:b + 1], beam, token2char=tok2chr, **kw)[0])
                  for b in range(0, B, 4)]    # every fourth utterance through the reference's B = 1 algorithm
    assert [eager[b] for b in range(0, B, 4)] == single
    lens = [len(h) for h in eager]
    assert max(lens) > 0 and len(set(lens)) > 1
