"""Body of __graft_entry__.smoke(): one tiny invocation of the hot path on cuda:0 - fbank ->
Conformer encoder -> CTC + bi-decoder loss -> backward -> greedy decode - checked against the CPU oracle."""
import torch


def run():
    from openeat_amd import hip
    from openeat_amd.frontend import Fbank, utt_normalize_
    from openeat_amd.models.asr_model import ASRModel
    from oracle import asr as O
    from oracle import fbank as FB
    hip.lib()
    dev = "cuda:0"
    kw = dict(encoder_num_blocks=2, decoder_num_blocks=1, r_decoder_num_blocks=1, d_model=32, attention_heads=4,
              linear_units=64, dropout_rate=0.0, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3)
    torch.manual_seed(0)
    model = ASRModel(80, 40, **kw)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    wav = (torch.rand(3, 9000) - 0.5) * 0.8
    ns = torch.tensor([9000, 7000, 5200])
    tgt = torch.randint(2, 39, (3, 5), dtype=torch.int32)
    tlen = torch.tensor([5, 4, 2], dtype=torch.int32)
    for b in range(3):
        tgt[b, int(tlen[b]):] = -1
        wav[b, int(ns[b]):] = 0
    feats, nfr = Fbank(80, device=dev)(wav.to(dev), ns.to(dev))
    utt_normalize_(feats, nfr)
    loss, acc = model(feats, nfr, tgt.to(dev), tlen.to(dev))
    loss.backward()
    hyp = model.ctc_greedy_search(feats, nfr)
    torch.cuda.synchronize()
    # oracle on the CPU
    T = feats.shape[1]
    ref_feats = torch.zeros(3, T, 80)
    for b in range(3):
        f = FB.utt_normalize(FB.fbank(wav[b, : int(ns[b])]))
        ref_feats[b, : f.shape[0]] = f
    cfg = O.Config(input_size=80, vocab_size=40, **kw)
    rl, racc = O.forward(sd, cfg, ref_feats, nfr.cpu(), tgt, tlen)
    rl.backward()
    torch.testing.assert_close(feats.cpu(), ref_feats, rtol=1e-3, atol=5e-3)
    torch.testing.assert_close(loss.cpu(), rl.detach(), rtol=1e-3, atol=1e-3)
    g = model.ctc.ctc_lo.weight.grad.cpu()
    torch.testing.assert_close(g, sd["ctc.ctc_lo.weight"].grad, rtol=2e-2, atol=2e-3 * float(g.abs().max()))
    assert hyp == O.ctc_greedy_search({k: v.detach() for k, v in sd.items()}, cfg, ref_feats, nfr.cpu())
    print(f"smoke OK: loss {float(loss):.4f} (oracle {float(rl):.4f}), acc {float(acc):.3f}, greedy {hyp}")
