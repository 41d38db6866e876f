"""GPU, BASELINE.json config-2 sizes (B=32 x 10 s -> T=998 frames, T'=248, d=256, h=4, ff=1024, V=3246, L+1=31):
size-independent properties of the HIP kernels at the sizes the bench runs, where the CPU oracle would take minutes.

* CTC: every live frame's gradient sums to 0 over the vocabulary; padded frames and infeasible utterances are
  exactly 0; the loss is the sum of the per-utterance terms; doubling the upstream gradient doubles the result.
* GEMM (all three layouts, precision 0 and 3): linearity in each operand, agreement between the layouts
  (x W^T computed as NT and as NN on the transposed weight), weight gradient = sum of the two half-batch gradients.
* Attention: with V = 1 every output is 1 (softmax rows sum to one under the key mask); keys beyond the mask do not
  matter; the dropout-free backward of sum(out) w.r.t. q and k is 0 when V is constant.
* LayerNorm: rows have mean beta, variance gamma^2; backward gradients sum to 0 over the feature axis.
* fbank -> per-utterance normalisation: zero mean / unit variance per mel bin over each utterance's frames.
* Greedy search: output = collapse(argmax) on 32 x 248 x 3246 logits.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, T, TP, D, H, FF, V, L1 = 32, 998, 248, 256, 4, 1024, 3246, 31


def test_ctc_full_size_properties():
    from openeat_amd import hip
    torch.manual_seed(0)
    Vp = (V + 3) // 4 * 4
    logits = torch.randn(B, TP, Vp, device=DEV)
    hlens = torch.randint(150, TP + 1, (B,), dtype=torch.int32)
    hlens[0] = TP
    hlens[3] = 20                                            # fewer frames than 2L+... -> infeasible with L = 30 repeated labels
    ys = torch.randint(1, V, (B, 30), dtype=torch.int32)
    ys[3] = 7                                                # 30 identical labels need 59 frames > 20
    ylens = torch.full((B,), 30, dtype=torch.int32)
    ylens[5] = 0
    Lb = hip.lib()
    ws = torch.empty(Lb.oe_ctc_workspace_floats(B, TP, 30), device=DEV)
    nll, tot = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    hl, yd, yl = hlens.to(DEV), ys.to(DEV), ylens.to(DEV)

    def run(scale):
        dl = torch.full((B, TP, Vp), float("nan"), device=DEV)
        hip.check(Lb.oe_ctc_loss_fused(hip.ptr(logits), Vp, B, TP, V, hip.ptr(hl), hip.ptr(yd), 30, hip.ptr(yl), scale, None,
                                       hip.ptr(nll), hip.ptr(tot), hip.ptr(dl), hip.ptr(ws), hip.stream()), "ctc")
        torch.cuda.synchronize()
        return dl[:, :, :V]
    g1 = run(1.0)
    assert torch.isfinite(g1).all() and torch.isfinite(nll).all()
    assert float(nll[3]) == 0.0 and bool((g1[3] == 0).all())                       # infeasible: zero_infinity
    torch.testing.assert_close(tot[0], nll.sum(), rtol=1e-5, atol=1e-3)
    frames = torch.arange(TP, device=DEV)[None, :] < hl[:, None]
    assert bool((g1[~frames] == 0).all())                                          # padded frames exactly 0
    live = frames.clone()
    live[3] = False
    rowsum = g1.sum(-1)
    # softmax - occupancies sums to 1 - 1.  The occupancies are exp(alpha + beta - lp - ll) with |ll| ~ 2000 for random
    # logits: one fp32 ulp of the log-likelihood is 1.2e-4, so the sum is 1 only to ~1e-3 (the same holds for aten's
    # fp32 alpha/beta recursion the reference calls)
    assert float(rowsum[live].abs().max()) < 5e-3 and float(rowsum[live].abs().mean()) < 5e-4
    assert float(g1[live].abs().max()) <= 1.0 + 5e-3                               # |softmax - occupancy| <= 1, same log-domain ulp
    g2 = run(2.0)
    torch.testing.assert_close(g2, 2 * g1, rtol=1e-6, atol=1e-7)


def test_ctc_north_star_shape_against_aten_on_a_sample():
    """B=64 x 16 s (T'=398, L=48: two states per lane in the recursion, 13 float4 per lane in the row kernel): the whole
    batch through oe_ctc_loss_fused in place; four of the utterances (full length, ragged, shortest feasible, empty
    target) against aten's CPU ctc_loss + autograd; row sums and exact zeros on the rest."""
    import torch.nn.functional as F
    from openeat_amd import hip
    torch.manual_seed(5)
    Bn, Tn, Ln = 64, 398, 48
    Vp = (V + 3) // 4 * 4
    logits = torch.randn(Bn, Tn, Vp) * 1.5
    hl = torch.randint(200, Tn + 1, (Bn,), dtype=torch.int32)
    yl = torch.randint(20, Ln + 1, (Bn,), dtype=torch.int32)
    ys = torch.randint(1, V, (Bn, Ln), dtype=torch.int32)
    hl[0], yl[0] = Tn, Ln
    ys[2, :] = 9
    hl[2], yl[2] = 2 * Ln - 1, Ln                        # 48 identical labels need exactly 95 frames: one feasible path
    yl[3] = 0
    ld = logits.to(DEV)
    ws = torch.empty(hip.lib().oe_ctc_workspace_floats(Bn, Tn, Ln), device=DEV)
    nll, tot = torch.empty(Bn, device=DEV), torch.empty(1, device=DEV)
    hip.call("oe_ctc_loss_fused", ld, Vp, Bn, Tn, V, hl.to(DEV), ys.to(DEV), Ln, yl.to(DEV), 1.0, None, nll, tot, ld, ws)
    torch.cuda.synchronize()
    g = ld[:, :, :V].cpu()
    nll = nll.cpu()
    sel = [0, 1, 2, 3]
    lg = logits[sel, :, :V].clone().requires_grad_()
    per = F.ctc_loss(lg.transpose(0, 1).log_softmax(2), ys[sel].long(), hl[sel].long(), yl[sel].long(), reduction="none", zero_infinity=True)
    per.sum().backward()
    torch.testing.assert_close(nll[sel], per.detach(), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(g[sel], lg.grad, rtol=5e-3, atol=5e-5)
    assert float(nll[2]) > 0 and torch.isfinite(nll).all() and torch.isfinite(g).all()
    torch.testing.assert_close(tot.cpu()[0], nll.sum(), rtol=1e-5, atol=1e-2)
    frames = torch.arange(Tn)[None, :] < hl[:, None]
    assert bool((g[~frames] == 0).all())
    assert float(g.sum(-1)[frames].abs().max()) < 2e-2                                 # softmax - occupancies: 1 - 1 (log-domain ulp at |ll| ~ 3000)


@pytest.mark.parametrize("prec,tol", [(0, 2e-4), (3, 2e-4)])
def test_gemm_full_size_linearity_and_layout_agreement(prec, tol):
    from openeat_amd import hip
    torch.manual_seed(1)
    M = B * TP
    x1, x2 = torch.randn(M, D, device=DEV), torch.randn(M, D, device=DEV)
    w = torch.randn(FF, D, device=DEV) / math.sqrt(D)

    def nt(x):
        y = torch.empty(M, FF, device=DEV)
        hip.gemm(x, w, y, M, FF, D, lda=D, ldb=D, ldc=FF, precision=prec)
        return y
    y1, y2, y12 = nt(x1), nt(x2), nt(x1 + x2)
    scale = float(y12.abs().max())
    assert float((y12 - (y1 + y2)).abs().max()) < tol * scale * 4
    wt = w.t().contiguous()                                                       # (D, FF): x @ wt as the NN layout
    ynn = torch.empty(M, FF, device=DEV)
    hip.gemm(x1, wt, ynn, M, FF, D, lda=D, ldb=FF, ldc=FF, b_kmajor=True, precision=prec)
    assert float((ynn - y1).abs().max()) < tol * scale * 4
    # weight gradient over the full batch = sum over two halves (split-K atomics, both operands k-major)
    dy = torch.randn(M, FF, device=DEV)
    from openeat_amd.ops import _split_k

    def tn(rows):
        dw = torch.zeros(FF, D, device=DEV)
        r0, r1 = rows
        hip.gemm(dy[r0:r1], x1[r0:r1], dw, FF, D, r1 - r0, lda=FF, ldb=D, ldc=D, a_kmajor=True, b_kmajor=True,
                 split_k=_split_k(FF, D, r1 - r0), atomic_out=True, precision=prec)
        return dw
    full, a, b = tn((0, M)), tn((0, M // 2)), tn((M // 2, M))
    assert float((full - (a + b)).abs().max()) < tol * float(full.abs().max()) * 4


@pytest.mark.parametrize("prec", [0, 3])
def test_attention_full_size_rows_sum_to_one(prec):
    from openeat_amd import hip
    torch.manual_seed(2)
    dk = D // H
    q, k = torch.randn(B, TP, H, dk, device=DEV), torch.randn(B, TP, H, dk, device=DEV)
    v = torch.ones(B, TP, H, dk, device=DEV)
    lens = torch.randint(100, TP + 1, (B,))
    lens[0] = TP
    mask = (torch.arange(TP)[None, :] < lens[:, None]).to(torch.uint8).to(DEV).view(B, 1, TP).contiguous()
    out = torch.full((B, TP, H, dk), float("nan"), device=DEV)
    lse = torch.empty(B, H, TP, device=DEV)
    st = (TP * D, D)
    a = hip.attn_args(q, k, v, out, lse, B, H, TP, TP, dk, 1 / math.sqrt(dk), q_strides=st, k_strides=st, v_strides=st, o_strides=st,
                      mask=mask, mask_strides=(TP, 0), precision=prec)
    hip.attention_fwd(a)
    torch.cuda.synchronize()
    assert float((out - 1).abs().max()) < 2e-5                                     # probabilities sum to one
    # keys beyond the mask do not matter
    k2 = k.clone()
    for b_ in range(B):
        k2[b_, int(lens[b_]):] = 1e3
    v2 = torch.randn_like(v)
    o1, o2 = torch.empty_like(out), torch.empty_like(out)
    for kk, oo in ((k, o1), (k2, o2)):
        a = hip.attn_args(q, kk, v2, oo, lse, B, H, TP, TP, dk, 1 / math.sqrt(dk), q_strides=st, k_strides=st, v_strides=st,
                          o_strides=st, mask=mask, mask_strides=(TP, 0), precision=prec)
        hip.attention_fwd(a)
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    # constant V: d sum(out) / dq = dk = 0
    dq, dkk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, TP, device=DEV)
    a = hip.attn_args(q, k, v, out, lse, B, H, TP, TP, dk, 1 / math.sqrt(dk), q_strides=st, k_strides=st, v_strides=st, o_strides=st,
                      mask=mask, mask_strides=(TP, 0), d_out=torch.ones_like(out), dq=dq, dk=dkk, dv=dv, delta=delta, precision=prec)
    hip.attention_fwd(a)
    hip.attention_bwd(a)
    torch.cuda.synchronize()
    assert float(dq.abs().max()) < 1e-4 and float(dkk.abs().max()) < 1e-4
    torch.testing.assert_close(dv.sum(1), torch.full((B, H, dk), float(TP), device=DEV), rtol=1e-4, atol=1e-2)   # each query spreads weight 1


def test_layernorm_full_size_moments_and_gradient_sum():
    from openeat_amd import ops
    torch.manual_seed(3)
    x = (torch.randn(B, TP, D, device=DEV) * 3 + 1).requires_grad_()
    g = (torch.rand(D, device=DEV) + 0.5).requires_grad_()
    b = torch.randn(D, device=DEV).requires_grad_()
    y = ops.layer_norm(x, g, b, 1e-5)
    z = (y.detach() - b.detach()) / g.detach()
    assert float(z.detach().mean(-1).abs().max()) < 1e-5
    assert float((z.var(-1, unbiased=False) - 1).abs().max()) < 1e-3
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert float(x.grad.sum(-1).abs().max()) < 1e-3                               # LN output is invariant to a shift of its row
    torch.testing.assert_close(b.grad, w.sum((0, 1)), rtol=1e-4, atol=1e-2)


def test_fbank_utt_norm_full_size_moments():
    from openeat_amd.frontend import Fbank, utt_normalize_
    torch.manual_seed(4)
    wav = (torch.rand(B, 160000, device=DEV) - 0.5)
    fb = Fbank(80, device=DEV)
    n = torch.randint(100000, 160001, (B,), dtype=torch.int32)
    n[0] = 160000
    feats, frames = fb(wav, n.to(DEV))
    assert feats.shape == (B, T, 80) and int(frames[0]) == T
    assert torch.isfinite(feats).all()
    utt_normalize_(feats, frames)
    for b_ in (0, 7, 31):
        f = feats[b_, : int(frames[b_])]
        assert float(f.mean(0).abs().max()) < 1e-3
        assert float((f.std(0, unbiased=False) - 1).abs().max()) < 1e-3


def test_greedy_full_size_is_collapsed_argmax():
    from openeat_amd import hip
    torch.manual_seed(5)
    logits = torch.randn(B, TP, V, device=DEV)
    logits[:, :, 0] += 2.0                                                        # a realistic share of blanks
    hlens = torch.randint(50, TP + 1, (B,), dtype=torch.int32, device=DEV)
    fb_ = torch.empty(B, TP, dtype=torch.int32, device=DEV)
    ot, ol = torch.empty_like(fb_), torch.empty(B, dtype=torch.int32, device=DEV)
    hip.call("oe_ctc_greedy", logits, V, B, TP, V, hlens, V - 1, fb_, ot, ol)
    torch.cuda.synchronize()
    am = logits.argmax(-1).cpu()
    for b_ in range(B):
        path = am[b_].tolist()
        hl = int(hlens[b_])
        path = path[:hl] + [V - 1] * (TP - hl)                                    # padded frames read as eos (asr_model.py:321-323)
        want, prev = [], None
        for t in path:
            if t != prev and t != 0:
                want.append(t)
            prev = t
        assert ot[b_, : int(ol[b_])].tolist() == want


def test_config5_shape_trains_a_step():
    """BASELINE.json configs[4] architecture (24L Conformer d=512, h=8, ff=2048, K=15; 192.5 M parameters at V=3246) on
    16 s utterances: two engine steps run, the loss is finite and decreases with lr 1e-3 on the same batch, every
    gradient is finite.  (Kernel dispatch differs from config 2: LayerNorm with two float4 per lane, 512-channel
    convolutions, 8 heads, 2048-wide FFN tiles.)"""
    from openeat_amd.engine import TrainEngine
    from openeat_amd.models.asr_model import ASRModel
    torch.manual_seed(0)
    conf = dict(encoder_num_blocks=24, decoder_num_blocks=3, r_decoder_num_blocks=3, d_model=512, attention_heads=8, linear_units=2048,
                dropout_rate=0.1, input_layer="conv2d", pos_enc_layer_type="rel_pos", activation_type="swish", macaron_style=True,
                use_cnn_module=True, cnn_module_kernel=15, ctc_weight=0.3, lsm_weight=0.1, reverse_weight=0.3)
    model = ASRModel(80, V, **conf).to(DEV).train()
    n_par = sum(p.numel() for p in model.parameters())
    assert abs(n_par - 192.5e6) < 1.5e6, n_par
    eng = TrainEngine(model, lr=1e-3, grad_clip=5.0)
    Bs, Ts = 4, 1598
    feats = torch.randn(Bs, Ts, 80, device=DEV)
    flen = torch.tensor([1598, 1400, 1111, 803], dtype=torch.int32, device=DEV)
    tgt = torch.randint(2, V - 1, (Bs, 40), dtype=torch.int32, device=DEV)
    tlen = torch.tensor([40, 33, 25, 12], dtype=torch.int32, device=DEV)
    for b_ in range(Bs):
        tgt[b_, int(tlen[b_]):] = -1
    batch = dict(features=feats, features_length=flen, targets=tgt, targets_length=tlen)
    l0, _ = eng.step(batch)
    assert torch.isfinite(eng.arena.grad).all()
    l1, _ = eng.step(batch)
    l2, _ = eng.step(batch)
    torch.cuda.synchronize()
    assert all(math.isfinite(float(l)) for l in (l0, l1, l2)), (float(l0), float(l1), float(l2))
    assert float(l2) < float(l0)
    eng.arena.deactivate()
