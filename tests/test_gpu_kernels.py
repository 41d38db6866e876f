"""GPU: each HIP kernel, called through the C ABI, against the CPU oracle /
a plain fp32 torch restatement of the same op on the same seeded inputs.

Tolerances (fp32 kernels; the MFMA GEMM is exact-fp32 products with a
different summation order than aten): rtol 1e-4 / atol scaled to the data.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openeat_amd import hip  # noqa: E402
from oracle import ctc_np  # noqa: E402

DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------ GEMM -----
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 70), (1000, 256, 1024), (77, 3246, 256), (4096, 1024, 256)])
def test_gemm_nt_bias(M, N, K):
    torch.manual_seed(0)
    x, w, b = torch.randn(M, K), torch.randn(N, K), torch.randn(N)
    ref = x.double() @ w.double().T + b.double()
    xd, wd, bd = cu(x), cu(w), cu(b)
    out = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(xd, wd, out, M, N, K, lda=K, ldb=K, ldc=N, bias=bd)
    sync()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-4 * math.sqrt(K))


def test_gemm_padded_ld_and_unaligned_k():
    """ldc with padding columns (V=3246 stored with ld 3248) and K not a multiple of 4/16."""
    torch.manual_seed(1)
    M, N, K, ldc = 130, 3246, 50, 3248
    x, w = torch.randn(M, K), torch.randn(N, K)
    out = torch.zeros(M, ldc, device=DEV)
    hip.gemm(cu(x), cu(w), out, M, N, K, lda=K, ldb=K, ldc=ldc)
    sync()
    torch.testing.assert_close(out[:, :N].cpu(), x @ w.T, rtol=1e-4, atol=1e-3)
    assert torch.all(out[:, N:] == 0)
    # dgrad through the padded tensor: dx[M,K] = dy[M,N(ld 3248)] @ w[N,K]
    dy = torch.zeros(M, ldc)
    dy[:, :N] = torch.randn(M, N)
    dx = torch.empty(M, K, device=DEV)
    hip.gemm(cu(dy), cu(w), dx, M, K, N, lda=ldc, ldb=K, ldc=K, b_kmajor=True)
    sync()
    torch.testing.assert_close(dx.cpu(), dy[:, :N] @ w, rtol=1e-4, atol=1e-2)


def test_gemm_nn_and_tn_splitk():
    torch.manual_seed(2)
    M, N, K = 900, 256, 1024     # y = x W^T ; x (M,K), W (N,K)
    x, w, dy = torch.randn(M, K), torch.randn(N, K), torch.randn(M, N)
    dx = torch.empty(M, K, device=DEV)
    hip.gemm(cu(dy), cu(w), dx, M, K, N, lda=N, ldb=K, ldc=K, b_kmajor=True)
    dw = torch.zeros(N, K, device=DEV)
    hip.gemm(cu(dy), cu(x), dw, N, K, M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=5, atomic_out=True)
    sync()
    torch.testing.assert_close(dx.cpu(), dy @ w, rtol=1e-4, atol=2e-3)
    torch.testing.assert_close(dw.cpu(), dy.T @ x, rtol=1e-4, atol=3e-3)
    # accumulate into an existing gradient (beta=1 semantics used for shared blocks)
    dw2 = torch.ones(N, K, device=DEV)
    hip.gemm(cu(dy), cu(x), dw2, N, K, M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, accumulate=True)
    sync()
    torch.testing.assert_close(dw2.cpu(), dy.T @ x + 1.0, rtol=1e-4, atol=3e-3)


@pytest.mark.parametrize("act", ["relu", "swish"])
def test_gemm_epilogue_activation_residual_rowmask(act):
    torch.manual_seed(3)
    M, N, K = 333, 192, 96
    x, w, b, res = torch.randn(M, K), torch.randn(N, K) * 0.2, torch.randn(N), torch.randn(M, N)
    rowmask = (torch.rand(M) > 0.3).to(torch.uint8)
    pre_ref = x @ w.T + b
    a_ref = F.relu(pre_ref) if act == "relu" else pre_ref * torch.sigmoid(pre_ref)
    out = torch.empty(M, N, device=DEV)
    pre = torch.empty(M, N, device=DEV)
    hip.gemm(cu(x), cu(w), out, M, N, K, lda=K, ldb=K, ldc=N, bias=cu(b), act=hip.ACT[act], preact_out=pre, ld_aux=N,
             rowmask=cu(rowmask), residual=cu(res), ldr=N, beta=0.5)
    sync()
    torch.testing.assert_close(pre.cpu(), pre_ref, rtol=1e-4, atol=1e-4)
    ref = res + 0.5 * a_ref * rowmask[:, None].float()
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4)
    # act-grad epilogue: dh = (dy @ W2) * act'(pre)
    dy, w2 = torch.randn(M, K), torch.randn(K, N) * 0.2            # here N plays the hidden dim
    prer = pre_ref.clone().requires_grad_()
    (F.relu(prer) if act == "relu" else prer * torch.sigmoid(prer)).backward(dy @ w2)
    dh = torch.empty(M, N, device=DEV)
    hip.gemm(cu(dy), cu(w2), dh, M, N, K, lda=K, ldb=N, ldc=N, b_kmajor=True, act=hip.ACT[act], actgrad_in=pre, ld_aux=N)
    sync()
    torch.testing.assert_close(dh.cpu(), prer.grad, rtol=1e-4, atol=1e-4)


def test_gemm_dropout_epilogue_statistics_and_determinism():
    M, N, K = 512, 256, 32
    x, w = torch.ones(M, K), torch.ones(N, K) / K
    a = torch.empty(M, N, device=DEV)
    b = torch.empty(M, N, device=DEV)
    hip.gemm(cu(x), cu(w), a, M, N, K, lda=K, ldb=K, ldc=N, drop_p=0.1, seed=1234)
    hip.gemm(cu(x), cu(w), b, M, N, K, lda=K, ldb=K, ldc=N, drop_p=0.1, seed=1234)
    c = torch.empty(M, N, device=DEV)
    hip.gemm(cu(x), cu(w), c, M, N, K, lda=K, ldb=K, ldc=N, drop_p=0.1, seed=99)
    sync()
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = (a != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    torch.testing.assert_close(a[a != 0], torch.full_like(a[a != 0], 1 / 0.9), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("prec", [0, 3])
@pytest.mark.parametrize("M,N,K", [(512, 256, 32), (300, 200, 64), (130, 36, 32), (992, 1024, 256)])
def test_dropout_mask_is_the_same_in_every_consumer(M, N, K, prec):
    """The GEMM epilogue (interior blocks share Philox calls between lane pairs, edge blocks go element by element,
    N % 8 != 0 likewise) and the stand-alone dropout kernel (eight elements per thread) must realise the same mask
    for the same seed: forward fuses it into the producing GEMM, backward applies it with oe_dropout_scale."""
    torch.manual_seed(6)
    x, w = cu(torch.randn(M, K)), cu(torch.randn(N, K))
    plain = torch.empty(M, N, device=DEV)
    fused = torch.empty(M, N, device=DEV)
    hip.gemm(x, w, plain, M, N, K, lda=K, ldb=K, ldc=N, precision=prec)
    hip.gemm(x, w, fused, M, N, K, lda=K, ldb=K, ldc=N, drop_p=0.3, seed=4242, precision=prec)
    after = torch.empty_like(plain)
    hip.call("oe_dropout_scale", plain, plain.numel(), N, 1.0, 0.3, 4242, None, None, after)
    sync()
    assert torch.equal(fused == 0, after == 0)
    torch.testing.assert_close(fused, after, rtol=1e-6, atol=1e-6)
    assert abs((fused != 0).float().mean().item() - 0.7) < 0.02


def test_gemm_conv2_implicit_forward_and_wgrad():
    """Conv2d(C,C,3,stride 2) over NHWC as an implicit GEMM (subsampling.py:79)."""
    torch.manual_seed(4)
    B, T1, F1, Cc = 3, 21, 11, 32
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    x = torch.randn(B, Cc, T1, F1)
    w = torch.randn(Cc, Cc, 3, 3) * 0.1
    bias = torch.randn(Cc)
    ref = F.conv2d(x, w, bias, stride=2)                                   # (B,C,T2,F2)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()                            # (B,T1,F1,C)
    w_g = w.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc).contiguous()           # [co][kh][kw][ci]
    M = B * T2 * F2
    out = torch.empty(M, Cc, device=DEV)
    conv = (T1, F1, T2, F2, Cc)
    hip.gemm(cu(x_nhwc), cu(w_g), out, M, Cc, 9 * Cc, lda=0, ldb=9 * Cc, ldc=Cc, bias=cu(bias), conv=conv,
             conv_gather=hip.GATHER_A)
    sync()
    got = out.cpu().view(B, T2, F2, Cc).permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    # wgrad: dW_g[co][k] = sum_m dy[m][co] * col[m][k]
    dy = torch.randn(B, Cc, T2, F2)
    xr = x.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    F.conv2d(xr, wr, None, stride=2).backward(dy)
    dy_m = dy.permute(0, 2, 3, 1).reshape(M, Cc).contiguous()
    dwg = torch.zeros(Cc, 9 * Cc, device=DEV)
    hip.gemm(cu(dy_m), cu(x_nhwc), dwg, Cc, 9 * Cc, M, lda=Cc, ldb=0, ldc=9 * Cc, a_kmajor=True, b_kmajor=True,
             split_k=3, atomic_out=True, conv=conv, conv_gather=hip.GATHER_B)
    sync()
    got_dw = dwg.cpu().view(Cc, 3, 3, Cc).permute(0, 3, 1, 2)
    torch.testing.assert_close(got_dw, wr.grad, rtol=1e-4, atol=1e-3)


def test_colsum():
    torch.manual_seed(5)
    x = torch.randn(1000, 300)
    out = torch.full((300,), 7.0, device=DEV)
    xd = cu(x)
    hip.check(hip.lib().oe_colsum_f32(hip.ptr(xd), 300, 1000, 300, 0.5, None, hip.ptr(out), 0, hip.stream()), "colsum")
    sync()
    torch.testing.assert_close(out.cpu(), 0.5 * x.sum(0), rtol=1e-4, atol=1e-3)


# -------------------------------------------------------------- LayerNorm ----
@pytest.mark.parametrize("rows,d,eps", [(1000, 256, 1e-12), (37, 32, 1e-5), (64, 512, 1e-5)])
def test_layernorm_fwd_bwd(rows, d, eps):
    torch.manual_seed(6)
    x = (torch.randn(rows, d) * 2 + 0.3).requires_grad_()
    g = (torch.randn(d) * 0.5 + 1).requires_grad_()
    b = torch.randn(d).requires_grad_()
    dy = torch.randn(rows, d)
    mask = (torch.rand(rows) > 0.25).to(torch.uint8)
    y_ref = F.layer_norm(x, (d,), g, b, eps) * mask[:, None].float()
    y_ref.backward(dy)
    xd, gd, bd, dyd, md = cu(x.detach()), cu(g.detach()), cu(b.detach()), cu(dy), cu(mask)
    y = torch.empty(rows, d, device=DEV)
    stats = torch.empty(rows, 2, device=DEV)
    L = hip.lib()
    hip.check(L.oe_layernorm_fwd(hip.ptr(xd), hip.ptr(gd), hip.ptr(bd), eps, rows, d, hip.ptr(md), 0, hip.ptr(y),
                                 hip.ptr(stats), hip.stream()), "ln_fwd")
    dx = torch.empty(rows, d, device=DEV)
    dg = torch.zeros(d, device=DEV)
    db = torch.zeros(d, device=DEV)
    add = torch.randn(rows, d)
    addd = cu(add)
    ws = torch.empty(L.oe_layernorm_bwd_workspace_floats(rows, d), device=DEV)
    hip.check(L.oe_layernorm_bwd(hip.ptr(dyd), hip.ptr(xd), hip.ptr(gd), hip.ptr(bd), 0, hip.ptr(stats), rows, d, hip.ptr(md),
                                 hip.ptr(addd), hip.ptr(dx), hip.ptr(dg), hip.ptr(db), hip.ptr(ws), hip.stream()), "ln_bwd")
    sync()
    dx = dx - addd
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx.cpu(), x.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dg.cpu(), g.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("rows,d,p,alpha,masked", [(50, 256, 0.1, 0.5, False), (37, 64, 0.25, 1.0, True), (16, 512, 0.0, 0.5, False),
                                                    (7936, 256, 0.1, 1.0, True)])
def test_layernorm_backward_second_output_is_dropout_scale_of_dx(rows, d, p, alpha, masked):
    """oe_layernorm_bwd_dx_drop: dx identical to oe_layernorm_bwd_dx, and gout BIT-identical to oe_dropout_scale(dx, alpha, p,
    seed, rowmask) - the mask a block's forward applied in its GEMM epilogue is the one its backward gets back."""
    torch.manual_seed(21)
    L = hip.lib()
    x, dy, add = (torch.randn(rows, d, device=DEV) for _ in range(3))
    gamma, beta = torch.randn(d, device=DEV), torch.randn(d, device=DEV)
    y, stats = torch.empty_like(x), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_fwd", x, gamma, beta, 1e-5, rows, d, None, 0, y, stats)
    rm = (torch.rand(rows, device=DEV) > 0.3).to(torch.uint8) if masked else None
    ws = torch.empty(L.oe_layernorm_bwd_workspace_floats(rows, d), device=DEV)
    dx0, dx1, g1 = (torch.full((rows, d), float("nan"), device=DEV) for _ in range(3))
    hip.call("oe_layernorm_bwd_dx", dy, x, gamma, beta, 0, stats, rows, d, None, add, dx0, ws)
    seed = 0x1234567
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)                  # device step counter mixed into the seed
    hip.call("oe_layernorm_bwd_dx_drop", dy, x, gamma, beta, 0, stats, rows, d, None, add, dx1, g1, alpha, p, seed, ctr, rm, ws)
    want = torch.empty_like(dx0)
    hip.call("oe_dropout_scale", dx0, dx0.numel(), d, alpha, p, seed, ctr, rm, want)
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    assert torch.equal(g1, want)
    if p > 0:
        kept = (g1 != 0).float().mean().item() / (float(rm.float().mean()) if masked else 1.0)     # share among the unmasked rows
        assert abs(kept - (1 - p)) < 0.05


# -------------------------------------------------------------------- CTC ----
@pytest.fixture(params=[(0, 4), (4, 4), (2, 4), (2, 3), (2, 8)],
                ids=["three-launch-form", "overlapped-form", "pipelined-4-chunks", "pipelined-3-chunks", "pipelined-8-chunks"])
def ctc_form(request):
    """The CTC tests run in the three-launch form (rows, alpha/beta, labels one after the other), with the overlapped form forced
    (row statistics, then mixed launches: recursion blocks beside dense-gradient blocks, fix-up blocks beside the rest of the dense
    blocks; ctc.hip) and with the pipelined form forced (the recursion in time chunks on two internal streams, resumed from the
    columns the previous chunk left: oe_ctc_config) - same numbers every way."""
    mode, chunks = request.param
    hip.lib().oe_ctc_config(mode, chunks)
    yield request.param
    hip.lib().oe_ctc_config(0, 4)


def run_ctc(logits, hlens, ys, ylens, ldv=None, scale=1.0, inplace=False, utt_weight=None, want_grad=True):
    B, T, V = logits.shape
    ldv = ldv or V
    buf = torch.zeros(B, T, ldv, device=DEV)
    buf[:, :, :V] = logits.to(DEV)
    Lmax = ys.shape[1]
    hl, yl, yd = cu(hlens.int()), cu(ylens.int()), cu(ys.int())
    L = hip.lib()
    ws = torch.empty(L.oe_ctc_workspace_floats(B, T, Lmax), device=DEV)
    nll = torch.empty(B, device=DEV)
    tot = torch.empty(1, device=DEV)
    dl = buf if inplace else torch.full((B, T, ldv), float("nan"), device=DEV)
    if not want_grad:
        dl = None
    uw = None if utt_weight is None else cu(utt_weight.float())
    hip.check(L.oe_ctc_loss_fused(hip.ptr(buf), ldv, B, T, V, hip.ptr(hl), hip.ptr(yd), Lmax, hip.ptr(yl), scale,
                                  hip.ptr(uw), hip.ptr(nll), hip.ptr(tot), hip.ptr(dl), hip.ptr(ws), hip.stream()), "ctc")
    sync()
    return nll.cpu(), tot.cpu(), (dl[:, :, :V].cpu() if want_grad else None)


def test_ctc_golden_f07(ctc_form):
    """The reference's own numbers (fixture F7): infeasible utterance, empty target, repeats."""
    from conftest import load_golden
    g = load_golden("f07_ctc")
    logits, hl, ys, yl = g["out"]["logits"], g["in"]["hlens"], g["in"]["ys"], g["in"]["ylens"]
    B = logits.shape[0]
    nll, tot, dl = run_ctc(logits, hl, ys, yl, ldv=20, scale=1.0 / B)
    torch.testing.assert_close(nll, g["out"]["per_utt"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(tot[0] / B, g["out"]["loss"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dl, g["grad"]["logits"], rtol=1e-3, atol=1e-5)
    assert torch.all(dl[2] == 0) and torch.all(dl[1, 9:] == 0)       # exact zeros, as the reference


@pytest.mark.parametrize("B,T,V,Lmax,ldv", [(6, 50, 37, 9, 40), (4, 120, 3246, 40, 3248), (3, 90, 501, 70, 504),
                                             (2, 300, 100, 140, 100)])
def test_ctc_vs_oracle(B, T, V, Lmax, ldv, ctc_form):
    torch.manual_seed(7)
    logits = torch.randn(B, T, V) * 2
    hl = torch.randint(T // 2, T + 1, (B,))
    hl[0] = T
    yl = torch.randint(1, Lmax + 1, (B,))
    yl[-1] = Lmax
    ys = torch.randint(1, V, (B, Lmax))
    lg = logits.clone().requires_grad_()
    logp = lg.transpose(0, 1).log_softmax(2)
    per = F.ctc_loss(logp, ys, hl, yl, reduction="none", zero_infinity=True)
    per.sum().backward()
    nll, tot, dl = run_ctc(logits, hl, ys, yl, ldv=ldv, inplace=True)
    torch.testing.assert_close(nll, per.detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(tot[0], per.sum().detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dl, lg.grad, rtol=2e-3, atol=2e-5)
    if B * T * V < 50000:   # third opinion: float64 loop-level alpha/beta
        n2, g2 = ctc_np.ctc_nll_and_grad(logits.numpy(), hl.numpy(), ys.numpy(), yl.numpy())
        np.testing.assert_allclose(nll.numpy(), n2, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(dl.numpy(), g2, rtol=2e-3, atol=2e-5)


# row kernel variants: rows kept in registers (4 / 8 / 16 / 32 float4 per lane) or read twice (rows that are not whole
# aligned float4s, or wider than 8192); utterance lengths at the edges of the recursion's 32-frame prefetch chunks
@pytest.mark.parametrize("V,ldv,inplace", [(37, 37, True), (37, 39, False), (37, 40, True), (1500, 1500, False), (3246, 3246, True),
                                           (3246, 3248, False), (5000, 5000, True), (9000, 9000, False)])
def test_ctc_row_variants_and_chunk_edges(V, ldv, inplace, ctc_form):
    torch.manual_seed(11)
    T, Lmax = 70, 5
    hl = torch.tensor([70, 1, 2, 32, 33, 34, 3, 65, 64])
    yl = torch.tensor([5, 1, 1, 5, 3, 0, 5, 4, 2])          # utterance 6: five labels in three frames -> infeasible
    B = hl.numel()
    ys = torch.randint(1, V, (B, Lmax))
    ys[0, 1] = ys[0, 0]                                     # a repeat (needs the blank between)
    logits = torch.randn(B, T, V) * 2
    w = torch.rand(B) + 0.5
    lg = logits.clone().requires_grad_()
    per = F.ctc_loss(lg.transpose(0, 1).log_softmax(2), ys, hl, yl, reduction="none", zero_infinity=True)
    (per * w).sum().backward()
    nll0, tot0, none = run_ctc(logits, hl, ys, yl, ldv=ldv, utt_weight=w, want_grad=False)      # loss only
    assert none is None
    nll, tot, dl = run_ctc(logits, hl, ys, yl, ldv=ldv, scale=0.25, inplace=inplace, utt_weight=w)
    assert torch.equal(nll, nll0) and torch.equal(tot, tot0)
    torch.testing.assert_close(nll, per.detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(tot[0], (per * w).sum().detach(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(dl, 0.25 * lg.grad, rtol=2e-3, atol=1e-5)
    assert float(nll[6]) == 0.0 and bool((dl[6] == 0).all())                                   # zero_infinity
    for b in range(B):
        assert bool((dl[b, int(hl[b]):] == 0).all())                                           # padded frames exactly 0


def test_ctc_pipelined_form_is_bit_identical_to_the_sequential_one():
    """Same arithmetic, other launch structure: losses and gradients equal bit for bit (ragged lengths that end inside every
    chunk, an infeasible utterance, an empty target, repeats)."""
    torch.manual_seed(12)
    B, T, V, Lmax = 9, 131, 200, 12
    hl = torch.tensor([131, 1, 2, 33, 66, 98, 100, 3, 131])
    yl = torch.tensor([12, 1, 1, 7, 0, 12, 5, 6, 3])          # utterance 7: six labels in three frames -> infeasible
    ys = torch.randint(1, V, (B, Lmax))
    ys[0, 1] = ys[0, 0]
    logits = torch.randn(B, T, V) * 2
    outs = []
    for mode, chunks in ((0, 4), (2, 4), (2, 5)):
        hip.lib().oe_ctc_config(mode, chunks)
        try:
            outs.append(run_ctc(logits, hl, ys, yl, scale=0.5))
        finally:
            hip.lib().oe_ctc_config(0, 4)
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_ctc_greedy_matches_topk_and_collapse():
    from oracle import asr as O
    torch.manual_seed(8)
    B, T, V = 5, 61, 50
    logits = torch.randn(B, T, V)
    logits[0, 3, 7] = logits[0, 3, 9] = 50.0           # exact tie -> lowest index (topk semantics)
    logits[:, ::3, 0] += 6.0                             # plenty of blanks
    logits[:, 1::3, :] = logits[:, 0::3, :][:, :20].clone()   # repeats
    hl = torch.tensor([61, 40, 61, 7, 1], dtype=torch.int32)
    eos = V - 1
    best = F.log_softmax(logits, -1).topk(1, dim=2)[1].view(B, T)
    best = best.masked_fill(O.pad_mask(hl, T), eos)
    want = [O.collapse_ctc_path(r.tolist()) for r in best]
    fb = torch.empty(B, T, dtype=torch.int32, device=DEV)
    ot = torch.empty(B, T, dtype=torch.int32, device=DEV)
    ol = torch.empty(B, dtype=torch.int32, device=DEV)
    lg_d, hl_d = cu(logits), cu(hl)     # keep the device copies alive until the launch is enqueued
    hip.check(hip.lib().oe_ctc_greedy(hip.ptr(lg_d), V, B, T, V, hip.ptr(hl_d), eos, hip.ptr(fb), hip.ptr(ot),
                                      hip.ptr(ol), hip.stream()), "greedy")
    sync()
    got = [ot[b, : int(ol[b])].tolist() for b in range(B)]
    assert got == want
    assert int(fb[0, 3]) == 7


# ------------------------------------------------------ bf16-MFMA GEMM modes -----
def _gemm_layout_errors(prec):
    """Errors against float64, relative to sqrt(K) * |a| * |b| (the natural scale of the sum), of every GEMM form the step
    launches: x W^T + b, dy W, dy^T x with split-K atomics, big tiles with ragged edges, the conv2 implicit GEMMs."""
    torch.manual_seed(20)
    out = {}
    M, N, K = 700, 300, 520
    x, w, dy = torch.randn(M, K), torch.randn(N, K), torch.randn(M, N)
    b = torch.randn(N)
    xd, wd, dyd, bd = cu(x), cu(w), cu(dy), cu(b)

    def err(got, ref, k):
        return float((got.cpu().double() - ref).abs().max()) / math.sqrt(k)

    y = torch.empty(M, N, device=DEV)
    hip.gemm(xd, wd, y, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, precision=prec)
    dx = torch.empty(M, K, device=DEV)
    hip.gemm(dyd, wd, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kmajor=True, precision=prec)
    dw = torch.zeros(N, K, device=DEV)
    hip.gemm(dyd, xd, dw, N, K, M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=3, atomic_out=True, precision=prec)
    sync()
    out["nt"] = err(y, x.double() @ w.double().T + b.double(), K)
    out["nn"] = err(dx, dy.double() @ w.double(), N)
    out["tn"] = err(dw, dy.double().T @ x.double(), M)
    # big-tile path + ragged edges
    M2, N2, K2 = 1100, 1030, 200
    a2, b2 = torch.randn(M2, K2), torch.randn(N2, K2)
    a2d, b2d = cu(a2), cu(b2)
    c2 = torch.empty(M2, N2, device=DEV)
    hip.gemm(a2d, b2d, c2, M2, N2, K2, lda=K2, ldb=K2, ldc=N2, precision=prec)
    dw2 = torch.zeros(N2, K2, device=DEV)
    g2 = torch.randn(M2, N2)
    g2d = cu(g2)
    hip.gemm(g2d, a2d, dw2, N2, K2, M2, lda=N2, ldb=K2, ldc=K2, a_kmajor=True, b_kmajor=True, split_k=1, atomic_out=True, precision=prec)
    sync()
    out["nt big"] = err(c2, a2.double() @ b2.double().T, K2)
    out["tn big"] = err(dw2, g2.double().T @ a2.double(), M2)
    # conv2 implicit GEMM (gather A forward, gather B wgrad)
    B_, T1, F1, Cc = 2, 21, 11, 32
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    xc = torch.randn(B_, Cc, T1, F1)
    wc = torch.randn(Cc, Cc, 3, 3) * 0.1
    ref = F.conv2d(xc.double(), wc.double(), None, stride=2)
    x_nhwc = cu(xc.permute(0, 2, 3, 1))
    w_g = cu(wc.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc))
    Mc = B_ * T2 * F2
    o = torch.empty(Mc, Cc, device=DEV)
    conv = (T1, F1, T2, F2, Cc)
    hip.gemm(x_nhwc, w_g, o, Mc, Cc, 9 * Cc, lda=0, ldb=9 * Cc, ldc=Cc, conv=conv, conv_gather=hip.GATHER_A, precision=prec)
    dyc = torch.randn(Mc, Cc)
    dycd = cu(dyc)
    dwg = torch.zeros(Cc, 9 * Cc, device=DEV)
    hip.gemm(dycd, x_nhwc, dwg, Cc, 9 * Cc, Mc, lda=Cc, ldb=0, ldc=9 * Cc, a_kmajor=True, b_kmajor=True, split_k=2,
             atomic_out=True, conv=conv, conv_gather=hip.GATHER_B, precision=prec)
    sync()
    got = o.cpu().view(B_, T2, F2, Cc).permute(0, 3, 1, 2)
    out["conv fwd"] = float((got.double() - ref).abs().max()) / math.sqrt(9 * Cc) / 0.2
    col = F.unfold(xc.double(), 3, stride=2).transpose(1, 2).reshape(Mc, Cc, 9)      # (m, ci, kh*3+kw)
    ref_dw = torch.einsum("mo,mck->okc", dyc.double(), col).reshape(Cc, 9 * Cc)
    out["conv wgrad"] = float((dwg.cpu().double() - ref_dw).abs().max()) / math.sqrt(Mc)
    return out


@pytest.mark.parametrize("prec,rel", [(3, 2e-5), (1, 6e-3)])
def test_gemm_bf16_precisions_all_layouts(prec, rel):
    """precision 3 (hi*hi + hi*lo + lo*hi) carries ~2^-17 per product; precision 1 is plain bf16 products."""
    for k, e in _gemm_layout_errors(prec).items():
        assert e < rel * 3, (k, e)


def test_gemm_bf16x6_all_layouts_at_the_fp32_kernels_error():
    """precision 6 (three exact pieces, six products) in every layout and kernel (ring, register-staged, planes, implicit
    conv gathers) against precision 0 (gemm_f32_kernel: an fp32 fma chain on v_mfma_f32_32x32x2_f32) on the same problems:
    its error against float64 is not above 1.5 x the fp32 kernel's (measured: equal to within a few per cent - both are
    one fp32 rounding per accumulation) and sits ~20 x below precision 3's."""
    e0, e6 = _gemm_layout_errors(0), _gemm_layout_errors(6)
    print("fp32  :", {k: f"{v:.2e}" for k, v in e0.items()})
    print("bf16x6:", {k: f"{v:.2e}" for k, v in e6.items()})
    for k in e0:
        assert e6[k] <= 1.5 * e0[k] + 2e-7, (k, e0[k], e6[k])
        assert e6[k] < 1.5e-5, (k, e6[k])


def test_gemm_bf16x6_not_above_fp32():
    """The six-term bf16 product against the exact-fp32 MFMA kernel on the same operands, all three layouts, errors against
    float64: the claim "within one fp32 rounding of the fp32 product" means the error of precision 6 is not above that of
    precision 0 (an fp32 fma chain, one rounding per product) - allowed 1.5x, measured below 1x (a bf16 MFMA adds 16
    exact products before it rounds into the accumulator).  Data with a wide dynamic range (exponents spread over 2^+-20)
    so that a fixed-point trick would not pass."""
    torch.manual_seed(21)
    M, N, K = 512, 384, 1024
    spread = lambda *s: torch.randn(*s) * torch.exp2(torch.randint(-20, 21, s).float())
    x, w, dy = spread(M, K), spread(N, K), spread(M, N)
    xd, wd, dyd = cu(x), cu(w), cu(dy)
    refs = (x.double() @ w.double().T, dy.double() @ w.double(), dy.double().T @ x.double())
    mags = (x.abs().double() @ w.abs().double().T, dy.abs().double() @ w.abs().double(), dy.abs().double().T @ x.abs().double())
    errs = {}
    for prec in (0, 6):
        y = torch.empty(M, N, device=DEV)
        hip.gemm(xd, wd, y, M, N, K, lda=K, ldb=K, ldc=N, precision=prec)
        dx = torch.empty(M, K, device=DEV)
        hip.gemm(dyd, wd, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kmajor=True, precision=prec)
        dw = torch.zeros(N, K, device=DEV)
        hip.gemm(dyd, xd, dw, N, K, M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=1, atomic_out=True, precision=prec)
        sync()
        # error relative to sum |a||b| of each output element: the quantity an fp32 chain is bounded in
        errs[prec] = [float(((g.cpu().double() - r).abs() / m.clamp_min(1e-300)).max()) for g, r, m in zip((y, dx, dw), refs, mags)]
    print("max |err| / sum|a||b|  fp32:", errs[0], " bf16x6:", errs[6])
    for e0, e6 in zip(errs[0], errs[6]):
        assert e6 <= max(1.5 * e0, 2.0 ** -23), (errs[0], errs[6])
