"""GPU: the LDS-DMA ring GEMM (gemm_dma.hip) against float64 on shapes that qualify for it
(tiles interior, 16-byte aligned, K % 32 == 0).

The library picks the ring only where it measured faster (weight gradients, x @ W with K >= 512); to cover
every layout / tile / ring-depth case the same checks also run in a child process with OE_GEMM_DMA=2
(ring wherever the problem qualifies) - the switch is read once per process.
"""
import math
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
CASES = [(128, 128, 32), (256, 128, 96), (512, 384, 256), (1024, 1024, 160), (192, 64, 64), (512, 256, 1024),
         # ragged M / N edges (clamped source rows, bounds-checked epilogue on the edge blocks): the decoder's 992 rows, ...
         (992, 256, 256), (250, 100, 64), (76, 36, 96), (992, 1024, 256), (248, 256, 256)]
PRECS = [(3, 2e-5), (1, 6e-3)]


def check_case(prec, rel, M, N, K):
    """One to 32 K-tiles (shorter than, equal to and longer than the 4-deep ring), the three operand layouts,
    bias / residual / activation epilogues, split-K atomics and the fused bias gradient."""
    from openeat_amd import hip

    torch.manual_seed(21)
    x, w, dy, b = torch.randn(M, K), torch.randn(N, K), torch.randn(M, N), torch.randn(N)
    res = torch.randn(M, N)
    xd, wd, dyd, bd, resd = (t.to(DEV) for t in (x, w, dy, b, res))

    def err(got, ref, k):
        return float((got.cpu().double() - ref).abs().max()) / math.sqrt(k)

    y = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(xd, wd, y, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, residual=resd, ldr=N, beta=0.5, precision=prec)
    pre = torch.full((M, N), float("nan"), device=DEV)
    ya = torch.full((M, N), float("nan"), device=DEV)
    hip.gemm(xd, wd, ya, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, act=2, preact_out=pre, ld_aux=N, precision=prec)
    # dgrad: dx[M,K'] = dy[M,N] @ W[N,K'] needs K' % 64 == 0 to qualify as an output width
    Kp = 64 * max(1, K // 64)
    w2 = torch.randn(N, Kp)
    w2d = w2.to(DEV)
    dx = torch.full((M, Kp), float("nan"), device=DEV)
    hip.gemm(dyd, w2d, dx, M, Kp, N, lda=N, ldb=Kp, ldc=Kp, b_kmajor=True, precision=prec)
    # same with the activation derivative fused (relu'(aux)) and a row mask
    auxd = torch.randn(M, Kp, device=DEV)
    mask = (torch.rand(M) > 0.2).to(torch.uint8).to(DEV)
    dxa = torch.full((M, Kp), float("nan"), device=DEV)
    hip.gemm(dyd, w2d, dxa, M, Kp, N, lda=N, ldb=Kp, ldc=Kp, b_kmajor=True, act=1, actgrad_in=auxd, ld_aux=Kp,
             rowmask=mask, precision=prec)
    # wgrad dW[N,Kp] = dy^T x2 with the reduction over M split three ways + bias gradient
    x2 = torch.randn(M, Kp)
    x2d = x2.to(DEV)
    dw = torch.zeros(N, Kp, device=DEV)
    db = torch.zeros(N, device=DEV)
    sk = 3 if M >= 96 * 3 else 1
    hip.gemm(dyd, x2d, dw, N, Kp, M, lda=N, ldb=Kp, ldc=Kp, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True,
             precision=prec, a_colsum=db)
    torch.cuda.synchronize()
    lin = x.double() @ w.double().T + b.double()
    assert err(y, res.double() + 0.5 * lin, K) < rel * 3
    assert err(pre, lin, K) < rel * 3
    assert err(ya, lin * torch.sigmoid(lin), K) < rel * 3 * 1.2
    ref_dx = dy.double() @ w2.double()
    assert err(dx, ref_dx, N) < rel * 3
    ref_dxa = ref_dx * (auxd.cpu().double() > 0) * mask.cpu().double()[:, None]
    assert err(dxa, ref_dxa, N) < rel * 3
    assert err(dw, dy.double().T @ x2.double(), M) < rel * 3
    torch.testing.assert_close(db.cpu().double(), dy.double().sum(0), rtol=1e-4, atol=1e-4 * math.sqrt(M))


@pytest.mark.parametrize("prec,rel", PRECS)
@pytest.mark.parametrize("M,N,K", CASES)
def test_gemm_aligned_shapes_default_dispatch(prec, rel, M, N, K):
    check_case(prec, rel, M, N, K)


@pytest.mark.parametrize("prec,rel", PRECS)
@pytest.mark.parametrize("Cc,B_,T1,F1,sk", [(64, 4, 33, 17, 1), (64, 4, 33, 17, 3), (128, 2, 65, 17, 2)])
def test_conv2_weight_gradient_gather_ring(prec, rel, Cc, B_, T1, F1, sk):
    """conv2 weight gradient with the im2col gather done by the DMA pieces' source addresses: channel counts whose
    kernel rows (3C floats) are whole column tiles and B*T2*F2 a multiple of the K-tile, against unfold + einsum."""
    import torch.nn.functional as F
    from openeat_amd import hip

    torch.manual_seed(5)
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    Mc = B_ * T2 * F2
    assert Mc % 32 == 0
    xc = torch.randn(B_, Cc, T1, F1)
    x_nhwc = xc.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyc = torch.randn(Mc, Cc)
    dycd = dyc.to(DEV)
    dwg = torch.zeros(Cc, 9 * Cc, device=DEV)
    db = torch.zeros(Cc, device=DEV)
    hip.gemm(dycd, x_nhwc, dwg, Cc, 9 * Cc, Mc, lda=Cc, ldb=0, ldc=9 * Cc, a_kmajor=True, b_kmajor=True, split_k=sk,
             atomic_out=True, conv=(T1, F1, T2, F2, Cc), conv_gather=hip.GATHER_B, precision=prec, a_colsum=db)
    torch.cuda.synchronize()
    col = F.unfold(xc.double(), 3, stride=2).transpose(1, 2).reshape(Mc, Cc, 9)      # (m, ci, kh*3+kw)
    ref_dw = torch.einsum("mo,mck->okc", dyc.double(), col).reshape(Cc, 9 * Cc)
    assert float((dwg.cpu().double() - ref_dw).abs().max()) / math.sqrt(Mc) < rel * 3
    torch.testing.assert_close(db.cpu().double(), dyc.double().sum(0), rtol=1e-4, atol=1e-4 * math.sqrt(Mc))


def test_gemm_dma_ring_forced_everywhere():
    env = dict(os.environ, OE_GEMM_DMA="2", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for tile in ("0", "22", "42"):          # 0 = the library's own tile choice
        env["OE_GEMM_TILE"] = tile
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"tile {tile}:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
        assert "dma-ring cases ok" in r.stdout


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    assert os.environ.get("OE_GEMM_DMA") == "2"
    for prec, rel in PRECS:
        for M, N, K in CASES:
            check_case(prec, rel, M, N, K)
        test_conv2_weight_gradient_gather_ring(prec, rel, 128, 2, 65, 17, 2)      # 128x128 tiles when OE_GEMM_TILE=22
    print("dma-ring cases ok")


TN_CASES = [  # (rows of dW, cols of dW, reduction length, split asked by the caller, atomic, with bias gradient)
    (1024, 256, 7936, 16, True, True), (256, 1024, 7936, 16, True, False), (256, 256, 7936, 24, True, True),
    (768, 256, 1984, 21, True, True), (200, 132, 1100, 3, True, True), (128, 128, 1024, 1, False, False),
    (260, 388, 1500, 1, False, True), (128, 4864, 1024, 1, True, False), (3246, 256, 2100, 4, True, True), (254, 130, 2050, 2, True, True)]


@pytest.mark.parametrize("prec,rel", PRECS)
@pytest.mark.parametrize("m,n,k,sk,atomic,bias", TN_CASES)
def test_weight_gradient_planes_kernel(prec, rel, m, n, k, sk, atomic, bias):
    """gemm_tn.hip (both operands k-major, bf16 planes in LDS, two K-groups per block): dW = alpha dY^T X against float64,
    accumulating on top of what the destination holds (atomic split-K) or overwriting / accumulating in place (one split),
    the bias gradient fused, ragged output edges (multiples of 4), a ragged last K-chunk, a device-side alpha."""
    from openeat_amd import hip
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    dy, x = torch.randn(k, m, generator=g), torch.randn(k, n, generator=g)
    start = torch.randn(m, n, generator=g)
    # rows padded to whole float4s (filled with NaN: nothing of the padding may reach the result), as the CTC head's
    # logits-gradient buffer is (3246 columns in rows of 3248)
    mp, npad = (m + 3) // 4 * 4, (n + 3) // 4 * 4
    dyd, xd = torch.full((k, mp), float("nan"), device=DEV), torch.full((k, npad), float("nan"), device=DEV)
    dyd[:, :m], xd[:, :n] = dy.to(DEV), x.to(DEV)
    alpha_dev = torch.tensor([0.5], device=DEV)
    for accumulate in ((True,) if atomic else (False, True)):
        dw = start.to(DEV).clone()
        db = torch.full((m,), 2.0, device=DEV)
        hip.gemm(dyd, xd, dw, m, n, k, lda=mp, ldb=npad, ldc=n, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=atomic,
                 accumulate=accumulate and not atomic, alpha=3.0, alpha_dev=alpha_dev, a_colsum=db if bias else None, precision=prec)
        torch.cuda.synchronize()
        ref = 1.5 * (dy.double().t() @ x.double()) + (start.double() if (atomic or accumulate) else 0.0)
        err = float((dw.cpu().double() - ref).abs().max()) / math.sqrt(k)
        assert err < 1.5 * rel * 2.5, (accumulate, err)        # alpha = 1.5; the maximum over up to 8e5 outputs of a long sum
        if bias:
            refb = 2.0 + 1.5 * dy.double().sum(0)
            assert float((db.cpu().double() - refb).abs().max()) < 2e-3 * math.sqrt(k)


@pytest.mark.parametrize("prec,rel", PRECS)
@pytest.mark.parametrize("target_blocks", [384, 40, 4000])
def test_grouped_weight_gradients(prec, rel, target_blocks):
    """oe_gemm_tn_grouped: several dW_i (+)= alpha_i dY_i^T X_i in one launch against float64 - outputs from one tile to
    many, different reduction lengths in one table (K = 248 rows of linear_pos beside K = 7936), a ragged row length inside a
    padded leading dimension, fused bias gradients, a device-side alpha; at splits from none to many."""
    from openeat_amd import hip
    g = torch.Generator().manual_seed(77)
    shapes = [(768, 256, 1984, True), (256, 256, 1984, True), (512, 256, 1984, False), (256, 256, 248, False), (1024, 256, 1984, True),
              (40, 32, 300, True), (326, 64, 1100, True)]
    alpha_dev = torch.tensor([0.25], device=DEV)
    problems, refs = [], []
    for i, (m, n, k, bias) in enumerate(shapes):
        mp = (m + 3) // 4 * 4 + (4 if i == 6 else 0)
        dy = torch.full((k, mp), float("nan"))
        dy[:, :m] = torch.randn(k, m, generator=g)
        x = torch.randn(k, n, generator=g)
        start = torch.randn(m, n, generator=g)
        alpha = 1.0 + 0.5 * i
        out = start.to(DEV).clone()
        db = torch.full((m,), 3.0, device=DEV) if bias else None
        dyd = dy.to(DEV)
        problems.append(dict(dy=dyd[:, :m], x=x.to(DEV), out=out, alpha=alpha, alpha_dev=alpha_dev if i % 2 else None, bias_out=db))
        a = alpha * (0.25 if i % 2 else 1.0)
        refs.append((start.double() + a * dy[:, :m].double().t() @ x.double(), None if not bias else 3.0 + a * dy[:, :m].double().sum(0), k, a))
    plan = hip.tn_grouped_plan(problems, target_blocks)
    assert plan is not None
    host, total = plan
    table = torch.empty(len(host), dtype=torch.uint8, device=DEV)
    table.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
    hip.tn_grouped_launch(table, len(problems), total, precision=prec)
    torch.cuda.synchronize()
    for q, (ref, refb, k, a) in zip(problems, refs):
        err = float((q["out"].cpu().double() - ref).abs().max()) / math.sqrt(k)
        assert err < max(a, 1.0) * rel * 2.5, (tuple(ref.shape), k, err)
        if refb is not None:
            assert float((q["bias_out"].cpu().double() - refb).abs().max()) < 2e-3 * math.sqrt(k) * max(a, 1.0)
    # a problem that cannot take the kernel is refused by the plan (leading dimension not a whole float4)
    bad = dict(problems[0], dy=torch.randn(1984, 770, device=DEV)[:, :768])
    assert hip.tn_grouped_plan([bad, problems[1]], 384) is None
