"""GPU: pre-split GEMM operands (precision 6): oe_split_planes, the planes outputs of LayerNorm and of the GEMM epilogue,
and csrc/gemm_pl.hip (tiles by LDS-DMA, six MFMA products per fragment pair, no conversion in the loop) against float64 and
against the exact-fp32 kernel - every layout, ragged edges, split-K with the fused bias gradient, the conv gathers."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openeat_amd import hip, planes  # noqa: E402

DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def sync():
    torch.cuda.synchronize()


@pytest.fixture(autouse=True)
def mode6():
    old, old_min, old_pol = hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY
    hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY = 6, 0, "all"
    hip.lib().oe_gemm_pl_config(0, -1, -1, -1)          # accept any grid: the small shapes here are meant for gemm_pl.hip too
    planes.clear()
    yield
    hip.GEMM_PRECISION, planes.MIN_SPLIT_ELEMS, planes.POLICY = old, old_min, old_pol
    hip.lib().oe_gemm_pl_config(96, 0, 0, 8)
    planes.clear()


def split(x):
    pl = planes.of(x, force=True)
    assert pl is not None
    return pl


def test_split_planes_is_exact_and_pieces_are_ordered():
    torch.manual_seed(1)
    x = cu(torch.randn(300, 264) * torch.exp2(torch.randint(-30, 31, (300, 264)).float()))
    # (values within 2^-8 of FLT_MAX round up to a bf16 infinity and values below 2^-110 have subnormal pieces: neither
    # occurs in activations, weights or gradients; the range exercised here is 2^-90 .. 2^+100)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 2.0 ** -90, 2.0 ** 100 * 1.2345678, 1 + 2 ** -23, 1 - 2 ** -24], device=DEV)
    pl = split(x)
    sync()
    assert pl.t.shape[1] == 304 and pl.kpad and float(pl.t[:, 300:].float().abs().max()) == 0.0   # zero pad rows to a multiple of 16
    p = pl.t[:, :pl.rows].double().cpu()
    back = (p[0] + p[1] + p[2]).float()                                          # float64 sum of three bf16 values: exact
    bad = (back != x.cpu()).nonzero()
    assert bad.numel() == 0, (bad[:5], x.cpu()[tuple(bad[0])] if bad.numel() else None, p[:, bad[0][0], bad[0][1]] if bad.numel() else None)
    big = x.abs().cpu().double().clamp_min(1e-300)
    assert float((p[1].abs() / big).max()) <= 2.0 ** -8 and float((p[2].abs() / big).max()) <= 2.0 ** -16


def _err(got, ref, k):
    return float((got.cpu().double() - ref).abs().max()) / math.sqrt(k)


@pytest.mark.parametrize("waves,tile", [(8, 0), (4, 0), (8, 11), (8, 24), (8, 44)])
@pytest.mark.parametrize("M,N,K", [(7936, 1024, 256), (640, 384, 1024), (704, 304, 512), (700, 304, 512), (130, 136, 48), (64, 8, 16)])
def test_planes_gemm_all_layouts_vs_float64_and_fp32_kernel(M, N, K, waves, tile):
    """x W^T (+ bias, planes output), dy W, dy^T x (split-K atomics + fused column sums) on pre-split operands: error against
    float64 not above 1.5 x the exact-fp32 kernel's on the same problem; the planes output equals a split of the output."""
    torch.manual_seed(2)
    x, w, dy, b = torch.randn(M, K), torch.randn(N, K), torch.randn(M, N), torch.randn(N)
    xd, wd, dyd, bd = cu(x), cu(w), cu(dy), cu(b)
    ref_y = x.double() @ w.double().T + b.double()
    ref_dx = dy.double() @ w.double()
    ref_dw = dy.double().T @ x.double()
    res = {}
    hip.lib().oe_gemm_pl_config(-1, tile, -1, waves)    # block shape: 8 waves (4 x 2), 4 waves (2 x 2), 64 x 64 tiles for x W^T
    n0 = hip.lib().oe_gemm_pl_launches()
    for prec, use_pl in ((0, False), (6, True)):
        xp, wp, dyp = (split(xd), split(wd), split(dyd)) if use_pl else (None, None, None)
        y = torch.empty(M, N, device=DEV)
        yp = planes.alloc(M, N, DEV) if use_pl else None
        hip.gemm(xd, wd, y, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, precision=prec, a_planes=xp, b_planes=wp, c_planes=yp)
        dx = torch.empty(M, K, device=DEV)
        hip.gemm(dyd, wd, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kmajor=True, precision=prec, a_planes=dyp, b_planes=wp)
        dw = torch.zeros(N, K, device=DEV)
        db = torch.zeros(N, device=DEV)
        sk = 3 if M >= 512 else 1
        hip.gemm(dyd, xd, dw, N, K, M, lda=N, ldb=K, ldc=K, a_kmajor=True, b_kmajor=True, split_k=sk, atomic_out=True, precision=prec,
                 a_planes=dyp, b_planes=xp, a_colsum=db if use_pl else None)
        sync()
        res[prec] = (_err(y, ref_y, K), _err(dx, ref_dx, N), _err(dw, ref_dw, M))
        if use_pl:
            p = yp.t[:, :yp.rows].double().cpu()
            assert torch.equal((p[0] + p[1] + p[2]).float(), y.cpu())
            assert float((db.cpu().double() - dy.double().sum(0)).abs().max()) <= 1e-4 * math.sqrt(M)
    # x W^T always qualifies when K is whole K-tiles (the other two depend on N, M and the forced tile): the numbers
    # above must come from gemm_pl.hip, not from a silent fallback
    ran = hip.lib().oe_gemm_pl_launches() - n0
    if K % 32 == 0:                       # (64 x 64 and 128 x 256 tiles take K-tiles of 32)
        assert ran >= 1, "a problem meant for gemm_pl.hip took another kernel"
    if K % 16 == 0 and N % 16 == 0 and M % 16 == 0 and tile == 0:
        assert ran == 3
    print(f"M={M} N={N} K={K}: fp32 {res[0]}  planes {res[6]}  ({ran} of 3 on gemm_pl.hip)")
    for e0, e6 in zip(res[0], res[6]):
        assert e6 <= 1.5 * e0 + 2e-7, (res[0], res[6])


def test_planes_gemm_epilogues_match_the_plain_kernels():
    """Every epilogue feature through gemm_pl.hip (activation + pre-activation copy + dropout; act-grad; residual + beta + row
    mask) equals the same call without planes (same dropout bits) to fp32 rounding."""
    torch.manual_seed(3)
    M, N, K = 1024, 512, 256
    x, w, b, res = cu(torch.randn(M, K)), cu(torch.randn(N, K) * 0.1), cu(torch.randn(N)), cu(torch.randn(M, N))
    rowmask = cu((torch.rand(M) > 0.2).to(torch.uint8))
    outs = {}
    n0 = hip.lib().oe_gemm_pl_launches()
    for use_pl in (False, True):
        xp, wp = (split(x), split(w)) if use_pl else (None, None)
        pre = torch.empty(M, N, device=DEV)
        a = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, a, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2, preact_out=pre, ld_aux=N, drop_p=0.1, seed=77, a_planes=xp, b_planes=wp)
        y = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, residual=res, ldr=N, beta=0.5, rowmask=rowmask, drop_p=0.1, seed=78,
                 a_planes=xp, b_planes=wp)
        g = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, g, M, N, K, lda=K, ldb=K, ldc=N, act=2, actgrad_in=pre, ld_aux=N, a_planes=xp, b_planes=wp)
        sync()
        outs[use_pl] = (pre.cpu(), a.cpu(), y.cpu(), g.cpu())
    assert hip.lib().oe_gemm_pl_launches() - n0 == 3
    for t0, t1 in zip(outs[False], outs[True]):
        torch.testing.assert_close(t1, t0, rtol=1e-5, atol=1e-5)
        assert torch.equal(t0 == 0, t1 == 0)                     # the same elements dropped / masked


def test_layernorm_planes_outputs():
    torch.manual_seed(4)
    rows, d = 777, 256
    x, gamma, beta = cu(torch.randn(rows, d)), cu(torch.rand(d) + 0.5), cu(torch.randn(d))
    y, stats = torch.empty_like(x), torch.empty(rows, 2, device=DEV)
    pl = planes.alloc(rows, d, DEV)
    hip.call("oe_layernorm_fwd_pl", x, gamma, beta, 1e-5, rows, d, None, 2, y, stats, pl.t, pl.stride)
    y0, st0 = torch.empty_like(x), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_fwd", x, gamma, beta, 1e-5, rows, d, None, 2, y0, st0)
    sync()
    assert torch.equal(y, y0)
    p = pl.t[:, :pl.rows].double().cpu()
    assert torch.equal((p[0] + p[1] + p[2]).float(), y.cpu())
    dy = cu(torch.randn(rows, d))
    ws = torch.empty(hip.lib().oe_layernorm_bwd_workspace_floats(rows, d), device=DEV)
    dx, g = torch.empty_like(x), torch.empty_like(x)
    hip.call("oe_layernorm_bwd_dx_drop_pl", dy, x, gamma, beta, 2, stats, rows, d, None, None, dx, g, 0.5, 0.1, 99, None, None, ws, pl.t, pl.stride)
    dx0, g0 = torch.empty_like(x), torch.empty_like(x)
    hip.call("oe_layernorm_bwd_dx_drop", dy, x, gamma, beta, 2, stats, rows, d, None, None, dx0, g0, 0.5, 0.1, 99, None, None, ws)
    sync()
    assert torch.equal(dx, dx0) and torch.equal(g, g0)
    p = pl.t[:, :pl.rows].double().cpu()
    assert torch.equal((p[0] + p[1] + p[2]).float(), g.cpu())                    # planes follow the dropped copy when there is one


@pytest.mark.parametrize("tile", [0, 24, 44])
@pytest.mark.parametrize("B_,T1,F1,Cc", [(2, 21, 11, 32), (3, 40, 39, 64), (4, 30, 21, 128), (4, 37, 33, 256),
                                         # ragged batches: B T2 F2 positions that are NOT a multiple of the K-tile (495 = 30 * 16 + 15,
                                         # 1530 = 95 * 16 + 10): the weight gradient runs the padded length on zero pad rows of dY's
                                         # planes and clamped gather positions (oe_gemm_args.planes_k_padded)
                                         (3, 31, 23, 128), (5, 37, 35, 256)])
def test_planes_conv2_gathers(B_, T1, F1, Cc, tile):
    """conv2 forward (im2col gather on a row-major A) and weight gradient (gather on a k-major B) on pre-split operands."""
    torch.manual_seed(5)
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    xc = torch.randn(B_, Cc, T1, F1)
    wc = torch.randn(Cc, Cc, 3, 3) * 0.1
    ref = F.conv2d(xc.double(), wc.double(), None, stride=2)
    x_nhwc = cu(xc.permute(0, 2, 3, 1))
    w_g = cu(wc.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc))
    Mc = B_ * T2 * F2
    conv = (T1, F1, T2, F2, Cc)
    xp, wp = split(x_nhwc.view(-1, Cc)), split(w_g)
    hip.lib().oe_gemm_pl_config(-1, tile, -1, -1)            # 24: 128 x 256 tiles (the conv2 forward of the real model, C = 256)
    n0 = hip.lib().oe_gemm_pl_launches()
    out = torch.empty(Mc, Cc, device=DEV)
    hip.gemm(x_nhwc, w_g, out, Mc, Cc, 9 * Cc, lda=0, ldb=9 * Cc, ldc=Cc, conv=conv, conv_gather=hip.GATHER_A, a_planes=xp, b_planes=wp)
    sync()
    assert hip.lib().oe_gemm_pl_launches() - n0 == 1
    got = out.cpu().view(B_, T2, F2, Cc).permute(0, 3, 1, 2)
    assert float((got.double() - ref).abs().max()) / math.sqrt(9 * Cc) < 1.5e-6
    if Cc % 32 == 0:
        # the same product with the reduction walked channel-chunk major (oe_gemm_args.conv_korder = 1; B's columns laid out to
        # match): the taps of 32 channels back to back - and a non-square window (2 x 3: the input gradient's parity classes)
        from openeat_amd.ops import _korder_cols
        out2 = torch.empty(Mc, Cc, device=DEV)
        hip.gemm(x_nhwc, w_g, out2, Mc, Cc, 9 * Cc, lda=0, ldb=9 * Cc, ldc=Cc, conv=conv, conv_gather=hip.GATHER_A, a_planes=xp,
                 b_planes=split(_korder_cols(w_g, 9, Cc)), conv_korder=1)
        sync()
        got2 = out2.cpu().view(B_, T2, F2, Cc).permute(0, 3, 1, 2)
        assert float((got2.double() - ref).abs().max()) / math.sqrt(9 * Cc) < 1.5e-6
        w23 = torch.randn(Cc, Cc, 2, 3) * 0.1
        T3, F3 = (T1 - 2) // 2 + 1, (F1 - 3) // 2 + 1
        ref23 = F.conv2d(xc.double(), w23.double(), None, stride=2)
        w23g = cu(w23.permute(0, 2, 3, 1).reshape(Cc, 6 * Cc))
        out3 = torch.empty(B_ * T3 * F3, Cc, device=DEV)
        hip.gemm(x_nhwc, w23g, out3, B_ * T3 * F3, Cc, 6 * Cc, lda=0, ldb=6 * Cc, ldc=Cc, conv=(T1, F1, T3, F3, Cc), conv_gather=hip.GATHER_A,
                 conv_kh=2, a_planes=xp, b_planes=split(_korder_cols(w23g, 6, Cc)), conv_korder=1)
        sync()
        got3 = out3.cpu().view(B_, T3, F3, Cc).permute(0, 3, 1, 2)
        assert float((got3.double() - ref23).abs().max()) / math.sqrt(6 * Cc) < 1.5e-6
    if Cc % 128 == 0 or (9 * Cc) % 128 == 0:
        dyc = torch.randn(Mc, Cc)
        dycd = cu(dyc)
        dwg = torch.zeros(Cc, 9 * Cc, device=DEV)
        hip.lib().oe_gemm_pl_config(-1, 44 if (tile == 44 and Cc % 256 == 0) else 0, -1, -1)    # k-major operands: 128 x 128 or 256 x 256 tiles
        n0 = hip.lib().oe_gemm_pl_launches() - 1
        hip.gemm(dycd, x_nhwc, dwg, Cc, 9 * Cc, Mc, lda=Cc, ldb=0, ldc=9 * Cc, a_kmajor=True, b_kmajor=True, split_k=2, atomic_out=True,
                 conv=conv, conv_gather=hip.GATHER_B, a_planes=split(dycd), b_planes=xp)
        sync()
        assert hip.lib().oe_gemm_pl_launches() - n0 == 2                     # the pre-split kernel took it, whatever Mc % 16 is
        col = F.unfold(xc.double(), 3, stride=2).transpose(1, 2).reshape(Mc, Cc, 9)
        ref_dw = torch.einsum("mo,mck->okc", dyc.double(), col).reshape(Cc, 9 * Cc)
        assert float((dwg.cpu().double() - ref_dw).abs().max()) / math.sqrt(Mc) < 5e-6


@pytest.mark.parametrize("M,N,K", [(7936, 1024, 256), (7936, 256, 1024), (7930, 768, 256), (3072, 3246, 256), (25472, 512, 512), (25472, 256, 64),
                                   (12800, 1024, 64)])
def test_weight_planes_gemm_vs_float64_and_fp32_kernel(M, N, K):
    """csrc/gemm_hyb.hip: x W^T and dy W with ONLY the weight operand pre-split (the activation is split on the fragment): error
    against float64 not above 1.5 x the exact-fp32 kernel's on the same problem, ragged M / N edges included; the launches counted."""
    lib = hip.lib()
    torch.manual_seed(5)
    x, w, dy, b = torch.randn(M, K), torch.randn(N, K), torch.randn(M, N), torch.randn(N)
    xd, wd, dyd, bd = cu(x), cu(w), cu(dy), cu(b)
    ref_y = x.double() @ w.double().T + b.double()
    ref_dx = dy.double() @ w.double()
    res = {}
    n0 = lib.oe_gemm_hyb_launches()
    for prec, use_pl in ((0, False), (6, True)):
        wp = split(wd) if use_pl else None
        y = torch.empty(M, N, device=DEV)
        hip.gemm(xd, wd, y, M, N, K, lda=K, ldb=K, ldc=N, bias=bd, precision=prec, b_planes=wp)
        dx = torch.empty(M, K, device=DEV)
        hip.gemm(dyd, wd, dx, M, K, N, lda=N, ldb=K, ldc=K, b_kmajor=True, precision=prec, b_planes=wp)
        sync()
        res[prec] = (_err(y, ref_y, K), _err(dx, ref_dx, N))
    ran = lib.oe_gemm_hyb_launches() - n0
    print(f"M={M} N={N} K={K}: fp32 {res[0]}  weight planes {res[6]}  ({ran} of 2 on gemm_hyb.hip)")
    assert ran >= 1, "x W^T with a pre-split weight at these sizes is meant for gemm_hyb.hip"
    for e0, e6 in zip(res[0], res[6]):
        assert e6 <= 1.5 * e0 + 2e-7, (res[0], res[6])


def test_weight_planes_gemm_epilogues_match_the_plain_kernels():
    """Every epilogue feature through gemm_hyb.hip equals the same call without planes (same dropout bits) to fp32 rounding."""
    lib = hip.lib()
    torch.manual_seed(3)
    M, N, K = 7936, 1024, 256
    x, w, b, res = cu(torch.randn(M, K)), cu(torch.randn(N, K) * 0.1), cu(torch.randn(N)), cu(torch.randn(M, N))
    rowmask = cu((torch.rand(M) > 0.2).to(torch.uint8))
    outs = {}
    n0 = lib.oe_gemm_hyb_launches()
    for use_pl in (False, True):
        wp = split(w) if use_pl else None
        pre = torch.empty(M, N, device=DEV)
        a = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, a, M, N, K, lda=K, ldb=K, ldc=N, bias=b, act=2, preact_out=pre, ld_aux=N, drop_p=0.1, seed=77, b_planes=wp)
        y = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, residual=res, ldr=N, beta=0.5, rowmask=rowmask, drop_p=0.1, seed=78, b_planes=wp)
        g = torch.empty(M, N, device=DEV)
        hip.gemm(x, w, g, M, N, K, lda=K, ldb=K, ldc=N, act=2, actgrad_in=pre, ld_aux=N, b_planes=wp)
        sync()
        outs[use_pl] = (pre.cpu(), a.cpu(), y.cpu(), g.cpu())
    assert lib.oe_gemm_hyb_launches() - n0 == 3
    for t0, t1 in zip(outs[False], outs[True]):
        torch.testing.assert_close(t1, t0, rtol=1e-5, atol=1e-5)
        assert torch.equal(t0 == 0, t1 == 0)


def test_whole_rounds_of_big_tiles_in_front_of_the_small_ones_give_the_same_product():
    """gemm_pl.hip's hybrid dispatch (oe_gemm_pl_hybrid): rows [0, m1) of a 128 x 256-tile problem on 256 x 256 tiles, the rest on
    128 x 256 - two launches over row ranges of the same operands.  A plain x W^T with bias and planes output, and the conv2 forward
    gather in channel-chunk order, at row counts where the split pays (>= one round of big tiles, a ragged last tile): equal to the
    one-launch result element for element (both tiles walk the reduction in the same order), which the tests above hold to float64."""
    from openeat_amd.ops import _korder_cols
    lib = hip.lib()
    torch.manual_seed(9)
    was = lib.oe_gemm_pl_hybrid(-1)
    lib.oe_gemm_pl_config(96, 0, 0, 8)
    try:
        # (a) x W^T + b, 70003 x 256 over K = 512, planes output
        M, N, K = 70003, 256, 512
        x, w, b = cu(torch.randn(M, K)), cu(torch.randn(N, K) * 0.1), cu(torch.randn(N))
        xp, wp = split(x), split(w)
        res = []
        for on in (0, 1):
            lib.oe_gemm_pl_hybrid(on)
            y = torch.full((M, N), float("nan"), device=DEV)
            yp = planes.alloc(M, N, DEV)
            n0 = lib.oe_gemm_pl_launches()
            hip.gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, bias=b, precision=6, a_planes=xp, b_planes=wp, c_planes=yp)
            sync()
            assert lib.oe_gemm_pl_launches() - n0 == 1 + on
            res.append((y.clone(), yp.t[:, :M].clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        rows = torch.randint(0, M, (512,))
        ref = x[rows].double().cpu() @ w.double().cpu().T + b.double().cpu()
        assert float((res[1][0][rows].cpu().double() - ref).abs().max()) / math.sqrt(K) < 1.5e-6
        # (a') two column tiles (N = 512): whole rounds count tiles of both columns, the row split stays a multiple of 256
        M2, N2, K2 = 40003, 512, 512
        x2, w2 = cu(torch.randn(M2, K2)), cu(torch.randn(N2, K2) * 0.1)
        xp2, wp2 = split(x2), split(w2)
        res2 = []
        for on in (0, 1):
            lib.oe_gemm_pl_hybrid(on)
            y2 = torch.full((M2, N2), float("nan"), device=DEV)
            n0 = lib.oe_gemm_pl_launches()
            hip.gemm(x2, w2, y2, M2, N2, K2, lda=K2, ldb=K2, ldc=N2, precision=6, a_planes=xp2, b_planes=wp2)
            sync()
            assert lib.oe_gemm_pl_launches() - n0 == 1 + on
            res2.append(y2)
        assert torch.equal(res2[0], res2[1])
        rows2 = torch.randint(0, M2, (256,))
        ref2 = x2[rows2].double().cpu() @ w2.double().cpu().T
        assert float((res2[1][rows2].cpu().double() - ref2).abs().max()) / math.sqrt(K2) < 1.5e-6
        del x2, w2, xp2, wp2, res2
        # (b) conv2 forward gather: B = 16, (T1, F1) = (161, 115) -> (80, 57): 72960 output positions, C = 256, K = 2304
        B_, T1, F1, Cc = 16, 161, 115, 256
        T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
        Mc = B_ * T2 * F2
        x_nhwc = cu(torch.randn(B_, T1, F1, Cc))
        w_g = cu(torch.randn(Cc, 9 * Cc) * 0.05)
        bias = cu(torch.randn(Cc))
        ap, bp = split(x_nhwc.view(-1, Cc)), split(_korder_cols(w_g, 9, Cc))
        outs = []
        for on in (0, 1):
            lib.oe_gemm_pl_hybrid(on)
            out = torch.full((Mc, Cc), float("nan"), device=DEV)
            n0 = lib.oe_gemm_pl_launches()
            hip.gemm(None, w_g, out, Mc, Cc, 9 * Cc, lda=0, ldb=9 * Cc, ldc=Cc, bias=bias, act=1, conv=(T1, F1, T2, F2, Cc), conv_gather=hip.GATHER_A,
                     a_planes=ap, b_planes=bp, conv_korder=1, precision=6)
            sync()
            assert lib.oe_gemm_pl_launches() - n0 == 1 + on
            outs.append(out)
        assert torch.equal(outs[0], outs[1])
        assert bool(torch.isfinite(outs[1]).all())
    finally:
        lib.oe_gemm_pl_hybrid(was)
        planes.clear()
