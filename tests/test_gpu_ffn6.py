"""GPU: the one-kernel feed forward in the headline arithmetic (precision 6, csrc/ffn6.hip) and its input gradient, through the
C ABI (oe_ffn_pack_weights / oe_ffn_fwd / oe_ffn_pack_weights_bwd / oe_ffn_bwd), against float64 restatements of
positionwise_feed_forward.py:36-43 with the caller's dropout / scaled residual (encoder_layer.py:81-83,104-106).

Tolerance: precision 6 is held to what the exact-fp32-input GEMM path itself shows against float64 on the same problem (the
two-GEMM path at oe_gemm_args.precision = 0), with a factor for the different summation order - not to a widened bf16 figure."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from openeat_amd import hip, ops  # noqa: E402

DEV = "cuda"


def cu(t):
    return t.to(DEV).contiguous()


def sync():
    torch.cuda.synchronize()


def test_precision6_shapes_are_supported():
    L = hip.lib()
    assert L.oe_ffn_supported(256, 1024, 6, 2) and L.oe_ffn_supported(512, 2048, 6, 2) and L.oe_ffn_supported(128, 512, 6, 1)
    assert not L.oe_ffn_supported(384, 1024, 6, 2) and not L.oe_ffn_supported(256, 1000, 6, 2) and not L.oe_ffn_supported(256, 1024, 6, 3)
    assert L.oe_ffn_packed_bytes(256, 1024, 6) == 256 * 1024 * 2 * 3


def _unfused_fp32(x, w1, b1, w2, b2, res, act, beta, p_in, s_in, p_out, s_out, ctr):
    """The same feed forward as two oe_gemm_f32 launches on the fp32-input MFMA (precision 0): the error yardstick."""
    rows, d = x.shape
    ff = w1.shape[0]
    pre = torch.empty(rows, ff, device=DEV)
    a = torch.empty(rows, ff, device=DEV)
    hip.gemm(x, w1, a, rows, ff, d, lda=d, ldb=d, ldc=ff, bias=b1, act=act, preact_out=pre, ld_aux=ff, drop_p=p_in, seed=s_in, seed_dev=ctr,
             precision=0)
    y = torch.empty(rows, d, device=DEV)
    hip.gemm(a, w2, y, rows, d, ff, lda=ff, ldb=ff, ldc=d, bias=b2, drop_p=p_out, seed=s_out, seed_dev=ctr, residual=res, ldr=d, beta=beta,
             precision=0)
    return y, pre, a


@pytest.mark.parametrize("rows,d,ff,act,p_in,p_out,nout", [
    (7936, 256, 1024, 2, 0.1, 0.1, 2),            # the benchmark's encoder feed-forward (124 blocks of 64 rows)
    (1000, 256, 1024, 2, 0.1, 0.1, 1),            # ragged last block (1000 = 15 * 64 + 40), pre-activation only
    (77, 256, 512, 1, 0.0, 0.0, 0),               # relu, nothing saved, a second block of 13 rows
    (333, 128, 512, 1, 0.0, 0.25, 2),             # d = 128 (configs[0] width): one output tile per wave
    (64, 128, 128, 0, 0.0, 0.0, 2),               # one chunk, no activation
    (1500, 512, 2048, 2, 0.1, 0.1, 2),            # configs[4] width: 32-row blocks, four output tiles per wave
    (31, 512, 256, 2, 0.0, 0.0, 1),               # less than one block of rows
])
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["auto", "64rows-4waves", "32rows-4waves"])
def test_fused_feed_forward_precision6(rows, d, ff, act, p_in, p_out, nout, mode):
    """mode: the block shape (oe_ffn6_config) - automatic = two staggered wave groups wherever ff % 256 == 0."""
    if mode and d != 256:
        pytest.skip("block shapes other than the automatic one exist at d = 256 only")
    hip.lib().oe_ffn6_config(mode)
    try:
        _fwd_case(rows, d, ff, act, p_in, p_out, nout)
    finally:
        hip.lib().oe_ffn6_config(0)


def _fwd_case(rows, d, ff, act, p_in, p_out, nout):
    torch.manual_seed(66)
    x = torch.randn(rows, d)
    w1, b1 = torch.randn(ff, d) / math.sqrt(d), torch.randn(ff) * 0.1
    w2, b2 = torch.randn(d, ff) / math.sqrt(ff), torch.randn(d) * 0.1
    res = torch.randn(rows, d)
    beta, s_in, s_out = 0.5, 0x1111, 0x2222
    L = hip.lib()
    assert L.oe_ffn_supported(d, ff, 6, act)
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    xd, w1d, w2d, b1d, b2d, resd = cu(x), cu(w1), cu(w2), cu(b1), cu(b2), cu(res)
    hip.call("oe_ffn_pack_weights", w1d, w2d, d, ff, 6, w1p, w2p)
    pre = torch.full((rows, ff), float("nan"), device=DEV) if nout >= 1 else None
    aout = torch.full((rows, ff), float("nan"), device=DEV) if nout == 2 else None
    y = torch.full((rows, d), float("nan"), device=DEV)
    ctr = torch.tensor([3], dtype=torch.int64, device=DEV)
    hip.ffn_fwd(xd, w1p, b1d, w2p, b2d, rows, d, ff, act, drop_in=p_in, seed_in=s_in, drop_out=p_out, seed_out=s_out, seed_dev=ctr,
                pre_out=pre, act_out=aout, residual=resd, ldr=d, beta=beta, y=y, precision=6)
    y0, pre0, a0 = _unfused_fp32(xd, w1d, b1d, w2d, b2d, resd, act, beta, p_in, s_in, p_out, s_out, ctr)
    ones_in, ones_out = torch.ones(rows, ff, device=DEV), torch.ones(rows, d, device=DEV)
    m_in, m_out = torch.empty_like(ones_in), torch.empty_like(ones_out)
    hip.call("oe_dropout_scale", ones_in, ones_in.numel(), ff, 1.0, p_in, s_in, ctr, None, m_in)
    hip.call("oe_dropout_scale", ones_out, ones_out.numel(), d, 1.0, p_out, s_out, ctr, None, m_out)
    sync()
    h = x.double() @ w1.double().t() + b1.double()
    a = (h * torch.sigmoid(h) if act == 2 else h.clamp(min=0) if act == 1 else h) * m_in.cpu().double()
    want = res.double() + beta * ((a @ w2.double().t() + b2.double()) * m_out.cpu().double())

    def err(got, ref):
        return float((got.cpu().double() - ref).abs().max())
    e6, e0 = err(y, want), err(y0, want)
    assert not torch.isnan(y).any()
    assert e6 <= 2.5 * e0 + 1e-6 * float(want.abs().max()), (e6, e0)         # at the fp32-input kernel's own error
    torch.testing.assert_close(y.cpu().double(), want, rtol=1e-4, atol=5e-5)  # mode 0's module tolerance
    if nout >= 1:
        assert err(pre, h) <= 2.5 * err(pre0, h) + 1e-6 * float(h.abs().max())
    if nout == 2:
        assert err(aout, a) <= 2.5 * err(a0, a) + 1e-6 * float(a.abs().max())
        if p_in > 0:                                                          # bit-identical masks to oe_gemm_f32's epilogue
            assert float(((aout == 0) != (m_in == 0)).float().mean()) < 1e-4  # (a itself can be exactly 0)


@pytest.mark.parametrize("rows,d,ff,act,p_in", [(7936, 256, 1024, 2, 0.1), (333, 256, 512, 1, 0.0), (100, 128, 256, 2, 0.2), (64, 128, 128, 0, 0.0),
                                                  (1100, 512, 2048, 2, 0.1)])
@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "64rows-4waves"])
def test_fused_feed_forward_input_gradient_precision6(rows, d, ff, act, p_in, mode):
    """oe_ffn_bwd in precision 6: dH = (dY W2) * mask * act'(pre), dX = dH W1 - against float64 and against the two
    oe_gemm_f32 launches of the unfused backward on the fp32-input MFMA."""
    if mode and d != 256:
        pytest.skip("block shapes other than the automatic one exist at d = 256 only")
    hip.lib().oe_ffn6_config(mode)
    try:
        _bwd_case(rows, d, ff, act, p_in)
    finally:
        hip.lib().oe_ffn6_config(0)


def _bwd_case(rows, d, ff, act, p_in):
    torch.manual_seed(67)
    dy = torch.randn(rows, d)
    w1, w2 = torch.randn(ff, d) / math.sqrt(d), torch.randn(d, ff) / math.sqrt(ff)
    pre = torch.randn(rows, ff) * 1.5
    s_in = 0x3333
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w2tp, w1tp = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    dyd, w1d, w2d, pred = cu(dy), cu(w1), cu(w2), cu(pre)
    hip.call("oe_ffn_pack_weights_bwd", w1d, w2d, d, ff, 6, w2tp, w1tp)
    dh = torch.full((rows, ff), float("nan"), device=DEV)
    dx = torch.full((rows, d), float("nan"), device=DEV)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    hip.ffn_bwd(dyd, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pred, dh=dh, dx=dx, precision=6)
    # the unfused backward in exact fp32 products
    old = hip.GEMM_PRECISION
    hip.GEMM_PRECISION = 0
    try:
        dh0 = ops.gemm_nn(dyd, w2d, act=act, actgrad_in=pred, ld_aux=ff, drop_p=p_in, seed=s_in, seed_dev=ctr)
        dx0 = ops.gemm_nn(dh0, w1d)
    finally:
        hip.GEMM_PRECISION = old
    ones = torch.ones(rows, ff, device=DEV)
    m_in = torch.empty_like(ones)
    hip.call("oe_dropout_scale", ones, ones.numel(), ff, 1.0, p_in, s_in, ctr, None, m_in)
    sync()
    h = pre.double()
    sg = torch.sigmoid(h)
    dact = sg * (1 + h * (1 - sg)) if act == 2 else (h > 0).double() if act == 1 else torch.ones_like(h)
    want_dh = (dy.double() @ w2.double()) * m_in.cpu().double() * dact
    want_dx = want_dh @ w1.double()

    def err(got, ref):
        return float((got.cpu().double() - ref).abs().max())
    assert not torch.isnan(dh).any() and not torch.isnan(dx).any()
    assert err(dh, want_dh) <= 2.5 * err(dh0, want_dh) + 1e-6 * float(want_dh.abs().max())
    assert err(dx, want_dx) <= 2.5 * err(dx0, want_dx) + 1e-6 * float(want_dx.abs().max())
    torch.testing.assert_close(dx.cpu().double(), want_dx, rtol=1e-4, atol=5e-5 * max(1.0, float(want_dx.abs().max())))


def test_feed_forward_op_fused_equals_unfused_in_precision6():
    """ops.feed_forward (the autograd function the encoder layers call) with the fused forward and the fused input gradient forced
    on, against the same op on the two-GEMM path: output, input gradient, every parameter gradient."""
    from openeat_amd import planes
    old = (hip.GEMM_PRECISION, ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_BWD, ops.FUSED_FFN)
    hip.GEMM_PRECISION = 6
    torch.manual_seed(68)
    rows, d, ff = 700, 256, 1024
    x0 = torch.randn(rows, d, device=DEV)
    w1 = (torch.randn(ff, d, device=DEV) / 16).requires_grad_()
    b1 = (torch.randn(ff, device=DEV) * 0.1).requires_grad_()
    w2 = (torch.randn(d, ff, device=DEV) / 32).requires_grad_()
    b2 = (torch.randn(d, device=DEV) * 0.1).requires_grad_()
    wgt = torch.randn(rows, d, device=DEV)
    out = {}
    try:
        for fused in (False, True):
            ops.FUSED_FFN, ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_BWD = fused, 0, fused
            planes.clear_all()
            ops.manual_seed(9)
            x = x0.clone().requires_grad_()
            for t in (w1, b1, w2, b2):
                t.grad = None
            y = ops.feed_forward(x, w1, b1, w2, b2, ops.ACT_SWISH, p_in=0.1, residual=x, out_scale=0.5, p_out=0.1)
            (y * wgt).sum().backward()
            ops.join_side_stream()
            sync()
            out[fused] = [y.detach().clone(), x.grad.clone()] + [t.grad.clone() for t in (w1, b1, w2, b2)]
    finally:
        hip.GEMM_PRECISION, ops.FUSED_FFN_MIN_ROWS, ops.FUSED_FFN_BWD, ops.FUSED_FFN = old
        planes.clear_all()
    for a, b, name in zip(out[True], out[False], ("y", "dx", "dw1", "db1", "dw2", "db2")):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-5 * max(1.0, float(b.abs().max())), msg=lambda m, name=name: f"{name}: {m}")


def test_pack_table_rows_keep_their_own_precision():
    """bench.py runs the same process in precision 6, then 0, 3, 1 and back in 6 (decode): feed-forwards registered under
    precision 3 / 1 have packed buffers of two / one plane.  A refresh launched in precision 6 walks ALL rows of the table - each
    with the plane count it was registered with (a global count overran the smaller buffers: GPU memory fault in the r04 bundle)."""
    from openeat_amd import planes
    old = (hip.GEMM_PRECISION, ops.FUSED_FFN_MIN_ROWS)
    ops.FUSED_FFN_MIN_ROWS = 0
    torch.manual_seed(69)
    rows, d, ff = 256, 256, 512
    x = torch.randn(rows, d, device=DEV)

    def params():
        return (torch.nn.Parameter(torch.randn(ff, d, device=DEV) / 16), torch.nn.Parameter(torch.randn(ff, device=DEV) * 0.1),
                torch.nn.Parameter(torch.randn(d, ff, device=DEV) / 22), torch.nn.Parameter(torch.randn(d, device=DEV) * 0.1))

    def ref(p):
        h = x.double() @ p[0].detach().double().t() + p[1].detach().double()
        return (h * torch.sigmoid(h)) @ p[2].detach().double().t() + p[3].detach().double()
    sets = {3: [params() for _ in range(2)], 1: [params() for _ in range(2)], 6: [params() for _ in range(2)]}
    try:
        with torch.no_grad():
            for prec in (3, 1, 6, 3, 6):                                  # every switch is a new generation: a table refresh over all rows
                hip.GEMM_PRECISION = prec
                planes.new_pass()
                for p in sets[prec]:
                    y = ops.feed_forward(x, p[0], p[1], p[2], p[3], ops.ACT_SWISH)
                    tol = dict(rtol=1e-4, atol=5e-5) if prec == 6 else dict(rtol=2e-4, atol=2e-4) if prec == 3 else dict(rtol=5e-2, atol=5e-2)
                    torch.testing.assert_close(y.double().cpu(), ref(p).cpu(), **tol)
        assert len(ops._FFN.entries) == 6
        planes_of = {k[2]: int(ops._FFN.host[i, 8]) for k, i in ops._FFN.rows.items()}
        assert planes_of == {3: 2, 1: 1, 6: 3}
    finally:
        hip.GEMM_PRECISION, ops.FUSED_FFN_MIN_ROWS = old
        planes.clear_all()


@pytest.mark.parametrize("rows,p,ln_mask", [(7936, 0.1, False), (4103, 0.0, True)])
def test_input_gradient_with_the_layernorm_backward_as_its_prologue(rows, p, ln_mask):
    """oe_ffn_bwd with oe_ffn_args.ln set (precision 6, d = 256, the two-group block shape): the rows of dY are made by the LayerNorm
    backward that used to be a launch in front of it - dx_ln and g as oe_layernorm_bwd_dx_drop writes them (to one unit in the last
    place: another kernel, another contraction of a*b+c), dH / dX as the plain oe_ffn_bwd gives on that g."""
    torch.manual_seed(68)
    d, ff, act, p_in, s_in = 256, 1024, 2, 0.1, 0x3333
    x = torch.randn(rows, d, device=DEV) * 1.3 - 0.2
    dy_ln = torch.randn(rows, d, device=DEV)
    add = torch.randn(rows, d, device=DEV)
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    lmask = (torch.rand(rows, device=DEV) > 0.15).to(torch.uint8) if ln_mask else None
    stats = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    w1, w2 = torch.randn(ff, d, device=DEV) / math.sqrt(d), torch.randn(d, ff, device=DEV) / math.sqrt(ff)
    pre = torch.randn(rows, ff, device=DEV) * 1.5
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w2tp, w1tp = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.call("oe_ffn_pack_weights_bwd", w1, w2, d, ff, 6, w2tp, w1tp)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    nws = L.oe_layernorm_bwd_workspace_floats(rows, d)
    # two launches
    dxl0, g0, ws0 = torch.empty_like(x), torch.empty_like(x), torch.zeros(nws, device=DEV)
    hip.call("oe_layernorm_bwd_dx_drop", dy_ln, x, gamma, beta, 0, stats, rows, d, lmask, add, dxl0, g0, 0.5, p, 0x77, ctr, None, ws0)
    dh0, dx0 = torch.empty(rows, ff, device=DEV), torch.empty(rows, d, device=DEV)
    hip.ffn_bwd(g0, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pre, dh=dh0, dx=dx0, precision=6)
    # one launch
    nan = float("nan")
    dxl1, g1, ws1 = torch.full_like(x, nan), torch.full_like(x, nan), torch.zeros(nws, device=DEV)
    dh1, dx1 = torch.full((rows, ff), nan, device=DEV), torch.full((rows, d), nan, device=DEV)
    ln = dict(dy=dy_ln, x=x, stats=stats, gamma=gamma, add=add, dx=dxl1, g=g1, ws=ws1, alpha=0.5, p=p, seed=0x77, rowmask=None, ln_rowmask=lmask)
    hip.ffn_bwd(g1, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pre, dh=dh1, dx=dx1, precision=6, ln=ln)
    dg0, db0, dg1, db1 = (torch.zeros(d, device=DEV) for _ in range(4))
    hip.call("oe_layernorm_param_reduce", ws0, rows, d, dg0, db0)
    hip.call("oe_layernorm_param_reduce", ws1, rows, d, dg1, db1)
    sync()
    torch.testing.assert_close(dxl1, dxl0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(g1, g0, rtol=2e-6, atol=1e-6)
    assert torch.equal(g1 == 0, g0 == 0)
    torch.testing.assert_close(dh1, dh0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx1, dx0, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(dg1, dg0, rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(db1, db0, rtol=2e-5, atol=2e-4)
    # the prologue exists for this block shape only: any other is refused, not silently run without it
    L.oe_ffn6_config(1)
    try:
        with pytest.raises(RuntimeError, match="prologue"):
            hip.ffn_bwd(g1, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pre, dh=dh1, dx=dx1, precision=6, ln=ln)
    finally:
        L.oe_ffn6_config(0)


def test_input_gradient_with_a_pair_of_layernorm_backwards_as_its_prologue():
    """The PAIR form of the prologue (oe_ln_prologue.gamma2): norm_final of an encoder layer and the norm behind it, whose joint
    backward (oe_layernorm_pair_bwd_dx_drop) precedes the second feed-forward's backward."""
    torch.manual_seed(69)
    rows, d, ff, act, p_in, s_in, p = 4999, 256, 512, 2, 0.0, 0, 0.1
    x = torch.randn(rows, d, device=DEV) * 1.3 - 0.2
    g1, b1 = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    g2, b2 = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    u, y2 = torch.empty_like(x), torch.empty_like(x)
    st1, st2 = torch.empty(rows, 2, device=DEV), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_pair_fwd", x, g1, b1, 1e-5, g2, b2, 1e-5, rows, d, u, st1, y2, st2)
    dy, du = torch.randn(rows, d, device=DEV), torch.randn(rows, d, device=DEV)
    w1, w2 = torch.randn(ff, d, device=DEV) / math.sqrt(d), torch.randn(d, ff, device=DEV) / math.sqrt(ff)
    pre = torch.randn(rows, ff, device=DEV) * 1.5
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w2tp, w1tp = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.call("oe_ffn_pack_weights_bwd", w1, w2, d, ff, 6, w2tp, w1tp)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    nws = L.oe_layernorm_bwd_workspace_floats(rows, d)
    dxl0, gq0 = torch.empty_like(x), torch.empty_like(x)
    wsa0, wsb0 = torch.zeros(nws, device=DEV), torch.zeros(nws, device=DEV)
    hip.call("oe_layernorm_pair_bwd_dx_drop", dy, x, g1, b1, st1, g2, st2, rows, d, du, dxl0, gq0, 0.5, p, 0x99, ctr, None, wsa0, wsb0)
    dh0, dx0 = torch.empty(rows, ff, device=DEV), torch.empty(rows, d, device=DEV)
    hip.ffn_bwd(gq0, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pre, dh=dh0, dx=dx0, precision=6)
    nan = float("nan")
    dxl1, gq1 = torch.full_like(x, nan), torch.full_like(x, nan)
    wsa1, wsb1 = torch.zeros(nws, device=DEV), torch.zeros(nws, device=DEV)
    dh1, dx1 = torch.full((rows, ff), nan, device=DEV), torch.full((rows, d), nan, device=DEV)
    ln = dict(dy=dy, x=x, stats=st1, gamma=g1, beta=b1, gamma2=g2, stats2=st2, add=du, dx=dxl1, g=gq1, ws=wsa1, ws2=wsb1, alpha=0.5, p=p, seed=0x99,
              rowmask=None, ln_rowmask=None)
    hip.ffn_bwd(gq1, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pre, dh=dh1, dx=dx1, precision=6, ln=ln)
    outs = []
    for wsa, wsb in ((wsa0, wsb0), (wsa1, wsb1)):
        r = [torch.zeros(d, device=DEV) for _ in range(4)]
        hip.call("oe_layernorm_param_reduce", wsa, rows, d, r[0], r[1])
        hip.call("oe_layernorm_param_reduce", wsb, rows, d, r[2], r[3])
        outs.append(r)
    sync()
    torch.testing.assert_close(dxl1, dxl0, rtol=3e-6, atol=2e-6)
    torch.testing.assert_close(gq1, gq0, rtol=3e-6, atol=2e-6)
    assert torch.equal(gq1 == 0, gq0 == 0)
    torch.testing.assert_close(dh1, dh0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx1, dx0, rtol=1e-5, atol=2e-5)
    for a, b in zip(outs[1], outs[0]):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=3e-4)


def test_forward_with_the_layernorm_as_its_prologue():
    """oe_ffn_fwd with oe_ffn_args.lnf set: the pre-norm in front of the feed-forward (encoder_layer.py:79-83, 103-106) computed on the
    way into the kernel - y_ln, the statistics and every output as oe_layernorm_fwd + oe_ffn_fwd give them (one unit in the last place)."""
    torch.manual_seed(70)
    rows, d, ff, act = 4133, 256, 1024, 2
    x = torch.randn(rows, d, device=DEV) * 1.4 + 0.2
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    w1, b1 = torch.randn(ff, d, device=DEV) / math.sqrt(d), torch.randn(ff, device=DEV) * 0.1
    w2, b2 = torch.randn(d, ff, device=DEV) / math.sqrt(ff), torch.randn(d, device=DEV) * 0.1
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.call("oe_ffn_pack_weights", w1, w2, d, ff, 6, w1p, w2p)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    kw = dict(drop_in=0.1, seed_in=11, drop_out=0.1, seed_out=12, seed_dev=ctr, residual=x, ldr=d, beta=0.5, precision=6)
    yl0, st0 = torch.empty_like(x), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_fwd", x, gamma, beta, 1e-5, rows, d, None, 0, yl0, st0)
    pre0, a0, y0 = torch.empty(rows, ff, device=DEV), torch.empty(rows, ff, device=DEV), torch.empty(rows, d, device=DEV)
    hip.ffn_fwd(yl0, w1p, b1, w2p, b2, rows, d, ff, act, pre_out=pre0, act_out=a0, y=y0, **kw)
    nan = float("nan")
    yl1, st1 = torch.full_like(x, nan), torch.full((rows, 2), nan, device=DEV)
    pre1, a1, y1 = torch.full((rows, ff), nan, device=DEV), torch.full((rows, ff), nan, device=DEV), torch.full((rows, d), nan, device=DEV)
    lnf = dict(x=x, gamma=gamma, beta=beta, eps=1e-5, y=yl1, stats=st1, rowmask=None)
    hip.ffn_fwd(yl1, w1p, b1, w2p, b2, rows, d, ff, act, pre_out=pre1, act_out=a1, y=y1, lnf=lnf, **kw)
    sync()
    torch.testing.assert_close(yl1, yl0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(st1, st0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(pre1, pre0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a1, a0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=2e-5)
    L.oe_ffn6_config(1)                                                  # another block shape: refused, not run without the norm
    try:
        with pytest.raises(RuntimeError, match="prologue"):
            hip.ffn_fwd(yl1, w1p, b1, w2p, b2, rows, d, ff, act, pre_out=pre1, act_out=a1, y=y1, lnf=lnf, **kw)
    finally:
        L.oe_ffn6_config(0)


def test_forward_with_a_pair_of_layernorms_as_its_prologue():
    """The PAIR form of oe_ffn_args.lnf: norm_final of an encoder layer and the next layer's first pre-norm (oe_layernorm_pair_fwd) in
    front of that layer's first feed-forward, whose residual IS the first norm's output - written by the kernel's prologue and read
    back by its own epilogue."""
    torch.manual_seed(71)
    rows, d, ff, act = 4157, 256, 1024, 2
    x = torch.randn(rows, d, device=DEV) * 1.4 + 0.2
    g1, b1n = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    g2, b2n = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    w1, b1 = torch.randn(ff, d, device=DEV) / math.sqrt(d), torch.randn(ff, device=DEV) * 0.1
    w2, b2 = torch.randn(d, ff, device=DEV) / math.sqrt(ff), torch.randn(d, device=DEV) * 0.1
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, 6)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    hip.call("oe_ffn_pack_weights", w1, w2, d, ff, 6, w1p, w2p)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    u0, y20 = torch.empty_like(x), torch.empty_like(x)
    sa0, sb0 = torch.empty(rows, 2, device=DEV), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_pair_fwd", x, g1, b1n, 1e-5, g2, b2n, 1e-5, rows, d, u0, sa0, y20, sb0)
    kw = dict(drop_in=0.1, seed_in=11, drop_out=0.1, seed_out=12, seed_dev=ctr, ldr=d, beta=0.5, precision=6)
    pre0, a0, out0 = torch.empty(rows, ff, device=DEV), torch.empty(rows, ff, device=DEV), torch.empty(rows, d, device=DEV)
    hip.ffn_fwd(y20, w1p, b1, w2p, b2, rows, d, ff, act, pre_out=pre0, act_out=a0, y=out0, residual=u0, **kw)
    nan = float("nan")
    u1, y21 = torch.full_like(x, nan), torch.full_like(x, nan)
    sa1, sb1 = torch.full((rows, 2), nan, device=DEV), torch.full((rows, 2), nan, device=DEV)
    pre1, a1, out1 = torch.full((rows, ff), nan, device=DEV), torch.full((rows, ff), nan, device=DEV), torch.full((rows, d), nan, device=DEV)
    lnf = dict(x=x, gamma=g1, beta=b1n, eps=1e-5, gamma2=g2, beta2=b2n, eps2=1e-5, u=u1, y=y21, stats=sa1, stats2=sb1, rowmask=None)
    for _ in range(3):                                   # (the residual read-back: stable over launches)
        hip.ffn_fwd(y21, w1p, b1, w2p, b2, rows, d, ff, act, pre_out=pre1, act_out=a1, y=out1, residual=u1, lnf=lnf, **kw)
        sync()
        torch.testing.assert_close(u1, u0, rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(y21, y20, rtol=3e-6, atol=2e-6)
        torch.testing.assert_close(sa1, sa0, rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(sb1, sb0, rtol=3e-6, atol=2e-6)
        torch.testing.assert_close(pre1, pre0, rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(out1, out0, rtol=1e-5, atol=3e-5)
