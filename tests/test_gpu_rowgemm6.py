"""GPU: the short-reduction Linears as row-block GEMMs in precision 6 (csrc/ffn6.hip::rowgemm6_kernel through oe_rowgemm6 /
oe_rowgemm6_pack_table) - attention.py:56-58,97 (linear_q / k / v / out), convolution.py:79-111 (pointwise convs) and their
input gradients - against float64 and against oe_gemm_f32 on the fp32-input MFMA (precision 0) with the same epilogue."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from openeat_amd import hip, ops, planes  # noqa: E402

DEV = "cuda"


@pytest.fixture(autouse=True)
def mode6():
    old = hip.GEMM_PRECISION
    hip.GEMM_PRECISION = 6
    planes.clear_all()
    yield
    hip.GEMM_PRECISION = old
    planes.clear_all()


def _err(got, ref):
    return float((got.cpu().double() - ref).abs().max())


@pytest.mark.parametrize("rows,k,n,bias,p,res,mask,beta", [
    (7936, 256, 768, True, 0.0, False, False, 1.0),        # fused q / k / v projection
    (7936, 256, 256, True, 0.1, True, False, 1.0),         # linear_out: bias, dropout, residual
    (7936, 256, 256, True, 0.1, True, True, 1.0),          # pointwise_conv2: + zeroed pad rows
    (7936, 256, 512, True, 0.0, False, False, 1.0),        # pointwise_conv1
    (4097, 256, 128, False, 0.0, True, False, 0.5),        # one chunk (the second wave group idles), ragged last block, scaled residual
    (4500, 512, 1536, True, 0.2, False, False, 1.0),       # configs[4] width
    (4100, 768, 256, True, 0.1, True, True, 0.5),          # K-phased kernel: three phases of 256, both groups
    (4097, 1024, 128, False, 0.0, False, False, 1.0),      # ... four phases, one chunk (the second group only stages rows), ragged
])
def test_x_wT_matches_float64_and_the_fp32_kernel(rows, k, n, bias, p, res, mask, beta):
    torch.manual_seed(71)
    x = torch.randn(rows, k)
    w = torch.nn.Parameter(torch.randn(n, k) / math.sqrt(k))
    b = torch.randn(n) * 0.1 if bias else None
    r = torch.randn(rows, n) if res else None
    m = (torch.rand(rows) > 0.1).to(torch.uint8) if mask else None
    xd, wd = x.to(DEV), torch.nn.Parameter(w.detach().to(DEV))
    bd, rd, md = (None if t is None else t.to(DEV) for t in (b, r, m))
    ctr = torch.tensor([7], dtype=torch.int64, device=DEV)
    epi = dict(drop_p=p, seed=0x4444, seed_dev=ctr, rowmask=md, residual=rd, ldr=n if res else 0, beta=beta)
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        y = ops.gemm_nt(xd, wd, bd, **epi)
    assert ops.ROWGEMM_LAUNCHES == n0 + 1                              # the row-block kernel took it
    y0 = torch.empty(rows, n, device=DEV)
    hip.gemm(xd, wd.detach(), y0, rows, n, k, lda=k, ldb=k, ldc=n, bias=bd, precision=0, **epi)
    ones = torch.ones(rows, n, device=DEV)
    dm = torch.empty_like(ones)
    hip.call("oe_dropout_scale", ones, ones.numel(), n, 1.0, p, 0x4444, ctr, None, dm)
    torch.cuda.synchronize()
    want = x.double() @ w.detach().double().t()
    if bias:
        want = want + b.double()
    want = want * dm.cpu().double()
    if mask:
        want = want * m.double().unsqueeze(1)
    want = (r.double() if res else 0.0) + beta * want
    e6, e0 = _err(y, want), _err(y0, want)
    assert e6 <= 2.5 * e0 + 1e-6 * float(want.abs().max()), (e6, e0)
    torch.testing.assert_close(y.cpu().double(), want, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("rows,n_fwd,k_fwd,mask", [(7936, 256, 256, False), (7936, 512, 256, True), (7936, 256, 512, False), (5000, 512, 1536, False),
                                                  (7936, 768, 256, False), (4999, 768, 256, True)])      # the fused q / k / v projection's input gradient
def test_dy_w_matches_float64_and_the_fp32_kernel(rows, n_fwd, k_fwd, mask):
    """The input gradient dx = dy W of a Linear with weight W (n_fwd, k_fwd): reduction over n_fwd (256 / 512), output k_fwd wide."""
    torch.manual_seed(72)
    dy = torch.randn(rows, n_fwd)
    w = torch.randn(n_fwd, k_fwd) / math.sqrt(k_fwd)
    m = (torch.rand(rows) > 0.1).to(torch.uint8) if mask else None
    dyd, wd = dy.to(DEV), torch.nn.Parameter(w.to(DEV))
    md = None if m is None else m.to(DEV)
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        dx = ops.gemm_nn(dyd, wd, rowmask=md)
    assert ops.ROWGEMM_LAUNCHES == n0 + 1
    dx0 = torch.empty(rows, k_fwd, device=DEV)
    hip.gemm(dyd, wd.detach(), dx0, rows, k_fwd, n_fwd, lda=n_fwd, ldb=k_fwd, ldc=k_fwd, b_kmajor=True, rowmask=md, precision=0)
    torch.cuda.synchronize()
    want = dy.double() @ w.double()
    if mask:
        want = want * m.double().unsqueeze(1)
    assert _err(dx, want) <= 2.5 * _err(dx0, want) + 1e-6 * float(want.abs().max())
    torch.testing.assert_close(dx.cpu().double(), want, rtol=1e-4, atol=5e-5)


def test_packed_weights_follow_the_weights():
    """A weight written through torch (version counter) or behind torch's back (planes.bump_generation: what FusedAdam / a graph
    replay / a new pass announce) is re-packed before its next use; an untouched one is not packed again."""
    torch.manual_seed(73)
    x = torch.randn(4096, 256, device=DEV)
    w = torch.nn.Parameter(torch.randn(256, 256, device=DEV) / 16)
    with torch.no_grad():
        y1 = ops.gemm_nt(x, w)
        w.mul_(2.0)                                                   # torch write: version counter
        y2 = ops.gemm_nt(x, w)
        w.data.view(-1)[:].copy_((w.data * 0.25).view(-1))            # (also a torch write)
        planes.bump_generation()
        y3 = ops.gemm_nt(x, w)
    torch.cuda.synchronize()
    torch.testing.assert_close(y2, 2.0 * y1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y3, 0.5 * y1, rtol=1e-5, atol=1e-5)


def test_unsupported_epilogues_and_shapes_stay_on_oe_gemm_f32():
    x = torch.randn(4096, 256, device=DEV)
    w = torch.nn.Parameter(torch.randn(384 + 6, 256, device=DEV) / 16)           # n = 390: not a multiple of 128
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        ops.gemm_nt(x, w)
        w2 = torch.nn.Parameter(torch.randn(256, 256, device=DEV) / 16)
        pre = torch.empty(4096, 256, device=DEV)
        ops.gemm_nt(x, w2, None, act=ops.ACT_SWISH, preact_out=pre, ld_aux=256)   # an activation epilogue
        ops.gemm_nt(x[:3000], w2)                                               # below ROWGEMM_MIN_ROWS, above the tile form's rows
    assert ops.ROWGEMM_LAUNCHES == n0


def test_temporaries_are_never_cached():
    """A model without a parameter arena concatenates q / k / v weights into a TEMPORARY for the fused projection (ops._fused_rows):
    it dies after the call, and the next layer's temporary lands on the same address with other values and the same tensor version.
    Such weights are not registered (their pack would be taken for fresh: the B = 32 whole-model parity test caught it as 2-9 % errors
    in the decoders' gradient norms)."""
    torch.manual_seed(74)
    x = torch.randn(4096, 256, device=DEV)
    wa, wb = torch.randn(256, 256, device=DEV) / 16, torch.randn(256, 256, device=DEV) / 16
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        ta = torch.cat([wa[:128], wa[128:]], 0)
        pa = ta.data_ptr()
        ya = ops.gemm_nt(x, ta)
        del ta
        tb = torch.cat([wb[:128], wb[128:]], 0)
        same_address = tb.data_ptr() == pa
        yb = ops.gemm_nt(x, tb)
    torch.cuda.synchronize()
    assert ops.ROWGEMM_LAUNCHES == n0                                  # temporaries stay on oe_gemm_f32
    torch.testing.assert_close(ya, x @ wa.t(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(yb, x @ wb.t(), rtol=1e-4, atol=1e-4)
    assert same_address or True                                        # (the hazard's precondition on this allocator; informational)


def test_captured_refresh_survives_later_registrations():
    """A captured graph bakes in the address of the pack table and of the packed buffers.  The table is therefore ONE persistent
    tensor updated in place, append-only (the first version rebuilt it at every registration: bench.py's decode graphs replayed a
    pack launch over a freed table and the GPU faulted at address 0x1e000).  Here: capture a pass, register more weights, let a
    registered weight die, churn the allocator, change the captured weight - the replay must follow."""
    import gc
    torch.manual_seed(75)
    x = torch.randn(4096, 256, device=DEV)
    w = torch.nn.Parameter(torch.randn(256, 256, device=DEV) / 16)
    with torch.no_grad():
        y_ref = ops.gemm_nt(x, w).clone()                              # eager: registers w, uploads the table
        xs = x.clone()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with planes.capture_scope(), torch.cuda.graph(g):
            planes.new_pass()                                          # (what ASRModel.forward / _encode announce)
            ys = ops.gemm_nt(xs, w)
        table_ptr = ops._ROW.dev.data_ptr()
        others = [torch.nn.Parameter(torch.randn(256, 256, device=DEV) / 16) for _ in range(6)]
        for o in others:
            ops.gemm_nt(x, o)                                          # later registrations
        del others[2:]                                                 # ... and some of them die
        gc.collect()
        junk = [torch.full((n,), -1, dtype=torch.int64, device=DEV) for n in (6, 12, 48, 96, 384, 4096 * 6) for _ in range(4)]
        planes.new_pass()
        ops.gemm_nt(x, others[0])                                      # a refresh that sweeps the dead rows
        w.mul_(2.0)
        g.replay()
        torch.cuda.synchronize()
    assert ops._ROW.dev.data_ptr() == table_ptr and junk
    torch.testing.assert_close(ys, 2.0 * y_ref, rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------------------------------------------------------- #
# tile form (oe_rowgemm6_form = 2): few rows, one 32 x 32 output tile per block, the eight waves split the reduction
# --------------------------------------------------------------------------------------------------------------------------- #
def test_form_table():
    L = hip.lib()
    assert L.oe_rowgemm6_form(992, 256, 256) == 2 and L.oe_rowgemm6_form(248, 256, 256) == 2 and L.oe_rowgemm6_form(5, 1024, 96) == 2
    assert L.oe_rowgemm6_form(992, 768, 256) == 2 and L.oe_rowgemm6_form(992, 512, 1024) == 2
    assert L.oe_rowgemm6_form(992, 384, 256) == 0 and L.oe_rowgemm6_form(992, 256, 3246) == 0 and L.oe_rowgemm6_form(992, 2048, 256) == 0
    assert L.oe_rowgemm6_form(7936, 256, 256) == 1 and L.oe_rowgemm6_form(7936, 1024, 256) == 1 and L.oe_rowgemm6_form(7936, 768, 512) == 0 and L.oe_rowgemm6_form(7936, 256, 96) == 0
    assert L.oe_rowgemm6_form(0, 256, 256) != 2


@pytest.mark.parametrize("rows,k,n,bias,p,res,mask,beta,act", [
    (992, 256, 768, True, 0.0, False, False, 1.0, 0),        # decoder self-attention q / k / v (decoder_layer.py:82-92)
    (992, 256, 256, True, 0.1, True, False, 1.0, 0),         # attention output: bias, dropout, residual
    (992, 1024, 256, True, 0.1, True, True, 0.5, 0),         # feed-forward w_2: k = 1024, scaled residual, dead rows
    (992, 256, 1024, True, 0.1, False, False, 1.0, 1),       # feed-forward w_1 + relu + dropout, pre-activation kept
    (992, 256, 1024, True, 0.0, False, False, 1.0, 2),       # ... + swish
    (248, 256, 256, False, 0.0, False, False, 1.0, 0),       # linear_pos (attention.py:166-171): no bias
    (37, 512, 96, True, 0.2, True, False, 1.0, 0),           # ragged single row block, n = 3 tiles
    (2048, 768, 64, False, 0.0, False, False, 1.0, 0),       # k = 768
])
def test_tile_form_x_wT_matches_float64_and_the_fp32_kernel(rows, k, n, bias, p, res, mask, beta, act):
    torch.manual_seed(81)
    x = torch.randn(rows, k)
    w = torch.randn(n, k) / math.sqrt(k)
    b = torch.randn(n) * 0.1 if bias else None
    r = torch.randn(rows, n) if res else None
    m = (torch.rand(rows) > 0.1).to(torch.uint8) if mask else None
    xd, wd = x.to(DEV), torch.nn.Parameter(w.to(DEV))
    bd, rd, md = (None if t is None else t.to(DEV) for t in (b, r, m))
    ctr = torch.tensor([9], dtype=torch.int64, device=DEV)
    pre = torch.full((rows, n), float("nan"), device=DEV) if act else None
    pre0 = torch.empty(rows, n, device=DEV) if act else None
    epi = dict(drop_p=p, seed=0x5151, seed_dev=ctr, rowmask=md, residual=rd, ldr=n if res else 0, beta=beta, act=act, ld_aux=n if act else 0)
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        y = ops.gemm_nt(xd, wd, bd, preact_out=pre, **epi)
    assert ops.ROWGEMM_LAUNCHES == n0 + 1
    y0 = torch.empty(rows, n, device=DEV)
    hip.gemm(xd, wd.detach(), y0, rows, n, k, lda=k, ldb=k, ldc=n, bias=bd, precision=0, preact_out=pre0, **epi)
    ones = torch.ones(rows, n, device=DEV)
    dm = torch.empty_like(ones)
    hip.call("oe_dropout_scale", ones, ones.numel(), n, 1.0, p, 0x5151, ctr, None, dm)
    torch.cuda.synchronize()
    want = x.double() @ w.double().t()
    if bias:
        want = want + b.double()
    want_pre = want
    if act == 1:
        want = want.clamp_min(0.0)
    elif act == 2:
        want = want * torch.sigmoid(want)
    want = want * dm.cpu().double()
    if mask:
        want = want * m.double().unsqueeze(1)
    want = (r.double() if res else 0.0) + beta * want
    if act == 1:
        # a pre-activation within rounding of zero may take the other branch: compare where it is clearly signed
        sure = (want_pre.abs() > 1e-5)
        assert float(((y.cpu().double() - want) * sure).abs().max()) < 5e-5
    else:
        e6, e0 = _err(y, want), _err(y0, want)
        assert e6 <= 2.5 * e0 + 1e-6 * float(want.abs().max()), (e6, e0)
        torch.testing.assert_close(y.cpu().double(), want, rtol=1e-4, atol=5e-5)
    if act:
        torch.testing.assert_close(pre.cpu().double(), want_pre, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("rows,n_fwd,k_fwd,act,p", [(992, 256, 256, 0, 0.0), (992, 1024, 256, 0, 0.0), (992, 256, 1024, 1, 0.1), (992, 256, 1024, 2, 0.1),
                                                   (992, 768, 256, 0, 0.0), (61, 512, 256, 0, 0.0)])
def test_tile_form_dy_w_matches_float64(rows, n_fwd, k_fwd, act, p):
    """dx = dy W (reduction over n_fwd); with an activation: dH = (dy W2) * mask * act'(pre) - the feed-forward's backward
    through w_2 (positionwise_feed_forward.py:36-43)."""
    torch.manual_seed(82)
    dy = torch.randn(rows, n_fwd)
    w = torch.randn(n_fwd, k_fwd) / math.sqrt(k_fwd)
    pre = torch.randn(rows, k_fwd) if act else None
    dyd, wd = dy.to(DEV), torch.nn.Parameter(w.to(DEV))
    pred = None if pre is None else pre.to(DEV)
    ctr = torch.tensor([3], dtype=torch.int64, device=DEV)
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        dx = ops.gemm_nn(dyd, wd, act=act, actgrad_in=pred, ld_aux=k_fwd if act else 0, drop_p=p, seed=0x77, seed_dev=ctr, out_planes=True)
    assert ops.ROWGEMM_LAUNCHES == n0 + 1
    ones = torch.ones(rows, k_fwd, device=DEV)
    dm = torch.empty_like(ones)
    hip.call("oe_dropout_scale", ones, ones.numel(), k_fwd, 1.0, p, 0x77, ctr, None, dm)
    torch.cuda.synchronize()
    want = dy.double() @ w.double()
    if act == 1:
        want = want * (pre.double() > 0)
    elif act == 2:
        s = torch.sigmoid(pre.double())
        want = want * (s * (1 + pre.double() * (1 - s)))
    want = want * dm.cpu().double()
    torch.testing.assert_close(dx.cpu().double(), want, rtol=1e-4, atol=5e-5)


def test_tile_form_declines_what_it_cannot_do():
    x = torch.randn(992, 256, device=DEV)
    w = torch.nn.Parameter(torch.randn(3246, 256, device=DEV) / 16)              # n not a multiple of 32
    w3 = torch.nn.Parameter(torch.randn(256, 384, device=DEV) / 16)              # k = 384
    w4 = torch.nn.Parameter(torch.randn(256, 256, device=DEV) / 16)
    n0 = ops.ROWGEMM_LAUNCHES
    with torch.no_grad():
        ops.gemm_nt(x, w)
        ops.gemm_nt(torch.randn(992, 384, device=DEV), w3)
        ops.gemm_nt(x, w4, alpha=0.5)                                            # an epilogue key the kernel does not have
    assert ops.ROWGEMM_LAUNCHES == n0


# --------------------------------------------------------------------------------------------------------------------------- #
# LayerNorm-backward prologue (oe_rowgemm_args.ln_*): the rows of the GEMM are made by the LayerNorm backward that preceded it
# --------------------------------------------------------------------------------------------------------------------------- #
@pytest.mark.parametrize("rows,p,with_add,with_mask,ln_mask", [(7936, 0.1, True, False, False), (4101, 0.0, False, False, False),
                                                               (7936, 0.1, True, True, False), (5000, 0.1, True, True, True)])
def test_ln_backward_prologue_equals_the_two_launches(rows, p, with_add, with_mask, ln_mask):
    """oe_layernorm_bwd_dx_drop followed by oe_rowgemm6 against ONE oe_rowgemm6 with the ln_* arguments: dx, g and the product to one
    unit in the last place, the same mask, the parameter-gradient partials to rounding (other grouping of the same sums)."""
    torch.manual_seed(91)
    d = 256
    x = torch.randn(rows, d, device=DEV) * 1.7 + 0.3
    dy = torch.randn(rows, d, device=DEV)
    add = torch.randn(rows, d, device=DEV) if with_add else None
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    gmask = (torch.rand(rows, device=DEV) > 0.1).to(torch.uint8) if with_mask else None
    lmask = (torch.rand(rows, device=DEV) > 0.15).to(torch.uint8) if ln_mask else None      # the LayerNorm's own row mask
    w = torch.nn.Parameter(torch.randn(d, d, device=DEV) / 16)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    mean = x.mean(1)
    rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    stats = torch.stack([mean, rstd], 1).contiguous()
    nws = hip.lib().oe_layernorm_bwd_workspace_floats(rows, d)
    # two launches
    dx0, g0, ws0 = torch.empty_like(x), torch.empty_like(x), torch.zeros(nws, device=DEV)
    old = ops._seed_dev
    ops._seed_dev = ctr
    try:
        hip.call("oe_layernorm_bwd_dx_drop", dy, x, gamma, beta, 0, stats, rows, d, lmask, add, dx0, g0, 0.5, p, 0x1234, ctr, gmask, ws0)
        with torch.no_grad():
            ops.LN_BWD_FUSE, keep = False, ops.LN_BWD_FUSE
            y0 = ops.gemm_nn(g0, w)
            ops.LN_BWD_FUSE = keep
            # one launch
            dx1, g1, ws1 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan")), torch.zeros(nws, device=DEV)
            dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
            pend = dict(dy=dy, x=x, gamma=gamma, beta=beta, stats=stats, add=add, dx=dx1, g=g1, ws=ws1, spec=(0.5, p, 0x1234, gmask), rows=rows, d=d,
                        dg=dg, db=db, table=False, done=False, rowmask=lmask)
            ops._PENDING_LN[g1.data_ptr()] = pend
            n0 = ops.LN_BWD_FUSED_LAUNCHES
            y1 = ops.gemm_nn(g1, w)
            assert ops.LN_BWD_FUSED_LAUNCHES == n0 + 1 and pend["done"] and not ops._PENDING_LN
    finally:
        ops._seed_dev = old
    dg0, db0 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    hip.call("oe_layernorm_param_reduce", ws0, rows, d, dg0, db0)
    torch.cuda.synchronize()
    # (the same formulas in another kernel: hipcc contracts a*b+c differently here and there - one unit in the last place)
    torch.testing.assert_close(dx1, dx0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(g1, g0, rtol=2e-6, atol=1e-6)
    assert torch.equal(g1 == 0, g0 == 0)                              # the same dropout mask / dead rows
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dg, dg0, rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(db, db0, rtol=2e-5, atol=2e-4)


def test_a_parked_layernorm_backward_is_resolved_on_every_other_path():
    """gemm_nn on a kernel that cannot take the prologue (k = 512 here; the tile form; oe_gemm_f32) launches the LayerNorm backward on
    its own first; so does a miss in _out_drop_grad; resolve_pending_ln() sweeps what nobody picked up."""
    torch.manual_seed(92)
    rows, d = 4096, 256
    x, dy = torch.randn(rows, d, device=DEV), torch.randn(rows, d, device=DEV)
    gamma, beta = torch.ones(d, device=DEV), torch.zeros(d, device=DEV)
    stats = torch.stack([x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    nws = hip.lib().oe_layernorm_bwd_workspace_floats(rows, d)

    def park():
        dx, g = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
        pend = dict(dy=dy, x=x, gamma=gamma, beta=beta, stats=stats, add=None, dx=dx, g=g, ws=torch.zeros(nws, device=DEV), spec=(1.0, 0.0, 0, None),
                    rows=rows, d=d, dg=torch.zeros(d, device=DEV), db=torch.zeros(d, device=DEV), table=False, done=False)
        ops._PENDING_LN[g.data_ptr()] = pend
        return pend

    want = torch.empty_like(x)
    hip.call("oe_layernorm_bwd_dx", dy, x, gamma, beta, 0, stats, rows, d, None, None, want, torch.zeros(nws, device=DEV))
    with torch.no_grad():
        w_wide = torch.nn.Parameter(torch.randn(256, 320, device=DEV) / 16)            # output 320 wide (not a multiple of 128): oe_gemm_f32
        pend = park()
        n0 = ops.LN_BWD_FUSED_LAUNCHES
        y = ops.gemm_nn(pend["g"], w_wide)
        assert pend["done"] and ops.LN_BWD_FUSED_LAUNCHES == n0 and not ops._PENDING_LN
        torch.cuda.synchronize()
        assert torch.equal(pend["dx"], want) and torch.equal(pend["g"], want) and bool(torch.isfinite(y).all())      # (the same kernel: exact)
        pend = park()
        assert ops.resolve_pending_ln() == 1 and pend["done"] and not ops._PENDING_LN
        torch.cuda.synchronize()
        assert torch.equal(pend["dx"], want)


@pytest.mark.parametrize("rows,n,with_mask", [(7936, 768, False), (4099, 512, True)])
def test_ln_forward_prologue_equals_the_two_launches(rows, n, with_mask):
    """oe_layernorm_fwd followed by oe_rowgemm6 against ONE oe_rowgemm6 with the lnf arguments (the pre-norm in front of the fused
    q / k / v projection and of pointwise_conv1): y, the statistics and the product to one unit in the last place."""
    torch.manual_seed(95)
    d = 256
    x = torch.randn(rows, d, device=DEV) * 1.7 + 0.3
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    mask = (torch.rand(rows, device=DEV) > 0.1).to(torch.uint8) if with_mask else None
    w = torch.nn.Parameter(torch.randn(n, d, device=DEV) / 16)
    b = torch.randn(n, device=DEV) * 0.1
    y0, st0 = torch.empty_like(x), torch.empty(rows, 2, device=DEV)
    hip.call("oe_layernorm_fwd", x, gamma, beta, 1e-5, rows, d, mask, 0, y0, st0)
    with torch.no_grad():
        out0 = ops.gemm_nt(y0, w, b)
        y1, st1 = torch.full_like(x, float("nan")), torch.full((rows, 2), float("nan"), device=DEV)
        pend = dict(x=x, gamma=gamma, beta=beta, eps=1e-5, rows=rows, d=d, rowmask=mask, y=y1, stats=st1, done=False)
        ops._PENDING_LNF[y1.data_ptr()] = pend
        n0 = ops.LN_FWD_FUSED_LAUNCHES
        out1 = ops.gemm_nt(y1, w, b)
        assert ops.LN_FWD_FUSED_LAUNCHES == n0 + 1 and pend["done"] and not ops._PENDING_LNF
        # any other op of the module that receives a parked tensor launches the LayerNorm on its own first
        y2, st2 = torch.full_like(x, float("nan")), torch.full((rows, 2), float("nan"), device=DEV)
        ops._PENDING_LNF[y2.data_ptr()] = dict(x=x, gamma=gamma, beta=beta, eps=1e-5, rows=rows, d=d, rowmask=mask, y=y2, stats=st2, done=False)
        z = ops.add(y2, y2) if hasattr(ops, "add") else None
    torch.cuda.synchronize()
    torch.testing.assert_close(y1, y0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(st1, st0, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(out1, out0, rtol=1e-5, atol=1e-5)
    assert not ops._PENDING_LNF and torch.equal(y2, y0) and torch.equal(st2, st0)
    if z is not None:
        assert torch.equal(z, y0 + y0)


@pytest.mark.parametrize("rows,act", [(7936, 2), (4107, 1), (5000, 0)])
def test_ln_backward_epilogue_equals_the_two_launches(rows, act):
    """oe_rowgemm6 with the lne arguments (the conv module's norm + activation between the depthwise convolution and pointwise_conv2,
    convolution.py:107-111): dz = g W2 followed by oe_layernorm_bwd_dx against ONE launch whose product leaves through that backward."""
    torch.manual_seed(97)
    d = 256
    gq = torch.randn(rows, d, device=DEV)
    w = torch.nn.Parameter(torch.randn(d, d, device=DEV) / 16)
    yc = torch.randn(rows, d, device=DEV) * 1.5 + 0.4
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    stats = torch.stack([yc.mean(1), 1.0 / torch.sqrt(yc.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    nws = hip.lib().oe_layernorm_bwd_workspace_floats(rows, d)
    with torch.no_grad():
        dz = ops.gemm_nn(gq, w)
        dx0, ws0 = torch.empty_like(yc), torch.zeros(nws, device=DEV)
        hip.call("oe_layernorm_bwd_dx", dz, yc, gamma, beta, act, stats, rows, d, None, None, dx0, ws0)
        dx1, ws1 = torch.full_like(yc, float("nan")), torch.zeros(nws, device=DEV)
        epi = dict(x=yc, stats=stats, gamma=gamma, beta=beta, act=act, dx=dx1, ws=ws1, done=False)
        ops.gemm_nn(gq, w, ln_epi=epi)
        assert epi["done"]
        epi2 = dict(epi, dx=torch.empty_like(yc), done=False)
        ops.gemm_nn(gq, torch.nn.Parameter(torch.randn(d, 320, device=DEV)), ln_epi=epi2)      # not 256 <- 256: plain product, nothing fused
        assert not epi2["done"]
    r = [torch.zeros(d, device=DEV) for _ in range(4)]
    hip.call("oe_layernorm_param_reduce", ws0, rows, d, r[0], r[1])
    hip.call("oe_layernorm_param_reduce", ws1, rows, d, r[2], r[3])
    torch.cuda.synchronize()
    scale = float(dx0.abs().max())
    assert float((dx1 - dx0).abs().max()) <= 3e-6 * scale + 1e-6
    torch.testing.assert_close(r[2], r[0], rtol=3e-5, atol=3e-4)
    torch.testing.assert_close(r[3], r[1], rtol=3e-5, atol=3e-4)


def test_ln_backward_epilogue_is_stable_over_many_launches():
    """Regression: with (x - mean) rstd compiled to packed-fp32 instructions (v_pk_add_f32 ... v_pk_mul_f32 op_sel:[0,1]) the low halves came
    out zero for lanes 48-63 while the SIMD's other wave issued MFMAs: rows 6 / 7 of ~10 % of the blocks wrong, other blocks at every launch.
    The kernel writes it as single-float instructions and starts the epilogue behind a block barrier (profiles/r04_experiments.md, Findings;
    reproducer: -DOE_LNE_REPRO, tools/probes/lne_variants.sh)."""
    torch.manual_seed(98)
    rows, d = 7936, 256
    gq = torch.randn(rows, d, device=DEV)
    w = torch.nn.Parameter(torch.randn(d, d, device=DEV) / 16)
    yc = torch.randn(rows, d, device=DEV) * 1.5 + 0.4
    gamma, beta = torch.randn(d, device=DEV) * 0.2 + 1.0, torch.randn(d, device=DEV) * 0.1
    stats = torch.stack([yc.mean(1), 1.0 / torch.sqrt(yc.var(1, unbiased=False) + 1e-5)], 1).contiguous()
    nws = hip.lib().oe_layernorm_bwd_workspace_floats(rows, d)
    with torch.no_grad():
        dz = ops.gemm_nn(gq, w)
        dx0 = torch.empty_like(yc)
        hip.call("oe_layernorm_bwd_dx", dz, yc, gamma, beta, 2, stats, rows, d, None, None, dx0, torch.zeros(nws, device=DEV))
        worst = 0.0
        for _ in range(60):
            dx1 = torch.full_like(yc, float("nan"))
            epi = dict(x=yc, stats=stats, gamma=gamma, beta=beta, act=2, dx=dx1, ws=torch.zeros(nws, device=DEV), done=False)
            ops.gemm_nn(gq, w, ln_epi=epi)
            assert epi["done"]
            worst = max(worst, float((dx1 - dx0).abs().max()))
    assert worst <= 3e-6 * float(dx0.abs().max()) + 1e-6, worst
