"""GPU: the launch-saving fusions of the training step against the unfused arithmetic they replace -
the joint loss as one launch (ops.combine_losses: asr_model.py:150-157 + :196-198), boolean masks reinterpreted as bytes,
the back-to-back LayerNorm pair at an encoder layer boundary (ops.layer_norm_pair: encoder_layer.py:109-110 followed by the
next layer's :79-80), and the depthwise convolution that normalises its own rows (ops.conv_module: convolution.py:100-111)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from openeat_amd import ops  # noqa: E402

DEV = "cuda"


def test_combine_losses_equals_the_scalar_ops_bitwise():
    g = torch.Generator().manual_seed(3)
    for wc, r in ((0.3, 0.3), (0.5, 0.0), (0.0, 0.7), (1.0, 0.25)):
        vals = (torch.rand(3, generator=g) * 100 + 1).to(DEV)
        la, lr, lc = [vals[i].clone().requires_grad_(True) for i in range(3)]
        out = ops.combine_losses(la, lr, lc, wc, r)
        la2, lr2, lc2 = [t.detach().clone().requires_grad_(True) for t in (la, lr, lc)]
        att = la2 * (1 - r) + lr2 * r
        ref = wc * lc2 + (1 - wc) * att
        assert out.shape == ref.shape == ()
        assert out.item() == ref.item()
        (out * 1.7).backward()
        (ref * 1.7).backward()
        for a, b in ((la, la2), (lr, lr2), (lc, lc2)):
            assert a.grad.item() == b.grad.item()
    # absent terms: no right-to-left decoder, no CTC branch
    la = torch.tensor(12.5, device=DEV, requires_grad=True)
    lc = torch.tensor(40.25, device=DEV, requires_grad=True)
    out = ops.combine_losses(la, None, lc, 0.3, 0.0)
    ref = 0.3 * lc.detach() + (1 - 0.3) * la.detach()
    assert out.item() == ref.item()
    out.backward()
    assert la.grad.item() == pytest.approx(0.7, rel=1e-6) and lc.grad.item() == pytest.approx(0.3, rel=1e-6)
    la.grad = None
    out = ops.combine_losses(la, torch.tensor(3.0, device=DEV), None, 0.0, 0.25)
    assert out.item() == (la.detach() * 0.75 + 3.0 * 0.25).item()


def test_mask_bytes_is_a_view_of_a_bool_mask():
    m = torch.rand(4, 1, 37, device=DEV) > 0.3
    b = ops.mask_bytes(m)
    assert b.dtype == torch.uint8 and b.data_ptr() == m.data_ptr() and torch.equal(b.bool(), m)
    i = (torch.rand(4, 1, 37, device=DEV) > 0.3).to(torch.int32)
    assert torch.equal(ops.mask_bytes(i), i.to(torch.uint8))


def _pair_inputs(rows, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3).to(DEV)
    ps = [(torch.randn(d, generator=g) * 0.5 + (1.0 if i % 2 == 0 else 0.0)).to(DEV) for i in range(4)]
    du, dy = torch.randn(rows, d, generator=g).to(DEV), torch.randn(rows, d, generator=g).to(DEV)
    return x, ps, du, dy


@pytest.mark.parametrize("rows,d", [(100, 256), (7936, 256), (37, 512), (19, 1024), (5, 144), (9, 2048)])
def test_layer_norm_pair_equals_the_two_norms(rows, d):
    x, ps, du, dy = _pair_inputs(rows, d, rows + d)
    leaves = lambda: [t.detach().clone().requires_grad_(True) for t in [x] + ps]
    # fused
    xf, g1, b1, g2, b2 = leaves()
    u, y = ops.layer_norm_pair(xf, g1, b1, 1e-12, g2, b2, 1e-5)
    ((u * du).sum() + (y * dy).sum()).backward()
    fused = [u.detach(), y.detach(), xf.grad, g1.grad, b1.grad, g2.grad, b2.grad]
    # the two ops it replaces
    xr, h1, c1, h2, c2 = leaves()
    ur = ops.layer_norm(xr, h1, c1, 1e-12)
    r, yr = ops.pre_norm(ur, h2, c2, 1e-5)
    ((r * du).sum() + (yr * dy).sum()).backward()
    ref = [ur.detach(), yr.detach(), xr.grad, h1.grad, c1.grad, h2.grad, c2.grad]
    assert torch.equal(fused[0], ref[0]) and torch.equal(fused[1], ref[1])          # same arithmetic, same order
    for a, b, name in zip(fused[2:], ref[2:], ("dx", "dgamma1", "dbeta1", "dgamma2", "dbeta2")):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, name
    # and against float64 autograd of the definition
    x64 = x.double().requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in ps]
    u64 = torch.nn.functional.layer_norm(x64, (d,), p64[0], p64[1], 1e-12)
    y64 = torch.nn.functional.layer_norm(u64, (d,), p64[2], p64[3], 1e-5)
    ((u64 * du.double()).sum() + (y64 * dy.double()).sum()).backward()
    assert float((fused[1].double() - y64).abs().max()) <= 2e-5 * float(y64.abs().max())
    assert float((fused[2].double() - x64.grad).abs().max()) <= 5e-5 * float(x64.grad.abs().max())
    for a, b in zip(fused[3:], [t.grad for t in p64]):
        assert float((a.double() - b).abs().max()) <= 5e-5 * float(b.abs().max())


def test_layer_norm_pair_second_output_only_and_the_dropped_gradient_copy():
    rows, d = 300, 256
    x, ps, _, dy = _pair_inputs(rows, d, 7)
    x0 = x.clone().requires_grad_(True)
    xt = x0 * 1.0
    spec = (0.5, 0.1, 1234, None)
    xt._oe_outdrop = spec
    seen = []
    xt.register_hook(lambda g: seen.append(g))
    pl = [t.clone().requires_grad_(True) for t in ps]
    ops.predrop_clear()
    y = ops.layer_norm_pair(xt, pl[0], pl[1], 1e-12, pl[2], pl[3], 1e-5, want_first=False, sole_consumer=True)
    (y * dy).sum().backward()
    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in ps]
    yr = ops.layer_norm(ops.layer_norm(xr, pr[0], pr[1], 1e-12), pr[2], pr[3], 1e-5)
    (yr * dy).sum().backward()
    assert torch.equal(y.detach(), yr.detach())
    assert float((x0.grad - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())
    # the copy the previous block's backward picks up: alpha * dropout_mask(seed) * dx of exactly this dx
    g = seen[0]
    got = ops._out_drop_grad(g, *spec[:3])
    want = ops.dropout_scale(g, *spec[:3])
    assert got.data_ptr() != want.data_ptr() and torch.equal(got, want)


@pytest.mark.parametrize("B,T,d,K,causal,act", [(3, 50, 256, 15, False, 2), (2, 33, 128, 7, True, 1), (1, 16, 512, 31, False, 2), (2, 5, 64, 3, False, 0)])
def test_dwconv_with_its_layernorm_equals_the_two_launches(B, T, d, K, causal, act):
    from openeat_amd import hip
    g = torch.Generator().manual_seed(B * 1000 + T)
    a = torch.randn(B * T, 2 * d, generator=g).to(DEV)
    wd = (torch.randn(d, K, generator=g) * 0.3).to(DEV)
    bd = torch.randn(d, generator=g).to(DEV)
    gam = (torch.randn(d, generator=g) * 0.3 + 1).to(DEV)
    bet = (torch.randn(d, generator=g) * 0.3).to(DEV)
    gpad = torch.randn(d, generator=g).to(DEV) if causal else None
    y0, z0, st0 = torch.empty(B * T, d, device=DEV), torch.empty(B * T, d, device=DEV), torch.empty(B * T, 2, device=DEV)
    hip.call("oe_dwconv_glu_fwd", a, wd, bd, gpad, B, T, d, K, int(causal), y0)
    hip.call("oe_layernorm_fwd", y0, gam, bet, 1e-5, B * T, d, None, act, z0, st0)
    y1, z1, st1 = torch.full_like(y0, float("nan")), torch.full_like(z0, float("nan")), torch.full_like(st0, float("nan"))
    hip.call("oe_dwconv_glu_ln_fwd", a, wd, bd, gpad, B, T, d, K, int(causal), y1, gam, bet, 1e-5, act, z1, st1)
    assert torch.equal(y0, y1)
    assert float((st0 - st1).abs().max()) <= 1e-6 * float(st0.abs().max())
    assert float((z0 - z1).abs().max()) <= 1e-6 * float(z0.abs().max())


def test_conv1_activation_as_planes_only_equals_the_fp32_copy_path():
    """Conv2dSubsampling4 at a size where the conv1 output is kept as bf16 planes alone (no fp32 copy; the ReLU mask of the
    input gradient read from plane 0): same output, same gradients as with the fp32 copy (subsampling.py:110-116)."""
    from openeat_amd import hip, planes
    old = hip.GEMM_PRECISION
    hip.GEMM_PRECISION = 6
    try:
        B, T, Fd, C, d = 8, 998, 80, 256, 256
        g = torch.Generator().manual_seed(11)
        x = torch.randn(B, T, Fd, generator=g).to(DEV)
        ps = [torch.randn(C, 1, 3, 3, generator=g) * 0.3, torch.randn(C, generator=g) * 0.1, torch.randn(d, C * 19, generator=g) * 0.02,
              torch.randn(d, generator=g) * 0.1, torch.randn(C, C, 3, 3, generator=g) * 0.03, torch.randn(C, generator=g) * 0.1]
        dout = torch.randn(B, 248, d, generator=g).to(DEV)
        res = {}
        for mode in (True, False):
            ops.CONV1_PLANES_ONLY = mode
            planes.clear()
            leaves = [p.to(DEV).requires_grad_(True) for p in ps]
            w1, b1, wl, bl, w2, b2 = leaves
            out = ops.ConvSubsamplingFn.apply(x, w1, b1, wl, bl, None, 16.0, ((3, 2),), w2, b2)
            assert bool(out.grad_fn.planes_only) == mode                      # the size qualifies: the switch alone decides
            (out * dout).sum().backward()
            res[mode] = [out.detach()] + [p.grad.clone() for p in leaves]
        assert torch.equal(res[True][0], res[False][0])
        for a, b, name in zip(res[True][1:], res[False][1:], ("dw1", "db1", "dwl", "dbl", "dw2", "db2")):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()), name       # (atomic accumulation orders differ run to run)
    finally:
        ops.CONV1_PLANES_ONLY = True
        hip.GEMM_PRECISION = old
        planes.clear()
