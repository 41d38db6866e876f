"""GPU: attention, conv, loss and optimiser kernels through the C ABI against
fp32 CPU restatements (oracle / plain torch) on the same seeded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from openeat_amd import hip  # noqa: E402
from oracle import asr as O  # noqa: E402

DEV = "cuda"
TOL = dict(rtol=2e-4, atol=2e-5)


def cu(t):
    return t.to(DEV).contiguous()


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------- attention -----
def ref_attention(q, k, v, mask, keybias, scale):
    """q (B,T1,H,D) ... -> out (B,T1,H,D); mask (B,1|T1,T2) bool; keybias (B,H,T2)."""
    s = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    if keybias is not None:
        s = s + keybias[:, :, None, :]
    if mask is not None:
        m = mask.unsqueeze(1).eq(0)
        s = s.masked_fill(m, -float("inf"))
        a = torch.softmax(s, dim=-1).masked_fill(m, 0.0)
    else:
        a = torch.softmax(s, dim=-1)
    return torch.einsum("bhij,bjhd->bihd", a, v)


@pytest.mark.parametrize("B,H,T1,T2,D,mask_kind,bias", [
    (2, 4, 50, 50, 64, "key", True),
    (3, 4, 13, 13, 8, "key", True),
    (2, 2, 31, 70, 32, "key", False),
    (3, 4, 9, 9, 16, "full", False),
    (1, 4, 130, 130, 64, "none", False),
    (2, 4, 40, 40, 36, "key", True),
    # the LDS-plane kernels (attention_bf16.hip: resident axis > 64 rows, precision 1 / 3): several 64-row chunks, ragged last
    # chunk and tile, waves without rows, a short streamed axis against a long resident one and vice versa, D < 64 padded
    (2, 4, 248, 248, 64, "key", True),
    (1, 2, 398, 398, 64, "key", True),
    (2, 2, 200, 31, 64, "key", False),
    (2, 2, 31, 200, 64, "key", True),
    (1, 2, 97, 97, 8, "full", False),
    (2, 2, 129, 70, 32, "key", True),
    (1, 2, 150, 150, 40, "none", False),
])
@pytest.mark.parametrize("prec", [0, 6, 3, 1])
def test_attention_fwd_bwd(B, H, T1, T2, D, mask_kind, bias, prec):
    """prec 0: exact fp32 matrix-core products; 6: six bf16 terms on three exact pieces - held to mode 0's tolerances (all
    three kernels on planes where the resident axis is long enough, the fp32 kernels elsewhere); 3: three-term bf16 split
    (2^-17 per product: an absolute floor that grows with the values summed); 1: plain bf16 products (~2^-8 per product)."""
    torch.manual_seed(11)
    q = torch.randn(B, T1, H, D, requires_grad=True)
    k = torch.randn(B, T2, H, D, requires_grad=True)
    v = torch.randn(B, T2, H, D, requires_grad=True)
    kb = (torch.randn(B, H, T2) * 0.5).requires_grad_() if bias else None
    scale = 1.0 / math.sqrt(D)
    mask = None
    if mask_kind == "key":
        lens = torch.randint(1, T2 + 1, (B,))
        lens[0] = T2
        mask = (~O.pad_mask(lens, T2)).unsqueeze(1)
    elif mask_kind == "full":
        lens = torch.randint(1, T2 + 1, (B,))
        mask = (~O.pad_mask(lens, T2)).unsqueeze(1) & O.causal_mask(T1).unsqueeze(0)
    out_ref = ref_attention(q, k, v, mask, kb, scale)
    w = torch.randn_like(out_ref)
    (out_ref * w).sum().backward()

    # device: q/k/v as slices of one fused (B,T,3*H*D) buffer when T1 == T2, to exercise strides
    d = H * D
    if T1 == T2:
        qkv = torch.cat([q.detach().reshape(B, T1, d), k.detach().reshape(B, T1, d), v.detach().reshape(B, T1, d)], -1).to(DEV)
        qd, kd, vd = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        qs = ks = vs = (T1 * 3 * d, 3 * d)
        dqkv = torch.full_like(qkv, float("nan"))
        dqd, dkd, dvd = dqkv[:, :, :d], dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:]
    else:
        qd, kd, vd = cu(q.detach()), cu(k.detach()), cu(v.detach())
        qs, ks, vs = (T1 * d, d), (T2 * d, d), (T2 * d, d)
        dqd, dkd, dvd = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    out = torch.full((B, T1, H, D), float("nan"), device=DEV)
    lse = torch.empty(B, H, T1, device=DEV)
    md = None if mask is None else cu(mask.to(torch.uint8))
    mstr = (0, 0) if mask is None else (mask.shape[1] * T2, 0 if mask.shape[1] == 1 else T2)
    kbd = None if kb is None else cu(kb.detach())
    a = hip.attn_args(qd, kd, vd, out, lse, B, H, T1, T2, D, scale, q_strides=qs, k_strides=ks, v_strides=vs,
                      o_strides=(T1 * d, d), mask=md, mask_strides=mstr, keybias=kbd, precision=prec)
    hip.attention_fwd(a)
    sync()
    # precision 3: ~2^-17 per product on sums of T2 terms of magnitude |p||v| - the floor grows with the values summed, not
    # with the (cancelled) result: 4e-5 covers T2 = 398 at |v| ~ 4
    tol_fwd = TOL if prec in (0, 6) else dict(rtol=2e-4, atol=4e-5) if prec == 3 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.cpu(), out_ref.detach(), **tol_fwd)

    wd = cu(w)
    dkb = torch.empty(B, H, T2, device=DEV) if bias else None
    delta = torch.empty(B, H, T1, device=DEV)
    a2 = hip.attn_args(qd, kd, vd, out, lse, B, H, T1, T2, D, scale, q_strides=qs, k_strides=ks, v_strides=vs,
                       o_strides=(T1 * d, d), mask=md, mask_strides=mstr, keybias=kbd, d_out=wd, dq=dqd, dk=dkd, dv=dvd,
                       dkeybias=dkb, delta=delta, precision=prec)
    hip.attention_bwd(a2)
    sync()
    # (precision 3: sums over up to 398 queries / keys of ~2^-17-accurate products: absolute floor 1e-4, see tol_fwd)
    g = dict(rtol=5e-4, atol=5e-5) if prec in (0, 6) else dict(rtol=5e-4, atol=1e-4) if prec == 3 else dict(rtol=5e-2, atol=8e-2)
    torch.testing.assert_close(dqd.cpu().reshape(B, T1, H, D), q.grad, **g)
    torch.testing.assert_close(dkd.cpu().reshape(B, T2, H, D), k.grad, **g)
    torch.testing.assert_close(dvd.cpu().reshape(B, T2, H, D), v.grad, **g)
    if bias:
        torch.testing.assert_close(dkb.cpu(), kb.grad, **g)


def test_attention_fully_masked_rows_give_zeros():
    torch.manual_seed(12)
    B, H, T, D = 2, 2, 10, 16
    q, k, v = (torch.randn(B, T, H, D) for _ in range(3))
    mask = torch.ones(B, 1, T, dtype=torch.uint8)
    mask[1] = 0                               # utterance 1: every key masked (length 0)
    out = torch.full((B, T, H, D), float("nan"), device=DEV)
    lse = torch.empty(B, H, T, device=DEV)
    d = H * D
    qd, kd, vd, md = cu(q), cu(k), cu(v), cu(mask)
    a = hip.attn_args(qd, kd, vd, out, lse, B, H, T, T, D, 0.25, q_strides=(T * d, d), k_strides=(T * d, d),
                      v_strides=(T * d, d), o_strides=(T * d, d), mask=md, mask_strides=(T, 0))
    hip.attention_fwd(a)
    sync()
    assert torch.all(out[1] == 0) and torch.isfinite(out[0]).all()


@pytest.mark.parametrize("T1,T2,prec", [(32, 32, 0), (70, 40, 3), (33, 96, 3), (20, 13, 0), (45, 45, 3),
                                        (136, 136, 3), (100, 75, 3), (40, 160, 3), (130, 96, 1),
                                        (136, 136, 6), (100, 75, 6), (40, 160, 6), (200, 300, 6)])
def test_attention_dropout_consistency(T1, T2, prec):
    """Dropout in the attention weights.  The kernel's own mask is recovered from a V = identity probe; then
    forward == (mask/keep * softmax) V and the three backward kernels agree with autograd through that mask.
    The mask is a hash of (row, key quad) (attn_common.h): forward / dQ draw it per lane, dK/dV shares it across the four
    lanes of a key quad by DPP broadcasts; both kernel families (attention.hip for short resident axes, attention_bf16.hip
    for long ones - some cases here mix them) must describe the same mask in all three kernels."""
    torch.manual_seed(13)
    B, H, D = 2, 2, 32
    p_drop = 0.25
    q = torch.randn(B, T1, H, D, requires_grad=True)
    k = torch.randn(B, T2, H, D, requires_grad=True)
    d = H * D
    qd, kd = cu(q.detach()), cu(k.detach())
    lse = torch.empty(B, H, T1, device=DEV)
    st_q, st_k = (T1 * d, d), (T2 * d, d)

    def args(v, out, **kw):
        return hip.attn_args(qd, kd, v, out, lse, B, H, T1, T2, D, 0.2, q_strides=st_q, k_strides=st_k, v_strides=st_k,
                             o_strides=st_q, drop_p=p_drop, seed=77, precision=prec, **kw)
    # probe: V[b, j, h, :] = e_j (needs D >= T2 per head: use several probes of D columns each)
    w_drop = torch.zeros(B, H, T1, T2)
    for c0 in range(0, T2, D):
        vprobe = torch.zeros(B, T2, H, D)
        for jj in range(c0, min(T2, c0 + D)):
            vprobe[:, jj, :, jj - c0] = 1.0
        out = torch.empty(B, T1, H, D, device=DEV)
        hip.attention_fwd(args(cu(vprobe), out))
        sync()
        w_drop[:, :, :, c0:min(T2, c0 + D)] = out.cpu().permute(0, 2, 1, 3)[..., : min(T2, c0 + D) - c0]
    attn = torch.softmax(torch.einsum("bihd,bjhd->bhij", q.detach(), k.detach()) * 0.2, -1)
    keep = w_drop != 0
    assert abs(keep.float().mean().item() - (1 - p_drop)) < 0.04
    tol = dict(rtol=2e-4, atol=2e-6) if prec != 1 else dict(rtol=3e-2, atol=3e-3)
    torch.testing.assert_close(w_drop[keep], (attn / (1 - p_drop))[keep], **tol)
    mask = keep.float() / (1 - p_drop)
    # full forward + backward against autograd with that mask
    v = torch.randn(B, T2, H, D, requires_grad=True)
    ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(torch.einsum("bihd,bjhd->bhij", q, k) * 0.2, -1) * mask, v)
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    vd, out = cu(v.detach()), torch.empty(B, T1, H, D, device=DEV)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(kd)
    delta = torch.empty(B, H, T1, device=DEV)
    a = args(vd, out, d_out=cu(w), dq=dq, dk=dk, dv=dv, delta=delta)
    hip.attention_fwd(a)
    hip.attention_bwd(a)
    sync()
    g = dict(rtol=1e-3, atol=1e-4) if prec != 1 else dict(rtol=5e-2, atol=8e-2)
    torch.testing.assert_close(out.cpu(), ref.detach(), **g)
    torch.testing.assert_close(dv.cpu(), v.grad, **g)
    torch.testing.assert_close(dq.cpu(), q.grad, **g)
    torch.testing.assert_close(dk.cpu(), k.grad, **g)


def test_relpos_prepare_and_backward():
    torch.manual_seed(14)
    B, T, H, D = 3, 17, 4, 8
    d = H * D
    qkv = torch.randn(B, T, 3 * d)
    k = qkv[:, :, d:2 * d].reshape(B, T, H, D).clone().requires_grad_()
    p = torch.randn(T, H, D, requires_grad=True)
    u = torch.randn(H, D, requires_grad=True)
    v = torch.randn(H, D, requires_grad=True)
    scale = 1 / math.sqrt(D)
    kp_ref = k + p[None]
    kb_ref = scale * (torch.einsum("hd,bthd->bht", u, k) + torch.einsum("hd,thd->ht", v, p)[None])
    gk, gb = torch.randn_like(kp_ref), torch.randn_like(kb_ref)
    ((kp_ref * gk).sum() + (kb_ref * gb).sum()).backward()
    qkvd, pd, ud, vd = cu(qkv), cu(p.detach().reshape(T, d)), cu(u.detach()), cu(v.detach())
    kd = qkvd[:, :, d:2 * d]
    kp = torch.empty(B, T, H, D, device=DEV)
    kb = torch.empty(B, H, T, device=DEV)
    hip.call("oe_relpos_prepare", kd, T * 3 * d, 3 * d, pd, d, ud, vd, B, T, H, D, scale, kp, kb)
    sync()
    torch.testing.assert_close(kp.cpu(), kp_ref.detach(), **TOL)
    torch.testing.assert_close(kb.cpu(), kb_ref.detach(), **TOL)
    dqkv = torch.zeros(B, T, 3 * d, device=DEV)
    dp = torch.empty(T, d, device=DEV)
    du, dv = torch.zeros(H, D, device=DEV), torch.zeros(H, D, device=DEV)
    gkd, gbd = cu(gk), cu(gb)
    hip.call("oe_relpos_backward", gkd, gbd, kd, T * 3 * d, 3 * d, pd, d, ud, vd, B, T, H, D, scale,
             dqkv[:, :, d:2 * d], dp, d, du, dv)
    sync()
    torch.testing.assert_close(dqkv[:, :, d:2 * d].cpu().reshape(B, T, H, D), k.grad, **TOL)
    torch.testing.assert_close(dp.cpu().view(T, H, D), p.grad, rtol=2e-4, atol=1e-4)
    torch.testing.assert_close(du.cpu(), u.grad, rtol=2e-4, atol=1e-4)
    torch.testing.assert_close(dv.cpu(), v.grad, rtol=2e-4, atol=1e-4)


# ------------------------------------------------------------ elementwise ----
def test_glu_dropout_embed_swap_axpby_cmvn():
    torch.manual_seed(15)
    rows, d = 77, 32
    a = torch.randn(rows, 2 * d, requires_grad=True)
    y_ref = F.glu(a, dim=1)
    gy = torch.randn(rows, d)
    y_ref.backward(gy)
    ad, gyd = cu(a.detach()), cu(gy)
    y, da = torch.empty(rows, d, device=DEV), torch.empty(rows, 2 * d, device=DEV)
    hip.call("oe_glu_fwd", ad, rows, d, y)
    hip.call("oe_glu_bwd", ad, gyd, rows, d, da)
    sync()
    torch.testing.assert_close(y.cpu(), y_ref.detach(), **TOL)
    torch.testing.assert_close(da.cpu(), a.grad, **TOL)
    # dropout_scale reproduces the GEMM epilogue mask (same seed / index)
    M, N, K = 64, 128, 16
    x, w = torch.randn(M, K), torch.randn(N, K)
    xd, wd = cu(x), cu(w)
    g1 = torch.empty(M, N, device=DEV)
    hip.gemm(xd, wd, g1, M, N, K, lda=K, ldb=K, ldc=N, drop_p=0.3, seed=5)
    g0 = torch.empty(M, N, device=DEV)
    hip.gemm(xd, wd, g0, M, N, K, lda=K, ldb=K, ldc=N)
    g2 = torch.empty(M, N, device=DEV)
    rm = cu((torch.rand(M) > 0.5).to(torch.uint8))
    hip.call("oe_dropout_scale", g0, M * N, N, 2.0, 0.3, 5, None, rm, g2)
    sync()
    torch.testing.assert_close(g2, 2.0 * g1 * rm[:, None].float(), rtol=1e-6, atol=1e-6)
    # embedding
    V, L, B = 30, 7, 3
    table = torch.randn(V, d, requires_grad=True)
    tok = torch.randint(0, V, (B, L))
    pe = O.sinusoid_table(d)[0, :L]
    ref = F.embedding(tok, table) * math.sqrt(d) + pe
    go = torch.randn(B, L, d)
    ref.backward(go)
    td, tokd, ped, god = cu(table.detach()), cu(tok), cu(pe), cu(go)
    out = torch.empty(B, L, d, device=DEV)
    dt = torch.zeros(V, d, device=DEV)
    hip.call("oe_embed_fwd", tokd, td, ped, B * L, L, d, V, math.sqrt(d), out)
    hip.call("oe_embed_bwd", tokd, god, B * L, d, V, math.sqrt(d), dt)
    sync()
    torch.testing.assert_close(out.cpu(), ref.detach(), **TOL)
    torch.testing.assert_close(dt.cpu(), table.grad, rtol=2e-4, atol=1e-4)
    # swap / axpby / cmvn
    t = torch.randn(5, 6, 9)
    td2 = cu(t)
    o = torch.empty(5, 9, 6, device=DEV)
    hip.call("oe_swap_last2", td2, 5, 6, 9, o, 0)
    z = torch.empty(5 * 6 * 9, device=DEV)
    hip.call("oe_axpby", td2, td2, 5 * 6 * 9, 2.0, 0.5, None, z)
    mean, istd = torch.randn(9), torch.rand(9) + 0.5
    md, isd = cu(mean), cu(istd)
    c = torch.empty(5, 6, 9, device=DEV)
    hip.call("oe_global_cmvn", td2, md, isd, 5 * 6 * 9, 9, c)
    sync()
    assert torch.equal(o.cpu(), t.transpose(1, 2).contiguous())
    torch.testing.assert_close(z.cpu(), 2.5 * t.flatten())
    torch.testing.assert_close(c.cpu(), (t - mean) * istd, **TOL)


# -------------------------------------------------------------------- conv ----
def test_conv1_fwd_wgrad_and_conv2_dgrad():
    torch.manual_seed(16)
    B, T, Fd, C = 3, 37, 20, 32
    x = torch.randn(B, T, Fd)
    w1 = (torch.randn(C, 1, 3, 3) * 0.5).requires_grad_()
    b1 = torch.randn(C, requires_grad=True)
    w2 = (torch.randn(C, C, 3, 3) * 0.1).requires_grad_()
    y1_ref = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2))            # (B,C,T1,F1)
    y2_ref = F.conv2d(y1_ref, w2, None, stride=2)
    g2 = torch.randn_like(y2_ref)
    y2_ref.backward(g2)
    T1, F1 = y1_ref.shape[2:]
    T2, F2 = y2_ref.shape[2:]
    xd, w1d, b1d = cu(x), cu(w1.detach()), cu(b1.detach())
    y1 = torch.empty(B, T1, F1, C, device=DEV)
    hip.call("oe_conv1_fwd", xd, w1d, b1d, B, T, Fd, C, y1)
    sync()
    torch.testing.assert_close(y1.cpu().permute(0, 3, 1, 2), y1_ref.detach(), **TOL)
    # conv2 input gradient: dcol = dy @ W2g (k-major B), then gather + relu mask
    M = B * T2 * F2
    w2g = cu(w2.detach().permute(0, 2, 3, 1).reshape(C, 9 * C))
    dy2 = cu(g2.permute(0, 2, 3, 1).reshape(M, C))
    dcol = torch.empty(M, 9 * C, device=DEV)
    hip.gemm(dy2, w2g, dcol, M, 9 * C, C, lda=C, ldb=9 * C, ldc=9 * C, b_kmajor=True)
    dy1 = torch.empty(B, T1, F1, C, device=DEV)
    hip.call("oe_col2im_relu", dcol, y1, B, T1, F1, C, dy1)
    dw1 = torch.zeros(C, 9, device=DEV)
    db1 = torch.zeros(C, device=DEV)
    hip.call("oe_conv1_wgrad", xd, dy1, B, T, Fd, C, dw1, db1)
    sync()
    torch.testing.assert_close(dw1.cpu().view(C, 1, 3, 3), w1.grad, rtol=5e-4, atol=5e-4)
    torch.testing.assert_close(db1.cpu(), b1.grad, rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize("prec,tol", [(0, 2e-4), (6, 2e-4), (3, 2e-4)])
@pytest.mark.parametrize("B,Ti,Fi,C", [(3, 21, 11, 32), (2, 22, 12, 64), (2, 498, 39, 64), (5, 7, 5, 32)])
def test_conv2_input_gradient_as_four_parity_gemms(prec, tol, B, Ti, Fi, C):
    """ops._conv_dgrad_k3s2 (gather from the zero-padded dy, row scatter + ReLU mask in the epilogue) against autograd of
    F.conv2d(stride 2): odd and even input sizes (an even size leaves its last row / column without any tap), the
    config-2 geometry 498 x 39, and a size smaller than one GEMM tile."""
    from openeat_amd import ops
    torch.manual_seed(3)
    To, Fo = (Ti - 3) // 2 + 1, (Fi - 3) // 2 + 1
    w = torch.randn(C, C, 3, 3) * 0.1
    x = torch.randn(B, C, Ti, Fi, requires_grad=True)
    dy = torch.randn(B, C, To, Fo)
    F.conv2d(x, w, stride=2).backward(dy)
    yin = torch.randn(B, Ti, Fi, C)                                   # the stage's input activation: only its sign matters
    want = x.grad.permute(0, 2, 3, 1) * (yin > 0)
    old = hip.GEMM_PRECISION
    hip.GEMM_PRECISION = prec
    try:
        got = ops._conv_dgrad_k3s2(cu(dy.permute(0, 2, 3, 1).contiguous().view(-1, C)), cu(w), cu(yin), B, Ti, Fi, To, Fo, C)
        sync()
    finally:
        hip.GEMM_PRECISION = old
    scale = float(want.abs().max())
    torch.testing.assert_close(got.cpu(), want, rtol=tol, atol=tol * scale)
    assert bool((got.cpu()[yin <= 0] == 0).all())                     # masked positions exactly zero


# K = 7 / 15 / 31 take the instantiations without a per-tap test, every other K the bounded ones (K < 7, < 15, < 31)
@pytest.mark.parametrize("causal,K,d,T", [(False, 15, 32, 21), (True, 15, 32, 21), (False, 7, 256, 50), (False, 31, 64, 40), (True, 7, 32, 5),
                                          (False, 3, 32, 21), (True, 5, 64, 19), (False, 9, 64, 40), (True, 13, 32, 33),
                                          (False, 17, 32, 30), (True, 29, 64, 37), (False, 15, 256, 248)])
def test_dwconv_glu_fwd_bwd(causal, K, d, T):
    torch.manual_seed(17)
    B = 3
    a = torch.randn(B, T, 2 * d, requires_grad=True)
    w = (torch.randn(d, 1, K) * 0.3).requires_grad_()
    b = torch.randn(d, requires_grad=True)
    g = F.glu(a.transpose(1, 2), dim=1)
    gp = torch.randn(d, requires_grad=True)          # value on the virtual left frames (causal only)
    if causal:
        g = torch.cat([gp[None, :, None].expand(B, d, K - 1), g], dim=2)
    y_ref = F.conv1d(g, w, b, padding=0 if causal else (K - 1) // 2, groups=d).transpose(1, 2)
    gy = torch.randn(B, T, d)
    y_ref.backward(gy)
    ad, wd, bd, gyd = cu(a.detach()), cu(w.detach()), cu(b.detach()), cu(gy)
    gpd = cu(gp.detach()) if causal else None
    y = torch.empty(B, T, d, device=DEV)
    hip.call("oe_dwconv_glu_fwd", ad, wd, bd, gpd, B, T, d, K, int(causal), y)
    da = torch.empty(B, T, 2 * d, device=DEV)
    dw, db = torch.zeros(d, K, device=DEV), torch.zeros(d, device=DEV)
    dgp = torch.zeros(d, device=DEV) if causal else None
    ws = torch.empty(hip.lib().oe_dwconv_glu_bwd_workspace_floats(B, T, d, K), device=DEV)
    hip.call("oe_dwconv_glu_bwd", ad, gyd, wd, gpd, B, T, d, K, int(causal), da, dw, db, dgp, ws)
    sync()
    if causal:
        torch.testing.assert_close(dgp.cpu(), gp.grad, rtol=5e-4, atol=2e-4)
    torch.testing.assert_close(y.cpu(), y_ref.detach(), **TOL)
    torch.testing.assert_close(da.cpu(), a.grad, rtol=5e-4, atol=5e-5)
    torch.testing.assert_close(dw.cpu().view(d, 1, K), w.grad, rtol=5e-4, atol=2e-4)
    torch.testing.assert_close(db.cpu(), b.grad, rtol=5e-4, atol=2e-4)


# -------------------------------------------------------------------- loss ----
@pytest.mark.parametrize("V,ldv,smooth,nl", [(23, 24, 0.1, False), (3246, 3248, 0.1, False), (50, 52, 0.0, False), (23, 23, 0.1, True)])
def test_label_smoothing_fused(V, ldv, smooth, nl):
    torch.manual_seed(18)
    B, L = 4, 6
    x = (torch.randn(B, L, V) * 2).requires_grad_()
    tgt = torch.randint(0, V, (B, L))
    tgt[1, 4:] = -1
    tgt[3, 2:] = -1
    cfg = O.Config(vocab_size=V, lsm_weight=smooth, length_normalized_loss=nl)
    loss = O.label_smoothing_loss(cfg, x, tgt)
    loss.backward()
    acc = O.token_accuracy(x.detach().view(-1, V), tgt, -1)
    buf = torch.zeros(B * L, ldv, device=DEV)
    buf[:, :V] = x.detach().reshape(B * L, V).to(DEV)
    tg = cu(tgt.reshape(-1))
    out3 = torch.empty(3, device=DEV)
    ws = torch.empty(hip.lib().oe_lsm_workspace_bytes(B * L), dtype=torch.uint8, device=DEV)
    hip.call("oe_lsm_loss_fused", buf, ldv, B * L, V, tg, -1, smooth, int(nl), float(B), 1.0, 1, out3, ws)
    sync()
    o = out3.cpu()
    torch.testing.assert_close(o[0], loss.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(o[1] / o[2], acc, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(buf[:, :V].cpu().view(B, L, V), x.grad, rtol=1e-3, atol=1e-6)


# --------------------------------------------------------------- optimiser ----
def test_grad_norm_and_adam_match_torch():
    torch.manual_seed(19)
    n = 100003
    p0, g0 = torch.randn(n), torch.randn(n) * 3
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = torch.zeros(n + 1, device=DEV)[:n]          # 16-byte aligned arena
    p.copy_(p0)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(2, device=DEV)
    ws = torch.empty(hip.lib().oe_grad_norm_workspace_floats(), device=DEV)
    norm = torch.empty(1, device=DEV)
    lr_dev = torch.tensor([1e-3], device=DEV)
    for it in range(3):
        g = g0 * (it + 1)
        ref.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 5.0)
        opt.step()
        gd = cu(g)
        hip.call("oe_grad_norm", gd, n, ws, norm)
        hip.call("oe_adam_step", p, gd, m, v, n, lr_dev, 0.0, 0.9, 0.999, 1e-8, 5.0, norm, state)
        sync()
        torch.testing.assert_close(norm.cpu()[0], tn, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    # a non-finite gradient norm skips the step entirely (executor.py:59-60)
    before, step_before = p.clone(), state.clone()
    bad = cu(torch.full((n,), float("inf")))
    hip.call("oe_grad_norm", bad, n, ws, norm)
    hip.call("oe_adam_step", p, bad, m, v, n, lr_dev, 0.0, 0.9, 0.999, 1e-8, 5.0, norm, state)
    sync()
    assert torch.equal(p, before) and torch.equal(state, step_before)


# ------------------------------------------------------ fused feed forward -----
@pytest.mark.parametrize("rows,d,ff,act,prec,p_in,p_out,nout", [
    (7936, 256, 1024, 2, 3, 0.0, 0.0, 2),          # config 2, swish
    (1000, 256, 1024, 2, 3, 0.1, 0.1, 2),          # ragged last block (1000 = 31 * 32 + 8), both dropouts
    (77, 256, 512, 1, 3, 0.0, 0.0, 0),             # relu, no saved intermediates, ff = 4 tiles per wave
    (992, 256, 1024, 2, 1, 0.0, 0.0, 1),           # plain bf16 products, pre-activation only
    (333, 128, 512, 1, 3, 0.0, 0.25, 2),           # d = 128 (configs[0] width)
    (64, 128, 128, 0, 1, 0.0, 0.0, 2),             # one ff tile per wave, no activation
])
def test_fused_feed_forward_kernel(rows, d, ff, act, prec, p_in, p_out, nout):
    """oe_ffn_fwd (csrc/ffn.hip) against float64: y = residual + beta * drop(W2 drop(act(W1 x + b1)) + b2), the saved
    pre-activation and activation, and - with dropout - bit-identical masks to oe_gemm_f32's epilogue on the same tensors
    (the unfused backward regenerates them from the same seeds)."""
    torch.manual_seed(60)
    x = torch.randn(rows, d)
    w1, b1 = torch.randn(ff, d) / math.sqrt(d), torch.randn(ff) * 0.1
    w2, b2 = torch.randn(d, ff) / math.sqrt(ff), torch.randn(d) * 0.1
    res = torch.randn(rows, d)
    beta, s_in, s_out = 0.5, 0x1111, 0x2222
    L = hip.lib()
    assert L.oe_ffn_supported(d, ff, prec, act)
    nb = L.oe_ffn_packed_bytes(d, ff, prec)
    w1p, w2p = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    xd, w1d, w2d, b1d, b2d, resd = cu(x), cu(w1), cu(w2), cu(b1), cu(b2), cu(res)
    hip.call("oe_ffn_pack_weights", w1d, w2d, d, ff, prec, w1p, w2p)
    pre = torch.full((rows, ff), float("nan"), device=DEV) if nout >= 1 else None
    aout = torch.full((rows, ff), float("nan"), device=DEV) if nout == 2 else None
    y = torch.full((rows, d), float("nan"), device=DEV)
    ctr = torch.tensor([3], dtype=torch.int64, device=DEV)
    hip.ffn_fwd(xd, w1p, b1d, w2p, b2d, rows, d, ff, act, drop_in=p_in, seed_in=s_in, drop_out=p_out, seed_out=s_out, seed_dev=ctr,
                pre_out=pre, act_out=aout, residual=resd, ldr=d, beta=beta, y=y, precision=prec)
    sync()
    # the masks the unfused path would draw: dropout_scale of ones with the same (seed, counter, element index)
    ones_in, ones_out = torch.ones(rows, ff, device=DEV), torch.ones(rows, d, device=DEV)
    m_in, m_out = torch.empty_like(ones_in), torch.empty_like(ones_out)
    hip.call("oe_dropout_scale", ones_in, ones_in.numel(), ff, 1.0, p_in, s_in, ctr, None, m_in)
    hip.call("oe_dropout_scale", ones_out, ones_out.numel(), d, 1.0, p_out, s_out, ctr, None, m_out)
    sync()
    h = x.double() @ w1.double().t() + b1.double()
    a = (h * torch.sigmoid(h) if act == 2 else h.clamp(min=0) if act == 1 else h) * m_in.cpu().double()
    want = res.double() + beta * ((a @ w2.double().t() + b2.double()) * m_out.cpu().double())
    tol = dict(rtol=2e-4, atol=3e-5 * float(want.abs().max())) if prec == 3 else dict(rtol=3e-2, atol=2e-2 * float(want.abs().max()))
    torch.testing.assert_close(y.cpu().double(), want, **tol)
    if nout >= 1:
        torch.testing.assert_close(pre.cpu().double(), h, **(dict(rtol=2e-4, atol=3e-5 * float(h.abs().max())) if prec == 3 else dict(rtol=3e-2, atol=3e-2)))
    if nout == 2:
        torch.testing.assert_close(aout.cpu().double(), a, **(dict(rtol=2e-4, atol=3e-5 * float(h.abs().max())) if prec == 3 else dict(rtol=3e-2, atol=3e-2)))
        if p_in > 0:
            assert torch.equal(aout == 0, m_in == 0) or float(((aout == 0) != (m_in == 0)).float().mean()) < 1e-4      # (a itself can be exactly 0)


@pytest.mark.parametrize("rows,d,ff,act,prec,p_in", [(7936, 256, 1024, 2, 3, 0.1), (333, 256, 512, 1, 3, 0.0), (100, 128, 256, 2, 3, 0.2),
                                                      (4100, 256, 1024, 2, 1, 0.1), (64, 128, 128, 0, 3, 0.0)])
def test_fused_feed_forward_input_gradient_kernel(rows, d, ff, act, prec, p_in):
    """oe_ffn_bwd (csrc/ffn.hip, backward mode) against float64: dH = (dY W2) * mask * act'(pre), dX = dH W1, the mask being
    the one oe_ffn_fwd / oe_gemm_f32 draw for the same (seed, counter, element)."""
    torch.manual_seed(61)
    dy = torch.randn(rows, d)
    w1, w2 = torch.randn(ff, d) / math.sqrt(d), torch.randn(d, ff) / math.sqrt(ff)
    pre = torch.randn(rows, ff) * 1.5
    s_in = 0x3333
    L = hip.lib()
    nb = L.oe_ffn_packed_bytes(d, ff, prec)
    w2tp, w1tp = torch.empty(nb, dtype=torch.uint8, device=DEV), torch.empty(nb, dtype=torch.uint8, device=DEV)
    dyd, w1d, w2d, pred = cu(dy), cu(w1), cu(w2), cu(pre)
    hip.call("oe_ffn_pack_weights_bwd", w1d, w2d, d, ff, prec, w2tp, w1tp)
    dh = torch.full((rows, ff), float("nan"), device=DEV)
    dx = torch.full((rows, d), float("nan"), device=DEV)
    ctr = torch.tensor([5], dtype=torch.int64, device=DEV)
    hip.ffn_bwd(dyd, w2tp, w1tp, rows, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=ctr, pre=pred, dh=dh, dx=dx, precision=prec)
    sync()
    ones = torch.ones(rows, ff, device=DEV)
    m_in = torch.empty_like(ones)
    hip.call("oe_dropout_scale", ones, ones.numel(), ff, 1.0, p_in, s_in, ctr, None, m_in)
    sync()
    h = pre.double()
    sg = torch.sigmoid(h)
    dact = sg * (1 + h * (1 - sg)) if act == 2 else (h > 0).double() if act == 1 else torch.ones_like(h)
    want_dh = (dy.double() @ w2.double()) * m_in.cpu().double() * dact
    want_dx = want_dh @ w1.double()
    t = (lambda ref: dict(rtol=2e-4, atol=3e-5 * float(ref.abs().max()))) if prec == 3 else (lambda ref: dict(rtol=3e-2, atol=2e-2 * float(ref.abs().max())))
    torch.testing.assert_close(dh.cpu().double(), want_dh, **t(want_dh))
    torch.testing.assert_close(dx.cpu().double(), want_dx, **t(want_dx))


@pytest.mark.parametrize("rows,V,k", [(37, 3246, 10), (5, 100, 10), (3, 11000, 16), (2, 25000, 4), (9, 7, 7), (130, 65, 1)])
def test_topk_rows_kernel(rows, V, k):
    """oe_topk_rows against log_softmax_rows -> torch.topk (asr_model.py:251, 258, 358): same values, same indices."""
    from openeat_amd import ops
    g = torch.Generator(device="cpu").manual_seed(rows * 1000 + V)
    x = (torch.randn(rows, V, generator=g) * 3).cuda()
    for lsm in (False, True):
        ref = ops.log_softmax_rows(x) if lsm else x
        rv, ri = ref.topk(k, dim=-1)
        gv, gi = ops.topk_rows(x, k, log_softmax=lsm)
        assert gi.dtype == torch.int64 and gv.shape == (rows, k)
        if lsm:                                              # the two kernels' log-sum-exp may differ in the last bit
            torch.testing.assert_close(gv, rv, rtol=0, atol=2e-6)
        else:
            assert torch.equal(gv, rv)
        assert torch.equal(gi, ri)                           # randn values: no ties
    # the beam-search score matrix (asr_model.py:258): -inf entries and exact ties -> lowest index first, all indices distinct
    s = torch.full((4, 100), -float("inf"), device="cuda")
    s[0, [5, 17, 60]] = torch.tensor([1.0, 3.0, 2.0], device="cuda")
    s[1, :] = 0.5
    s[2, 40:] = torch.arange(60, device="cuda").float()
    gv, gi = ops.topk_rows(s, 10)
    assert gi[0, :3].tolist() == [17, 60, 5] and gv[0, :3].tolist() == [3.0, 2.0, 1.0] and bool(torch.isinf(gv[0, 3:]).all())
    assert len(set(gi[0].tolist())) == 10
    assert gi[1].tolist() == list(range(10)) and gi[2].tolist() == list(range(99, 89, -1))
    assert len(set(gi[3].tolist())) == 10
    # 3-D input, as the batched rescoring calls it
    x3 = torch.randn(3, 7, 50, device="cuda")
    gv, gi = ops.topk_rows(x3, 5, log_softmax=True)
    rv, ri = ops.log_softmax_rows(x3).topk(5, dim=-1)
    torch.testing.assert_close(gv, rv, rtol=0, atol=2e-6)
    assert torch.equal(gi, ri)


@pytest.mark.parametrize("B,T,V,beam,sharp", [(5, 60, 50, 10, 1.0), (8, 40, 6, 4, 0.3), (3, 120, 12, 10, 3.0), (4, 33, 40, 1, 1.0),
                                                (2, 50, 300, 16, 2.0), (6, 25, 5, 5, 0.0)])
def test_device_prefix_beam_equals_the_host_recursion(B, T, V, beam, sharp):
    """oe_ctc_prefix_beam (one wave per utterance) against oe_ctc_prefix_beam_host_batch, the bit-exact restatement of
    asr_model.py:359-396 that tests/test_oracle_golden pins: same n-best prefixes in the same order, scores equal to
    1e-9 (device exp/log), on small vocabularies where prefixes merge all the time, with exact ties (sharp = 0: every
    frame uniform), ragged lengths, beam 1 and 16."""
    from openeat_amd import hip, ops
    g = torch.Generator().manual_seed(B * 100 + T + V)
    logits = torch.randn(B, T, V, generator=g) * sharp
    logits[:, :, 0] += 1.0                                     # blanks are common, as in a CTC posterior
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = T
    top_p, top_i = ops.topk_rows(logits.cuda(), beam, log_softmax=True)
    want = hip.ctc_prefix_beam_host_batch(top_p.cpu(), top_i.cpu(), lens.tolist(), beam)
    got = hip.ctc_prefix_beam_device(top_p, top_i, lens.cuda(), beam)
    assert len(got) == B
    for b in range(B):
        assert [p for p, _ in got[b]] == [p for p, _ in want[b]], (b, got[b][:3], want[b][:3])
        for (_, s1), (_, s2) in zip(got[b], want[b]):
            assert s1 == s2 or abs(s1 - s2) < 1e-9 * max(1.0, abs(s2)), (b, s1, s2)
    # no lengths given: every utterance uses all T frames
    want_all = hip.ctc_prefix_beam_host_batch(top_p.cpu(), top_i.cpu(), [T] * B, beam)
    got_all = hip.ctc_prefix_beam_device(top_p, top_i, None, beam)
    assert [[p for p, _ in u] for u in got_all] == [[p for p, _ in u] for u in want_all]


def test_logprob_gather_kernel():
    """oe_logprob_gather == log_softmax_rows(...).gather(...), bit for bit (the same log-sum-exp arithmetic), plus the fixed
    <eos> column; out-of-range indices give 0."""
    from openeat_amd import ops
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(7, 13, 3246, generator=g) * 4).cuda()
    idx = torch.randint(0, 3246, (7, 13), generator=g).cuda()
    lp = ops.log_softmax_rows(x)
    a, b = ops.logprob_gather(x, idx, also=3245)
    assert torch.equal(a, lp.gather(2, idx.unsqueeze(2)).squeeze(2)) and torch.equal(b, lp[..., 3245])
    only = ops.logprob_gather(x, idx)
    assert torch.equal(only, a)
    bad = idx.clone()
    bad[0, 0], bad[1, 1] = -1, 3246
    z = ops.logprob_gather(x, bad)
    assert float(z[0, 0]) == 0.0 and float(z[1, 1]) == 0.0 and torch.equal(z[2:], a[2:])


@pytest.mark.parametrize("T", [300, 129, 97])
def test_attention_causal_hint_changes_nothing(T):
    """oe_attn_args.causal: with a (B, T, T) mask that is zero above the diagonal (plus ragged lengths) the forward kernel may
    skip the key blocks above the diagonal - output and log-sum-exp identical to the run without the hint, bit for bit."""
    torch.manual_seed(T)
    B, H, D = 3, 4, 64
    q, k, v = (torch.randn(B, T, H, D, device=DEV) for _ in range(3))
    lens = torch.tensor([T, T - 40, T // 2], device=DEV)
    idx = torch.arange(T, device=DEV)
    mask = ((idx[None, None, :] <= idx[None, :, None]) & (idx[None, None, :] < lens[:, None, None])).to(torch.uint8).contiguous()
    st = (T * H * D, H * D)
    outs = []
    for causal in (False, True):
        out = torch.full_like(q, float("nan"))
        lse = torch.full((B, H, T), float("nan"), device=DEV)
        a = hip.attn_args(q, k, v, out, lse, B, H, T, T, D, 1 / math.sqrt(D), q_strides=st, k_strides=st, v_strides=st, o_strides=st,
                          mask=mask, mask_strides=(T * T, T), precision=3, causal=causal)
        hip.attention_fwd(a)
        sync()
        outs.append((out, lse))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.softmax((torch.einsum("bihd,bjhd->bhij", q, k) / math.sqrt(D)).masked_fill(mask[:, None] == 0, float("-inf")), -1)
    want = torch.einsum("bhij,bjhd->bihd", ref, v)
    torch.testing.assert_close(outs[1][0], want, rtol=2e-4, atol=1e-4)
