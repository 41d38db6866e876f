"""ctypes binding of ``libopeneat_hip.so`` (C ABI declared in ``include/openeat_hip.h``).

This is the only door from Python into the compute path.  There is no CPU
fallback: if the library is missing or a call fails, an exception is raised.
PyTorch is used here only as the owner of device memory and of the current
HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

# When set to a list, every oe_gemm_f32 launch is bracketed by events on the launch stream and
# (start, end, algorithmic flops) is appended: bench.py's live roofline measurement.
PROFILE = None

# Default GEMM arithmetic: 0 = fp32-input MFMA (exact), 1 = bf16 inputs, 3 = 3-term bf16 split (~2^-17 per product),
# 6 = 6-term split of three exact bf16 pieces per operand (the fp32 product to within one fp32 rounding: bench.py's mode).
GEMM_PRECISION = int(os.environ.get("OE_GEMM_PRECISION", "0"))

# OE_HIP_LIB: diagnostic builds only (tools/gemm_stamps.py loads the stamped variant of the library)
_LIB_PATH = os.environ.get("OE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libopeneat_hip.so")
_lib = None

ACT = {"none": 0, None: 0, "relu": 1, "swish": 2, "tanh": 3, "hardtanh": 4, "selu": 5, "gelu": 6}
GATHER_NONE, GATHER_A, GATHER_B = 0, 1, 2

c_fp = C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("a", c_fp), ("lda", C.c_long), ("a_kmajor", C.c_int),
        ("b", c_fp), ("ldb", C.c_long), ("b_kmajor", C.c_int),
        ("c", c_fp), ("ldc", C.c_long),
        ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
        ("split_k", C.c_int),
        ("alpha", C.c_float), ("alpha_dev", c_fp),
        ("bias", c_fp),
        ("act", C.c_int),
        ("preact_out", c_fp), ("actgrad_in", c_fp), ("ld_aux", C.c_long),
        ("drop_p", C.c_float), ("seed", C.c_ulonglong), ("seed_dev", c_fp),
        ("rowmask", c_fp),
        ("residual", c_fp), ("ldr", C.c_long), ("res_row_mod", C.c_int), ("beta", C.c_float),
        ("accumulate", C.c_int), ("atomic_out", C.c_int),
        ("conv_gather", C.c_int), ("conv_t1", C.c_int), ("conv_f1", C.c_int), ("conv_t2", C.c_int),
        ("conv_f2", C.c_int), ("conv_c", C.c_int),
        ("a_colsum", c_fp),
        ("precision", C.c_int),
        ("conv_k", C.c_int), ("conv_s", C.c_int),
        ("conv_kh", C.c_int),
        ("out_scatter", C.c_int), ("sc_t1", C.c_int), ("sc_f1", C.c_int), ("sc_t2", C.c_int), ("sc_f2", C.c_int), ("sc_s", C.c_int),
        ("a_planes", c_fp), ("a_plane_stride", C.c_long), ("b_planes", c_fp), ("b_plane_stride", C.c_long),
        ("c_planes", c_fp), ("c_plane_stride", C.c_long), ("ldcp", C.c_long),
        ("conv_korder", C.c_int),
        ("actgrad_bf16", C.c_int),
        ("planes_k_padded", C.c_int),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", c_fp), ("q_bstride", C.c_long), ("q_rstride", C.c_long),
        ("k", c_fp), ("k_bstride", C.c_long), ("k_rstride", C.c_long),
        ("v", c_fp), ("v_bstride", C.c_long), ("v_rstride", C.c_long),
        ("out", c_fp), ("o_bstride", C.c_long), ("o_rstride", C.c_long),
        ("lse", c_fp),
        ("mask", c_fp), ("mask_bstride", C.c_long), ("mask_rstride", C.c_long),
        ("keybias", c_fp),
        ("B", C.c_int), ("H", C.c_int), ("T1", C.c_int), ("T2", C.c_int), ("D", C.c_int),
        ("scale", C.c_float),
        ("drop_p", C.c_float), ("seed", C.c_ulonglong), ("seed_dev", c_fp),
        ("d_out", c_fp), ("dq", c_fp), ("dk", c_fp), ("dv", c_fp), ("dkeybias", c_fp), ("delta", c_fp),
        ("precision", C.c_int), ("causal", C.c_int),
    ]


class LnPrologue(C.Structure):
    """oe_ln_prologue (include/openeat_hip.h)."""
    _fields_ = [
        ("dy", c_fp), ("x", c_fp), ("stats", c_fp), ("gamma", c_fp), ("add", c_fp),
        ("dx", c_fp), ("g", c_fp), ("ws", c_fp),
        ("g_alpha", C.c_float), ("g_p", C.c_float), ("g_seed", C.c_ulonglong), ("g_rowmask", c_fp), ("ln_rowmask", c_fp),
        ("beta", c_fp), ("gamma2", c_fp), ("stats2", c_fp), ("ws2", c_fp),
    ]

    def fill(self, ln):
        """ln: dict(dy, x, stats, gamma, add, dx, g, ws, alpha, p, seed, rowmask, ln_rowmask [, beta, gamma2, stats2, ws2: a pair])."""
        dp = lambda t: None if t is None else t.data_ptr()
        self.dy, self.x, self.stats, self.gamma, self.add = dp(ln["dy"]), dp(ln["x"]), dp(ln["stats"]), dp(ln["gamma"]), dp(ln.get("add"))
        self.dx, self.g, self.ws = dp(ln["dx"]), dp(ln["g"]), dp(ln["ws"])
        self.g_alpha, self.g_p, self.g_seed = ln["alpha"], ln["p"], ln["seed"]
        self.g_rowmask, self.ln_rowmask = dp(ln.get("rowmask")), dp(ln.get("ln_rowmask"))
        self.beta, self.gamma2, self.stats2, self.ws2 = dp(ln.get("beta")), dp(ln.get("gamma2")), dp(ln.get("stats2")), dp(ln.get("ws2"))


class LnfPrologue(C.Structure):
    """oe_lnf_prologue (include/openeat_hip.h)."""
    _fields_ = [("x", c_fp), ("gamma", c_fp), ("beta", c_fp), ("eps", C.c_float), ("y", c_fp), ("stats", c_fp), ("rowmask", c_fp),
                ("gamma2", c_fp), ("beta2", c_fp), ("eps2", C.c_float), ("u", c_fp), ("stats2", c_fp)]

    def fill(self, q):
        """q: dict(x, gamma, beta, eps, y, stats, rowmask)."""
        dp = lambda t: None if t is None else t.data_ptr()
        self.x, self.gamma, self.beta, self.eps = dp(q["x"]), dp(q["gamma"]), dp(q["beta"]), float(q["eps"])
        self.y, self.stats, self.rowmask = dp(q["y"]), dp(q["stats"]), dp(q.get("rowmask"))
        self.gamma2, self.beta2, self.eps2 = dp(q.get("gamma2")), dp(q.get("beta2")), float(q.get("eps2") or 0.0)
        self.u, self.stats2 = dp(q.get("u")), dp(q.get("stats2"))


class LnEpilogue(C.Structure):
    """oe_ln_epilogue (include/openeat_hip.h)."""
    _fields_ = [("x", c_fp), ("stats", c_fp), ("gamma", c_fp), ("beta", c_fp), ("act", C.c_int), ("dx", c_fp), ("ws", c_fp)]


class FfnArgs(C.Structure):
    _fields_ = [
        ("x", c_fp), ("ldx", C.c_long),
        ("w1p", c_fp), ("b1", c_fp), ("w2p", c_fp), ("b2", c_fp),
        ("rows", C.c_int), ("d", C.c_int), ("ff", C.c_int), ("act", C.c_int), ("precision", C.c_int),
        ("drop_in", C.c_float), ("seed_in", C.c_ulonglong), ("drop_out", C.c_float), ("seed_out", C.c_ulonglong),
        ("seed_dev", c_fp),
        ("pre_out", c_fp), ("act_out", c_fp),
        ("residual", c_fp), ("ldr", C.c_long), ("beta", C.c_float),
        ("y", c_fp), ("ldy", C.c_long),
        ("ln", LnPrologue), ("lnf", LnfPrologue),
    ]


class RowGemmArgs(C.Structure):
    """oe_rowgemm_args (include/openeat_hip.h)."""
    _fields_ = [
        ("x", c_fp), ("ldx", C.c_long),
        ("wp", c_fp), ("bias", c_fp),
        ("rows", C.c_int), ("k", C.c_int), ("n", C.c_int),
        ("drop_p", C.c_float), ("seed", C.c_ulonglong), ("seed_dev", c_fp),
        ("rowmask", c_fp),
        ("residual", c_fp), ("ldr", C.c_long), ("beta", C.c_float),
        ("y", c_fp), ("ldy", C.c_long),
        ("act", C.c_int), ("preact_out", c_fp), ("actgrad_in", c_fp), ("ld_aux", C.c_long),
        ("ln", LnPrologue), ("lnf", LnfPrologue), ("lne", LnEpilogue),
    ]


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load the HIP library once; fail loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipLibraryMissing(
                f"{_LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(openeat_amd has no CPU fallback)")
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def exported_symbols():
    """Names that include/openeat_hip.h declares (kept in sync by tests/test_abi.py)."""
    return list(_SIGNATURES)


class TnProblem(C.Structure):
    """oe_tn_problem (include/openeat_hip.h)."""
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("a_colsum", C.c_void_p), ("alpha_dev", C.c_void_p),
                ("lda", C.c_long), ("ldb", C.c_long), ("ldc", C.c_long), ("m", C.c_int), ("n", C.c_int), ("k", C.c_int),
                ("alpha", C.c_float), ("k_chunk", C.c_int), ("gx", C.c_int), ("gy", C.c_int), ("nz", C.c_int),
                ("block_start", C.c_int), ("reserved", C.c_int)]


I, L, F, P, U64, SZ = C.c_int, C.c_long, C.c_float, c_fp, C.c_ulonglong, C.c_size_t
_SIGNATURES = {
    "oe_last_error": (C.c_char_p, []),
    "oe_abi_version": (I, []),
    "oe_capture_unjoined_streams": (I, [P, C.POINTER(C.c_void_p), I, C.POINTER(I)]),
    "oe_stamp": (I, [P, I, P]),
    "oe_gemm_f32": (I, [C.POINTER(GemmArgs), P]),
    "oe_split_planes": (I, [P, L, L, L, P, L, L, P]),
    "oe_gemm_pl_launches": (L, []),
    "oe_gemm_hyb_launches": (L, []),
    "oe_gemm_pl_config": (I, [I, I, I, I]),
    "oe_gemm_pl_hybrid": (I, [I]),
    "oe_gemm_tn_grouped_plan": (I, [C.POINTER(TnProblem), I, I]),
    "oe_gemm_tn_grouped": (I, [P, I, I, I, P]),
    "oe_ffn_packed_bytes": (SZ, [I, I, I]),
    "oe_ffn_supported": (I, [I, I, I, I]),
    "oe_ffn_pack_weights": (I, [P, P, I, I, I, P, P, P]),
    "oe_ffn_pack_weights_bwd": (I, [P, P, I, I, I, P, P, P]),
    "oe_ffn_pack_weights_table": (I, [P, I, I, I, I, P]),
    "oe_ffn_fwd": (I, [C.POINTER(FfnArgs), P]),
    "oe_ffn_bwd": (I, [C.POINTER(FfnArgs), P]),
    "oe_ffn6_config": (I, [I]),
    "oe_rowgemm6_supported": (I, [I, I]),
    "oe_rowgemm6": (I, [C.POINTER(RowGemmArgs), P]),
    "oe_rowgemm6_pack_table": (I, [P, I, L, P]),
    "oe_rowgemm6_form": (I, [I, I, I]),
    "oe_colsum_f32": (I, [P, L, I, I, F, P, P, I, P]),
    "oe_layernorm_fwd": (I, [P, P, P, F, I, I, P, I, P, P, P]),
    "oe_layernorm_fwd_pl": (I, [P, P, P, F, I, I, P, I, P, P, P, L, P]),
    "oe_layernorm_bwd_workspace_floats": (SZ, [I, I]),
    "oe_layernorm_pair_fwd": (I, [P, P, P, F, P, P, F, I, I, P, P, P, P, P]),
    "oe_layernorm_pair_bwd_dx_drop": (I, [P, P, P, P, P, P, P, I, I, P, P, P, F, F, U64, P, P, P, P, P]),
    "oe_layernorm_bwd": (I, [P, P, P, P, I, P, I, I, P, P, P, P, P, P, P]),
    "oe_layernorm_bwd_dx": (I, [P, P, P, P, I, P, I, I, P, P, P, P, P]),
    "oe_layernorm_param_reduce": (I, [P, I, I, P, P, P]),
    "oe_layernorm_param_reduce_table": (I, [P, I, I, I, P]),
    "oe_layernorm_bwd_dx_drop": (I, [P, P, P, P, I, P, I, I, P, P, P, P, F, F, U64, P, P, P, P]),
    "oe_layernorm_bwd_dx_drop_pl": (I, [P, P, P, P, I, P, I, I, P, P, P, P, F, F, U64, P, P, P, P, L, P]),
    "oe_ctc_workspace_floats": (SZ, [I, I, I]),
    "oe_ctc_config": (I, [I, I]),
    "oe_ctc_loss_fused": (I, [P, L, I, I, I, P, P, I, P, F, P, P, P, P, P, P]),
    "oe_ctc_loss_fused_stats": (I, [P, L, I, I, I, P, P, I, P, F, P, P, P, P, P, P, I, P]),
    "oe_ctc_greedy": (I, [P, L, I, I, I, P, I, P, P, P, P]),
    "oe_attention_fwd": (I, [C.POINTER(AttnArgs), P]),
    "oe_attention_bwd": (I, [C.POINTER(AttnArgs), P]),
    "oe_relpos_prepare": (I, [P, L, L, P, L, P, P, I, I, I, I, F, P, P, P]),
    "oe_relpos_backward": (I, [P, P, P, L, L, P, L, P, P, I, I, I, I, F, P, P, L, P, P, P]),
    "oe_glu_fwd": (I, [P, L, I, P, P]),
    "oe_glu_bwd": (I, [P, P, L, I, P, P]),
    "oe_dropout_scale": (I, [P, L, I, F, F, U64, P, P, P, P]),
    "oe_embed_fwd": (I, [P, P, P, L, I, I, I, F, P, P]),
    "oe_embed_bwd": (I, [P, P, L, I, I, F, P, P]),
    "oe_swap_last2": (I, [P, L, I, I, P, I, P]),
    "oe_pad1_nhwc": (I, [P, I, I, I, I, P, P]),
    "oe_pad1_nhwc_planes": (I, [P, I, I, I, I, P, P, L, P]),
    "oe_conv_dgrad_k3s2_weights": (I, [P, I, P, P]),
    "oe_axpby": (I, [P, P, L, F, F, P, P, P]),
    "oe_loss_combine": (I, [P, P, P, F, F, F, P, P]),
    "oe_loss_combine_bwd": (I, [P, F, F, F, P, P, P, P]),
    "oe_act_fwd": (I, [P, L, I, P, P]),
    "oe_act_grad": (I, [P, P, L, I, P, P]),
    "oe_log_softmax": (I, [P, L, I, P, P]),
    "oe_topk_rows": (I, [P, L, I, I, I, P, P, P]),
    "oe_logprob_gather": (I, [P, L, I, P, I, P, P, P]),
    "oe_att_inputs": (I, [P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "oe_masked_softmax_fwd": (I, [P, P, L, L, I, I, I, I, F, U64, P, P, P, P]),
    "oe_masked_softmax_bwd": (I, [P, P, L, I, F, U64, P, P, P]),
    "oe_global_cmvn": (I, [P, P, P, L, I, P, P]),
    "oe_conv1_fwd": (I, [P, P, P, I, I, I, I, P, P]),
    "oe_conv1_fwd_pl": (I, [P, P, P, I, I, I, I, P, P, L, P]),
    "oe_conv1_wgrad": (I, [P, P, I, I, I, I, P, P, P]),
    "oe_col2im_relu": (I, [P, P, I, I, I, I, P, P]),
    "oe_col2im_relu_ks": (I, [P, P, I, I, I, I, I, I, P, P]),
    "oe_dwconv_glu_fwd": (I, [P, P, P, P, I, I, I, I, I, P, P]),
    "oe_dwconv_glu_ln_fwd": (I, [P, P, P, P, I, I, I, I, I, P, P, P, F, I, P, P, P]),
    "oe_dwconv_glu_bwd_workspace_floats": (SZ, [I, I, I, I]),
    "oe_dwconv_glu_bwd": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "oe_lsm_workspace_bytes": (SZ, [L]),
    "oe_lsm_loss_fused": (I, [P, L, L, I, P, I, F, I, F, F, I, P, P, P]),
    "oe_fbank": (I, [P, P, I, L, I, I, I, I, F, F, P, P, P, P, P, F, P, P, P, P]),
    "oe_fbank_dither": (I, [P, P, I, L, I, I, I, I, F, F, P, P, P, P, P, F, P, P, F, U64, P, P]),
    "oe_utt_normalize": (I, [P, P, I, I, I, P]),
    "oe_spec_augment": (I, [P, P, I, I, I, P, I, P, I, F, P]),
    "oe_spec_substitute": (I, [P, I, I, I, P, I, I, P]),
    "oe_feature_dither": (I, [P, P, I, I, I, F, U64, P]),
    "oe_speed_perturb": (I, [P, L, P, P, I, I, P, L, P, P]),
    "oe_ctc_prefix_beam_workspace_bytes": (SZ, [I, I, I]),
    "oe_ctc_prefix_beam": (I, [P, P, I, I, P, I, I, P, P, P, P, P]),
    "oe_ctc_prefix_beam_host": (I, [P, P, I, I, I, P, P, P]),
    "oe_ctc_prefix_beam_host_batch": (I, [P, P, I, I, P, I, I, P, P, P, I]),
    "oe_grad_norm_workspace_floats": (SZ, []),
    "oe_grad_norm": (I, [P, L, P, P, P]),
    "oe_adam_step": (I, [P, P, P, P, L, P, F, F, F, F, F, P, P, P]),
}


def _declare(l):
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().oe_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def capture_unjoined_streams(origin: "torch.cuda.Stream", sides):
    """Streams among `sides` whose captured work the capturing stream `origin` is not yet ordered after."""
    sides = [s for s in sides if s is not None]
    arr = (C.c_void_p * max(len(sides), 1))(*[s.cuda_stream for s in sides])
    first = I(-1)
    n = lib().oe_capture_unjoined_streams(C.c_void_p(origin.cuda_stream), arr, len(sides), C.byref(first))
    if n < 0:
        check(n, "oe_capture_unjoined_streams")
    return n, (sides[first.value] if n > 0 else None)


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise TypeError(f"{name}: expected a float32 CUDA tensor, got {t.dtype} on {t.device}")


# --------------------------------------------------------------------------- #
# thin typed wrappers
# --------------------------------------------------------------------------- #
def gemm(a, b, c, m, n, k, *, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, split_k=1, alpha=1.0, alpha_dev=None,
         bias=None, act=0, preact_out=None, actgrad_in=None, ld_aux=0, drop_p=0.0, seed=0, seed_dev=None, rowmask=None,
         residual=None, ldr=0, res_row_mod=0, beta=1.0, accumulate=False, atomic_out=False, conv=None, conv_gather=GATHER_NONE, precision=None, a_colsum=None,
         conv_kh=0, scatter=None, a_planes=None, b_planes=None, c_planes=None, conv_korder=0, actgrad_bf16=False):
    """a_planes / b_planes / c_planes: Planes (openeat_amd.planes) of A / B (pre-split copies, precision 6) and for the output."""
    g = GemmArgs()
    # a / b None: the operand exists as pre-split planes only (a_planes / b_planes); the launch fails if gemm_pl.hip declines
    g.a, g.lda, g.a_kmajor = (None if a is None else a.data_ptr()), lda, int(a_kmajor)
    g.b, g.ldb, g.b_kmajor = (None if b is None else b.data_ptr()), ldb, int(b_kmajor)
    g.c, g.ldc = c.data_ptr(), ldc
    g.m, g.n, g.k, g.split_k = m, n, k, split_k
    g.alpha = alpha
    g.alpha_dev = None if alpha_dev is None else alpha_dev.data_ptr()
    g.bias = None if bias is None else bias.data_ptr()
    g.act = act
    g.preact_out = None if preact_out is None else preact_out.data_ptr()
    g.actgrad_in = None if actgrad_in is None else actgrad_in.data_ptr()
    g.actgrad_bf16 = int(actgrad_bf16)
    g.ld_aux = ld_aux
    g.drop_p, g.seed = drop_p, seed
    g.seed_dev = None if seed_dev is None else seed_dev.data_ptr()
    g.rowmask = None if rowmask is None else rowmask.data_ptr()
    g.residual = None if residual is None else residual.data_ptr()
    g.ldr, g.beta, g.res_row_mod = ldr, beta, res_row_mod
    g.accumulate, g.atomic_out = int(accumulate), int(atomic_out)
    g.conv_gather = conv_gather
    g.precision = GEMM_PRECISION if precision is None else precision
    g.a_colsum = None if (a_colsum is None or g.precision == 0) else a_colsum.data_ptr()
    g.conv_kh = conv_kh
    g.conv_korder = conv_korder
    if g.precision == 6:
        if a_planes is not None and b_planes is not None:
            g.a_planes, g.a_plane_stride = a_planes.ptr, a_planes.stride
            g.b_planes, g.b_plane_stride = b_planes.ptr, b_planes.stride
            # a weight gradient's reduction may run to the next multiple of 16 when both buffers were allocated with zero pad rows
            g.planes_k_padded = int(a_kmajor and b_kmajor and getattr(a_planes, "kpad", False) and getattr(b_planes, "kpad", False))
        elif b_planes is not None and a is not None:             # the weight operand alone (gemm_hyb.hip)
            g.b_planes, g.b_plane_stride = b_planes.ptr, b_planes.stride
        if c_planes is not None:
            g.c_planes, g.c_plane_stride, g.ldcp = c_planes.ptr, c_planes.stride, c_planes.ld
    if scatter is not None:                      # (T1, F1, T2, F2, S): output rows (b, t, f) -> (b*T1 + S*t)*F1 + S*f
        g.out_scatter = 1
        g.sc_t1, g.sc_f1, g.sc_t2, g.sc_f2, g.sc_s = scatter
    if conv is not None:                         # (T1, F1, T2, F2, C) or (T1, F1, T2, F2, C, kernel size, stride)
        g.conv_t1, g.conv_f1, g.conv_t2, g.conv_f2, g.conv_c = conv[:5]
        g.conv_k, g.conv_s = (conv[5], conv[6]) if len(conv) > 5 else (0, 0)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().oe_gemm_f32(C.byref(g), stream()), "oe_gemm_f32")
        e1.record()
        PROFILE.append((e0, e1, 2.0 * m * n * k, (m, n, k, int(a_kmajor), int(b_kmajor), int(conv_gather), int(split_k))))
        return
    check(lib().oe_gemm_f32(C.byref(g), stream()), "oe_gemm_f32")


def tn_grouped_plan(problems, target_blocks: int = 384):
    """problems: list of dicts(dy (K, M), x (K, N), out (M, N) accumulated, alpha, alpha_dev, bias_out) ->
    (host bytes of the filled oe_tn_problem array, total blocks), or None if one of them does not qualify."""
    n = len(problems)
    arr = (TnProblem * n)()
    dp = lambda t: None if t is None else t.data_ptr()
    for i, q in enumerate(problems):
        dy, x, out = q["dy"], q["x"], q["out"]
        p = arr[i]
        p.a, p.b, p.c, p.a_colsum, p.alpha_dev = dy.data_ptr(), x.data_ptr(), out.data_ptr(), dp(q.get("bias_out")), dp(q.get("alpha_dev"))
        p.lda, p.ldb, p.ldc = dy.stride(0), x.stride(0), out.stride(0)
        p.m, p.n, p.k, p.alpha = dy.shape[1], x.shape[1], dy.shape[0], float(q.get("alpha", 1.0))
    total = lib().oe_gemm_tn_grouped_plan(arr, n, int(target_blocks))
    if total < 0:
        return None
    return bytes(arr), total


def tn_grouped_launch(table_dev: torch.Tensor, n: int, total_blocks: int, precision=None, flops: float = 0.0):
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().oe_gemm_tn_grouped(C.c_void_p(table_dev.data_ptr()), n, total_blocks, GEMM_PRECISION if precision is None else precision,
                                       stream()), "oe_gemm_tn_grouped")
        e1.record()
        PROFILE.append((e0, e1, flops, ("tn_grouped", n, total_blocks, 0, 1, 1, 0)))
        return
    check(lib().oe_gemm_tn_grouped(C.c_void_p(table_dev.data_ptr()), n, total_blocks, GEMM_PRECISION if precision is None else precision,
                                   stream()), "oe_gemm_tn_grouped")


def ffn_fwd(x2, w1p, b1, w2p, b2, rows, d, ff, act, *, drop_in=0.0, seed_in=0, drop_out=0.0, seed_out=0, seed_dev=None, pre_out=None,
            act_out=None, residual=None, ldr=0, beta=1.0, y=None, precision=None, lnf=None):
    """lnf: LayerNorm-forward prologue (LnfPrologue.fill) - the rows of x2 are then made by the kernel (x2 = lnf["y"], written too)."""
    a = FfnArgs()
    dp = lambda t: None if t is None else t.data_ptr()
    a.x, a.ldx = x2.data_ptr(), x2.stride(0)
    if lnf is not None:
        a.lnf.fill(lnf)
    a.w1p, a.b1, a.w2p, a.b2 = w1p.data_ptr(), dp(b1), w2p.data_ptr(), dp(b2)
    a.rows, a.d, a.ff, a.act = rows, d, ff, act
    a.precision = GEMM_PRECISION if precision is None else precision
    a.drop_in, a.seed_in, a.drop_out, a.seed_out, a.seed_dev = drop_in, seed_in, drop_out, seed_out, dp(seed_dev)
    a.pre_out, a.act_out, a.residual, a.ldr, a.beta = dp(pre_out), dp(act_out), dp(residual), ldr, beta
    a.y, a.ldy = y.data_ptr(), y.stride(0)
    if PROFILE is not None:                      # both GEMMs of the feed-forward are in this one launch: 2 x 2*rows*d*ff
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().oe_ffn_fwd(C.byref(a), stream()), "oe_ffn_fwd")
        e1.record()
        PROFILE.append((e0, e1, 4.0 * rows * d * ff, ("ffn_fwd", rows, d, ff, 0, 0, 1)))
        return
    check(lib().oe_ffn_fwd(C.byref(a), stream()), "oe_ffn_fwd")


def ffn_bwd(dy2, w2tp, w1tp, rows, d, ff, act, *, drop_in=0.0, seed_in=0, seed_dev=None, pre=None, dh=None, dx=None, precision=None, ln=None):
    """dH = (dY W2) * mask * act'(pre), dX = dH W1 in one launch (oe_ffn_bwd).  ln: LayerNorm-backward prologue (LnPrologue.fill) -
    the rows of dY are then made by the kernel (dy2 is where they are ALSO written: ln["g"])."""
    a = FfnArgs()
    dp = lambda t: None if t is None else t.data_ptr()
    a.x, a.ldx = dy2.data_ptr(), dy2.stride(0)
    if ln is not None:
        a.ln.fill(ln)
    a.w1p, a.b1, a.w2p, a.b2 = w2tp.data_ptr(), None, w1tp.data_ptr(), None
    a.rows, a.d, a.ff, a.act = rows, d, ff, act
    a.precision = GEMM_PRECISION if precision is None else precision
    a.drop_in, a.seed_in, a.drop_out, a.seed_out, a.seed_dev = drop_in, seed_in, 0.0, 0, dp(seed_dev)
    a.pre_out, a.act_out, a.residual, a.ldr, a.beta = pre.data_ptr(), dh.data_ptr(), None, 0, 1.0
    a.y, a.ldy = dx.data_ptr(), dx.stride(0)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().oe_ffn_bwd(C.byref(a), stream()), "oe_ffn_bwd")
        e1.record()
        PROFILE.append((e0, e1, 4.0 * rows * d * ff, ("ffn_bwd", rows, d, ff, 0, 0, 1)))
        return
    check(lib().oe_ffn_bwd(C.byref(a), stream()), "oe_ffn_bwd")


def rowgemm6(x, wp, y, rows, k, n, *, bias=None, drop_p=0.0, seed=0, seed_dev=None, rowmask=None, residual=None, ldr=0, beta=1.0,
             act=0, preact_out=None, actgrad_in=None, ld_aux=0, ln=None, lnf=None, lne=None):
    """y = residual + beta * rowmask * dropout(act(x @ Wg^T + bias)) on the row-block / tile kernels (oe_rowgemm6; wp = packed Wg)."""
    a = RowGemmArgs()
    dp = lambda t: None if t is None else t.data_ptr()
    a.x, a.ldx, a.wp, a.bias = (0 if x is None else x.data_ptr()), (k if x is None else x.stride(0)), wp.data_ptr(), dp(bias)
    a.rows, a.k, a.n = rows, k, n
    a.drop_p, a.seed, a.seed_dev, a.rowmask = drop_p, seed, dp(seed_dev), dp(rowmask)
    a.residual, a.ldr, a.beta = dp(residual), ldr, beta
    a.y, a.ldy = y.data_ptr(), y.stride(0)
    a.act, a.preact_out, a.actgrad_in, a.ld_aux = act, dp(preact_out), dp(actgrad_in), ld_aux
    if ln is not None:        # LayerNorm-backward prologue (LnPrologue.fill)
        a.ln.fill(ln)
    if lnf is not None:       # LayerNorm-forward prologue (LnfPrologue.fill)
        a.lnf.fill(lnf)
    if lne is not None:       # LayerNorm-backward epilogue: dict(x, stats, gamma, beta, act, dx, ws)
        a.lne.x, a.lne.stats, a.lne.gamma, a.lne.beta = dp(lne["x"]), dp(lne["stats"]), dp(lne["gamma"]), dp(lne["beta"])
        a.lne.act, a.lne.dx, a.lne.ws = int(lne["act"]), dp(lne["dx"]), dp(lne["ws"])
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib().oe_rowgemm6(C.byref(a), stream()), "oe_rowgemm6")
        e1.record()
        PROFILE.append((e0, e1, 2.0 * rows * k * n, ("rowgemm6", rows, n, k, 0, 0, 1)))
        return
    check(lib().oe_rowgemm6(C.byref(a), stream()), "oe_rowgemm6")


def call(name, *args):
    """Invoke a C-ABI entry point with (tensor | None | scalar) arguments on the current stream."""
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            conv.append(C.c_void_p(a.data_ptr()))
        else:
            conv.append(a)
    check(getattr(lib(), name)(*conv, stream()), name)


def attn_args(q, k, v, out, lse, B, H, T1, T2, D, scale, *, q_strides, k_strides, v_strides, o_strides, mask=None,
              mask_strides=(0, 0), keybias=None, drop_p=0.0, seed=0, seed_dev=None, d_out=None, dq=None, dk=None, dv=None,
              dkeybias=None, delta=None, precision=None, causal=False):
    a = AttnArgs()
    a.precision = GEMM_PRECISION if precision is None else precision
    dp = lambda t: None if t is None else t.data_ptr()
    a.q, (a.q_bstride, a.q_rstride) = dp(q), q_strides
    a.k, (a.k_bstride, a.k_rstride) = dp(k), k_strides
    a.v, (a.v_bstride, a.v_rstride) = dp(v), v_strides
    a.out, (a.o_bstride, a.o_rstride) = dp(out), o_strides
    a.lse = dp(lse)
    a.mask, (a.mask_bstride, a.mask_rstride) = dp(mask), mask_strides
    a.keybias = dp(keybias)
    a.B, a.H, a.T1, a.T2, a.D = B, H, T1, T2, D
    a.scale, a.drop_p, a.seed = scale, drop_p, seed
    a.seed_dev = dp(seed_dev)
    a.d_out, a.dq, a.dk, a.dv, a.dkeybias, a.delta = dp(d_out), dp(dq), dp(dk), dp(dv), dp(dkeybias), dp(delta)
    a.causal = int(bool(causal))
    return a


def attention_fwd(a: AttnArgs):
    check(lib().oe_attention_fwd(C.byref(a), stream()), "oe_attention_fwd")


def attention_bwd(a: AttnArgs):
    check(lib().oe_attention_bwd(C.byref(a), stream()), "oe_attention_bwd")


def ctc_prefix_beam_device(top_logp: torch.Tensor, top_idx: torch.Tensor, lens: Optional[torch.Tensor], beam: int, raw: bool = False):
    """top_logp (B, T, beam) float32 / top_idx (B, T, beam) int64 CUDA tensors (ops.topk_rows), lens (B) int32 CUDA or None ->
    per utterance [(prefix tuple, score)], as ctc_prefix_beam_host_batch returns them.  One kernel, one wave per
    utterance; one device-to-host copy of the n-best lists.  raw=True: no copy at all - the device tensors
    (prefixes (B, beam, T) int32, lengths (B, beam) int32 with -1 for missing entries, scores (B, beam) float64) and a
    status word tensor the caller checks after its own synchronisation."""
    if not (top_logp.is_cuda and top_idx.is_cuda and top_logp.dtype == torch.float32 and top_idx.dtype == torch.int64):
        raise TypeError("ctc_prefix_beam_device: float32 / int64 CUDA tensors required")
    top_logp, top_idx = top_logp.contiguous(), top_idx.contiguous()
    B, T = top_logp.shape[0], top_logp.shape[1]
    ml = max(T, 1)
    dev = top_logp.device
    ws = torch.empty(lib().oe_ctc_prefix_beam_workspace_bytes(B, T, beam) // 4, dtype=torch.int32, device=dev)
    ws[-1:].zero_()
    prefixes = torch.zeros(B, beam, ml, dtype=torch.int32, device=dev)
    plen = torch.empty(B, beam, dtype=torch.int32, device=dev)
    scores = torch.empty(B, beam, dtype=torch.float64, device=dev)
    call("oe_ctc_prefix_beam", top_logp, top_idx, B, T, lens, beam, ml, ws, prefixes, plen, scores)
    if raw:
        return prefixes, plen, scores, ws[-1:]
    prefixes, plen, scores, bad = prefixes.cpu().numpy(), plen.cpu().numpy(), scores.cpu().numpy(), int(ws[-1])
    if bad:
        raise RuntimeError("oe_ctc_prefix_beam: a prefix exceeded max_len")
    return [[(tuple(prefixes[b, i, : plen[b, i]].tolist()), float(scores[b, i])) for i in range(beam) if plen[b, i] >= 0]
            for b in range(B)]


def ctc_prefix_beam_host_batch(top_logp: torch.Tensor, top_idx: torch.Tensor, lens, beam: int, n_threads: int = 0):
    """top_logp (B, T, beam) float32 / top_idx (B, T, beam) int64 CPU tensors, lens[b] valid frames ->
    per utterance [(prefix tuple, score)] (native host code, utterances spread over host threads)."""
    import numpy as np
    lp = np.ascontiguousarray(top_logp.numpy())
    ix = np.ascontiguousarray(top_idx.numpy())
    B, T = lp.shape[0], lp.shape[1]
    ln = np.ascontiguousarray(np.asarray(lens, dtype=np.int32))
    ml = max(T, 1)
    prefixes = np.zeros((B, beam, ml), dtype=np.int32)
    plen = np.zeros((B, beam), dtype=np.int32)
    scores = np.zeros((B, beam), dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().oe_ctc_prefix_beam_host_batch(vp(lp), vp(ix), B, T, vp(ln), beam, ml, vp(prefixes), vp(plen), vp(scores), n_threads),
          "oe_ctc_prefix_beam_host_batch")
    return [[(tuple(prefixes[b, i, : plen[b, i]].tolist()), float(scores[b, i])) for i in range(beam) if plen[b, i] >= 0]
            for b in range(B)]


def ctc_prefix_beam_host(top_logp: torch.Tensor, top_idx: torch.Tensor, beam: int):
    """top_logp (T, beam) float32 / top_idx (T, beam) int64 CPU tensors -> [(prefix tuple, score)] (native host code)."""
    import numpy as np
    lp = top_logp.contiguous().numpy()
    ix = top_idx.contiguous().numpy()
    T = lp.shape[0]
    prefixes = np.zeros((beam, max(T, 1)), dtype=np.int32)
    lens = np.zeros(beam, dtype=np.int32)
    scores = np.zeros(beam, dtype=np.float64)
    rc = lib().oe_ctc_prefix_beam_host(lp.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), T, beam, max(T, 1),
                                       prefixes.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                       scores.ctypes.data_as(C.c_void_p))
    check(rc, "oe_ctc_prefix_beam_host")
    return [(tuple(prefixes[i, : lens[i]].tolist()), float(scores[i])) for i in range(beam) if lens[i] >= 0]
