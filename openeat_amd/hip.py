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

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libopeneat_hip.so")
_lib = None

ACT = {"none": 0, None: 0, "relu": 1, "swish": 2}
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
        ("drop_p", C.c_float), ("seed", C.c_ulonglong),
        ("rowmask", c_fp),
        ("residual", c_fp), ("ldr", C.c_long), ("beta", C.c_float),
        ("accumulate", C.c_int), ("atomic_out", C.c_int),
        ("conv_gather", C.c_int), ("conv_t1", C.c_int), ("conv_f1", C.c_int), ("conv_t2", C.c_int),
        ("conv_f2", C.c_int), ("conv_c", C.c_int),
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


I, L, F, P, U64, SZ = C.c_int, C.c_long, C.c_float, c_fp, C.c_ulonglong, C.c_size_t
_SIGNATURES = {
    "oe_last_error": (C.c_char_p, []),
    "oe_abi_version": (I, []),
    "oe_gemm_f32": (I, [C.POINTER(GemmArgs), P]),
    "oe_colsum_f32": (I, [P, L, I, I, F, P, P, I, P]),
    "oe_layernorm_fwd": (I, [P, P, P, F, I, I, P, P, P, P]),
    "oe_layernorm_bwd": (I, [P, P, P, P, I, I, P, P, P, P, P]),
    "oe_ctc_workspace_floats": (SZ, [I, I, I]),
    "oe_ctc_loss_fused": (I, [P, L, I, I, I, P, P, I, P, F, P, P, P, P, P]),
    "oe_ctc_greedy": (I, [P, L, I, I, I, P, I, P, P, P, P]),
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
         bias=None, act=0, preact_out=None, actgrad_in=None, ld_aux=0, drop_p=0.0, seed=0, rowmask=None,
         residual=None, ldr=0, beta=1.0, accumulate=False, atomic_out=False, conv=None, conv_gather=GATHER_NONE):
    g = GemmArgs()
    g.a, g.lda, g.a_kmajor = a.data_ptr(), lda, int(a_kmajor)
    g.b, g.ldb, g.b_kmajor = b.data_ptr(), ldb, int(b_kmajor)
    g.c, g.ldc = c.data_ptr(), ldc
    g.m, g.n, g.k, g.split_k = m, n, k, split_k
    g.alpha = alpha
    g.alpha_dev = None if alpha_dev is None else alpha_dev.data_ptr()
    g.bias = None if bias is None else bias.data_ptr()
    g.act = act
    g.preact_out = None if preact_out is None else preact_out.data_ptr()
    g.actgrad_in = None if actgrad_in is None else actgrad_in.data_ptr()
    g.ld_aux = ld_aux
    g.drop_p, g.seed = drop_p, seed
    g.rowmask = None if rowmask is None else rowmask.data_ptr()
    g.residual = None if residual is None else residual.data_ptr()
    g.ldr, g.beta = ldr, beta
    g.accumulate, g.atomic_out = int(accumulate), int(atomic_out)
    g.conv_gather = conv_gather
    if conv is not None:
        g.conv_t1, g.conv_f1, g.conv_t2, g.conv_f2, g.conv_c = conv
    check(lib().oe_gemm_f32(C.byref(g), stream()), "oe_gemm_f32")
