"""Autograd-level composition of the HIP kernels.

Each ``torch.autograd.Function`` below is one block of the reference's hot path
(feed-forward, attention, conv module, subsampling, CTC head, label-smoothing
head ...) whose forward AND backward are sequences of C-ABI calls
(``openeat_amd.hip``).  PyTorch only owns the buffers and the tape.  No op
here has a CPU path: tensors must be float32 CUDA tensors.
"""
from __future__ import annotations

import itertools
import os
import math
import weakref
from typing import Optional

import torch

from . import hip
from . import arena as _arena
from . import planes as _planes

ACT_NONE, ACT_RELU, ACT_SWISH, ACT_TANH, ACT_HARDTANH, ACT_SELU, ACT_GELU = 0, 1, 2, 3, 4, 5, 6      # oe_common.h
ACT_IDS = {"relu": ACT_RELU, "swish": ACT_SWISH, "tanh": ACT_TANH, "hardtanh": ACT_HARDTANH, "selu": ACT_SELU, "gelu": ACT_GELU,
           "none": ACT_NONE}
# fused into GEMM epilogues; the others (never used by a shipped config) run as separate elementwise kernels so that
# their erf / tanh / exp code does not cost every GEMM kernel registers
GEMM_FUSED_ACTS = (ACT_NONE, ACT_RELU, ACT_SWISH)


# --------------------------------------------------------------------------- #
# dropout seeds: a host counter gives every dropout site a distinct stream; an
# optional device step counter (set by the training engine) is mixed in inside
# the kernels so that a captured HIP graph still draws fresh masks per replay.
# --------------------------------------------------------------------------- #
_seed_counter = itertools.count(0x5EED0001)
_seed_dev: Optional[torch.Tensor] = None


def next_seed() -> int:
    return next(_seed_counter) * 0x9E3779B1 & 0xFFFFFFFFFFFF


def set_seed_device_counter(t: Optional[torch.Tensor]):
    global _seed_dev
    _seed_dev = t


def manual_seed(seed: int):
    global _seed_counter
    _seed_counter = itertools.count(0x5EED0001 + (seed & 0xFFFFFFF) * 7919)


def _chk(t: torch.Tensor, name: str, keep_pending: bool = False) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise TypeError(f"{name}: openeat_amd ops need float32 CUDA tensors (got {t.dtype} on {t.device}); "
                        "there is no CPU fallback")
    if _PENDING_LNF and not keep_pending:
        # the output of a parked LayerNorm forward (PreNormFn) reaching an op that does not make it itself: launch it now
        pend = _PENDING_LNF.pop(t.data_ptr(), None)
        if pend is not None:
            _resolve_lnf(pend)
    return t.contiguous()


# The LayerNorm FORWARD of a pre-norm fork as the prologue of the GEMM that consumes it (oe_rowgemm6 / oe_ffn_fwd's lnf arguments):
# EncoderLayer promises (fuse_fwd) that the normed branch goes straight into an op of this module; PreNormFn.forward then parks the
# launch here, keyed by the address of its (still unwritten) output; the consumer's first kernel - the feed-forward, the fused
# q / k / v projection, the conv module's pointwise_conv1 - makes the rows on its way to LDS and writes y and the statistics for the
# backward.  Any other op of this module that receives a parked tensor launches the LayerNorm on its own first (_chk); leftovers are
# launched where the step's bookkeeping is reset (predrop_clear).  36 launches per step at config 2.
LN_FWD_FUSE = os.environ.get("OE_LN_FWD_FUSE", "1") == "1"
LN_FWD_FUSED_LAUNCHES = 0       # (tests)
_PENDING_LNF = {}


def _resolve_lnf(pend):
    if pend.get("done"):
        return
    pend["done"] = True
    if pend.get("gamma2") is not None:
        hip.call("oe_layernorm_pair_fwd", pend["x"], pend["gamma"], pend["beta"], pend["eps"], pend["gamma2"], pend["beta2"], pend["eps2"], pend["rows"],
                 pend["d"], pend["u"], pend["stats"], pend["y"], pend["stats2"])
        return
    _ln_fwd(pend["x"], pend["gamma"], pend["beta"], pend["eps"], pend["rows"], pend["d"], pend["rowmask"], ACT_NONE, pend["y"], pend["stats"])


def resolve_pending_lnf():
    n = 0
    for pend in list(_PENDING_LNF.values()):
        if not pend.get("done"):
            _resolve_lnf(pend)
            n += 1
    _PENDING_LNF.clear()
    return n


def _lnf_take(t):
    """The parked LayerNorm forward whose output is t (removed from the registry), or None."""
    return _PENDING_LNF.pop(t.data_ptr(), None) if _PENDING_LNF else None


def _new(*shape, like: torch.Tensor, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, device=like.device, dtype=torch.float32)


# --------------------------------------------------------------------------- #
# GEMM helpers
# --------------------------------------------------------------------------- #
def _split_k(out_rows: int, out_cols: int, k: int) -> int:
    """Split of the reduction dimension of a weight-gradient GEMM.  Large outputs use 128x128 tiles
    (fewer operand re-reads) with just enough splits to cover the 256 CUs; small ones 64x64 tiles with
    ~3 blocks per CU.  K-chunks stay >= 256 and the atomic traffic (splits x output bytes) bounded."""
    if hip.GEMM_PRECISION != 0 and out_rows >= 128 and out_cols >= 128:
        t128 = -(-out_rows // 128) * -(-out_cols // 128)
        sk = max(1, 256 // t128)          # at most one 128x128 block per CU (the ring kernel holds one per CU)
        if t128 * sk >= 200 and sk <= 24:
            return int(max(1, min(sk, k // 256 if k >= 512 else 1)))
    tiles = -(-out_rows // 64) * -(-out_cols // 64)
    want = max(1, 768 // max(tiles, 1))
    return int(max(1, min(want, 24, k // 256 if k >= 512 else 1)))


def _operand_planes(act2d, w2d):
    """Pre-split copies (planes.py) of an activation operand and a weight operand of one GEMM - both or neither."""
    if not _planes.active():
        # the weight alone, from the arena's planes (gemm_hyb.hip splits the activation on the fragment)
        return None, (_planes.arena_weight(w2d) if (_planes.weights_presplit() and act2d.shape[0] >= _planes.HYB_MIN_ROWS) else None)
    ap = _planes.of(act2d, make=_planes.split_activations())
    if ap is None:
        return None, None
    bp = _planes.weight(w2d)
    return (ap, bp) if bp is not None else (None, None)


# ---- the short-reduction Linears as row-block GEMMs (csrc/ffn6.hip::rowgemm6_kernel) ---------------------------------------
# x W^T and dy W with k in {256, 512} and an output width that is a multiple of 128, from ROWGEMM_MIN_ROWS rows on, in precision 6:
# the weight is consumed as packed fragments, refreshed for EVERY registered matrix by one table-driven launch per weights
# generation (as the feed-forwards' packs).  Anything the kernel's epilogue does not do stays on oe_gemm_f32.
ROWGEMM = os.environ.get("OE_ROWGEMM", "1") == "1"
ROWGEMM_MIN_ROWS = int(os.environ.get("OE_ROWGEMM_MIN_ROWS", "4096"))
ROWGEMM_LAUNCHES = 0       # (tests: which path a call took)
_ROW_EPI_OK = {"drop_p", "seed", "seed_dev", "rowmask", "residual", "ldr", "beta"}
# the tile form (few rows: oe_rowgemm6_form = 2, any k in {256, 512, 768, 1024}) also has oe_gemm_f32's activation epilogue
_ROWTILE_EPI_OK = _ROW_EPI_OK | {"act", "preact_out", "actgrad_in", "ld_aux"}
ROWTILE = os.environ.get("OE_ROWTILE", "1") == "1"


class _PackTable:
    """Device table of a table-driven pack launch (oe_rowgemm6_pack_table / oe_ffn_pack_weights_table) with the lifetime rules a
    CAPTURED launch needs: the device tensor is allocated once and only ever updated in place, rows are append-only (a captured
    launch bakes in the table's address and its row count: it keeps refreshing exactly the rows that existed when it was
    captured), and a row whose weights died is NEUTRALISED (source pointer 0: the kernel skips it) instead of removed - its pack
    buffer goes back to the allocator, and a replay must neither read the dead weights (their memory may be unmapped after
    empty_cache: a GPU fault) nor write the buffer.  The first version rebuilt the table tensor at every registration: the decode
    graphs of bench.py replayed a pack launch over a freed table (memory access fault at address 0x1e000, round 4)."""
    CAP = 4096

    def __init__(self, words):
        self.words, self.host, self.dev, self.n, self.dirty = words, None, None, 0, False
        self.event = self.stream = None
        self.rows = {}                     # key -> row index
        self.entries = {}                  # key -> entry dict (alive ones)

    def add(self, key, row, entry, device):
        if self.n >= self.CAP:
            return False
        if self.host is None:
            self.host = torch.zeros(self.CAP, self.words, dtype=torch.int64)
            self.dev = torch.zeros(self.CAP, self.words, dtype=torch.int64, device=device)
        self.host[self.n] = torch.tensor(row, dtype=torch.int64)
        self.rows[key], self.entries[key] = self.n, entry
        self.n += 1
        self.dirty = True
        return True

    def drop(self, key):
        self.host[self.rows.pop(key)].zero_()
        del self.entries[key]
        self.dirty = True

    def sweep(self, is_dead):
        """Neutralise the rows of dead entries (called before every refresh launch outside a capture)."""
        for k in [k for k, e in self.entries.items() if is_dead(k, e)]:
            self.drop(k)

    def upload(self):
        if self.dirty and self.dev is not None:
            self.dev.copy_(self.host, non_blocking=False)
            self.dirty = False

    def clear(self):
        """Forget every entry (test teardown); the device tensor stays (an old graph may still hold its address)."""
        for k in list(self.entries):
            self.drop(k)
        self.upload()
        self.event = self.stream = None


_ROW = _PackTable(6)        # rows: { W, packed, R, Cc, row stride, transposed }; key = (address, R, Cc, row stride, transposed)


def _pack_done(tab):
    """The refresh launch sits on the current stream: remember it, consumers on OTHER streams (the right-to-left decoder, the CTC
    head) order themselves behind it."""
    ev = torch.cuda.Event()
    ev.record()
    tab.event, tab.stream = ev, torch.cuda.current_stream()


def _pack_wait(tab):
    if tab.event is not None and tab.stream != torch.cuda.current_stream():
        torch.cuda.current_stream().wait_event(tab.event)


def row_packs_clear():
    _ROW.clear()


def _row_pieces(k):
    return (k[1] // 32) * (k[2] // 16) if not k[4] else (k[2] // 32) * (k[1] // 16)


def _row_packed(w, transposed):
    """Packed fragments of Wg = w (x W^T) or w^T (dy W), fresh for the weight's current values; None if w cannot be registered."""
    owner = w._base if w._base is not None else w
    # only weights that OWN their memory for good: a Parameter (or a view of one), or a window of the parameter arena.  A temporary -
    # q / k / v weights concatenated for the fused projection of a model without an arena - dies after the call and the next
    # temporary lands on its address with other values and the same version: its cached pack would be taken for fresh
    a = _arena.active()
    if not (isinstance(owner, torch.nn.Parameter) or (a is not None and owner is a.flat)):
        return None
    key = (w.data_ptr(), w.shape[0], w.shape[1], w.stride(0), int(transposed))
    ent = _ROW.entries.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is not None and ent["owner"]() is not owner:       # the address has a new tenant: the old row dies, a new one is appended
        if capturing:
            return None
        _ROW.drop(key)
        ent = None
    if ent is None:
        if capturing:
            return None                                       # (its buffer would live in the graph's pool: stay on oe_gemm_f32)
        ent = dict(owner=weakref.ref(owner), buf=torch.empty(w.shape[0] * w.shape[1] * 6, dtype=torch.uint8, device=w.device), gen=-1, ver=None)
        if not _ROW.add(key, [key[0], ent["buf"].data_ptr(), key[1], key[2], key[3], key[4]], ent, w.device):
            return None
    gen, ver = _planes.weights_generation(), owner._version
    if ent["gen"] != gen or ent["ver"] != ver:
        if not capturing:
            _ROW.sweep(lambda k, e: e["owner"]() is None)
            _ROW.upload()
        elif _ROW.dirty:
            return None
        # every registered matrix in one launch (all of them are stale together: the generation moved)
        hip.call("oe_rowgemm6_pack_table", _ROW.dev, _ROW.n, max(_row_pieces(k) for k in _ROW.entries))
        _pack_done(_ROW)
        for e in _ROW.entries.values():
            o = e["owner"]()
            e["gen"], e["ver"] = gen, (None if o is None else o._version)
    else:
        _pack_wait(_ROW)
    return ent["buf"]


def _rowgemm_try(x, w, transposed, bias, out, epi, ln_pending=None, lnf_pending=None, ln_epi=None):
    """x (M, k) @ Wg^T on the row-block / tile kernels if the problem and its epilogue qualify; returns out or None.
    ln_pending: x is the `g` of a parked LayerNorm backward (_PENDING_LN) - launched as the kernel's prologue when this is the
    k = 256 row-block kernel, and on its own BEFORE anything reads x in every other case (also when None is returned)."""
    try:
        return _rowgemm_try_impl(x, w, transposed, bias, out, epi, ln_pending, lnf_pending, ln_epi)
    finally:
        if ln_pending is not None:
            _resolve_ln(ln_pending)              # (no-op when the fused launch has happened)
        if lnf_pending is not None:
            _resolve_lnf(lnf_pending)


def _rowgemm_try_impl(x, w, transposed, bias, out, epi, ln_pending, lnf_pending=None, ln_epi=None):
    if not ROWGEMM or hip.GEMM_PRECISION != 6 or not x.is_cuda:
        return None
    k = x.shape[1]
    n = w.shape[1] if transposed else w.shape[0]
    if (w.shape[0] if transposed else w.shape[1]) != k:
        return None
    form = hip.lib().oe_rowgemm6_form(x.shape[0], k, n)      # 2: tile form (few rows), 1: row-block form
    if form == 0 or (form == 1 and x.shape[0] < ROWGEMM_MIN_ROWS) or (form == 2 and not ROWTILE):
        return None
    ok = _ROW_EPI_OK if form == 1 else _ROWTILE_EPI_OK
    for key, v in epi.items():                                 # an epilogue feature the kernel does not have: oe_gemm_f32
        unset = v is None or (not isinstance(v, torch.Tensor) and (v is False or v == 0 or (key == "beta" and v == 1.0)))
        if not unset and key not in ok:
            return None
    act = epi.get("act", 0) or 0
    pre_out, aux_in, ld_aux = epi.get("preact_out"), epi.get("actgrad_in"), epi.get("ld_aux", 0) or 0
    if act not in (0, ACT_RELU, ACT_SWISH) or (pre_out is not None and aux_in is not None):
        return None
    for t in (pre_out, aux_in):
        if t is not None and (t.dtype != torch.float32 or t.data_ptr() % 8 or ld_aux % 2):
            return None
    if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or w.stride(1) != 1 or w.stride(0) % 4 or w.data_ptr() % 16:
        return None
    res = epi.get("residual")
    if res is not None and (res.stride(-1) != 1 or res.data_ptr() % 16 or epi.get("ldr", 0) % 4):
        return None
    if bias is not None and bias.data_ptr() % 16:
        return None
    wp = _row_packed(w, transposed)
    if wp is None:
        return None
    M = x.shape[0]
    if out is None:
        out = _new(M, n, like=x)
    elif out.stride(1) != 1 or out.stride(0) % 4 or out.data_ptr() % 16:
        return None
    global ROWGEMM_LAUNCHES, LN_BWD_FUSED_LAUNCHES
    ROWGEMM_LAUNCHES += 1
    ln = None
    if ln_pending is not None and not ln_pending.get("done"):
        pd = ln_pending
        sd = epi.get("seed_dev")
        if (form == 1 and k == 256 and x.is_contiguous() and x.data_ptr() == pd["g"].data_ptr() and M == pd["rows"] and
                (sd is None or _seed_dev is None or sd.data_ptr() == _seed_dev.data_ptr()) and (epi.get("drop_p", 0.0) or 0.0) == 0.0):
            ln = _ln_dict(pd)
        else:
            _resolve_ln(pd)                     # this launch reads x: the LayerNorm backward first, on its own
    lnf = None
    if lnf_pending is not None and not lnf_pending.get("done"):
        q = lnf_pending
        if (form == 1 and k == 256 and ln is None and x.is_contiguous() and x.data_ptr() == q["y"].data_ptr() and M == q["rows"] and
                q.get("gamma2") is None):
            lnf = dict(x=q["x"], gamma=q["gamma"], beta=q["beta"], eps=q["eps"], y=q["y"], stats=q["stats"], rowmask=q["rowmask"])
        else:
            _resolve_lnf(q)                     # this launch reads x: the LayerNorm first, on its own
    lne = None
    if (ln_epi is not None and form == 1 and k == 256 and n == 256 and lnf is None and bias is None and not epi and
            M == ln_epi["dx"].shape[0] and ln_epi["dx"].is_contiguous() and ln_epi["x"].is_contiguous()):
        lne = ln_epi                            # the product leaves through a LayerNorm backward: `out` is not written, ln_epi["dx"] is
    hip.rowgemm6(x, wp, out, M, k, n, bias=bias, drop_p=epi.get("drop_p", 0.0) or 0.0, seed=epi.get("seed", 0) or 0,
                 seed_dev=(_seed_dev if ln is not None else epi.get("seed_dev")),
                 rowmask=epi.get("rowmask"), residual=res, ldr=epi.get("ldr", 0) or 0, beta=epi.get("beta", 1.0),
                 act=act, preact_out=pre_out, actgrad_in=aux_in, ld_aux=ld_aux, ln=ln, lnf=lnf, lne=lne)
    if lne is not None:
        ln_epi["done"] = True
    if ln is not None:
        ln_pending["done"] = True
        LN_BWD_FUSED_LAUNCHES += 1
        _ln_reduce(ln_pending)
    if lnf is not None:
        global LN_FWD_FUSED_LAUNCHES
        lnf_pending["done"] = True
        LN_FWD_FUSED_LAUNCHES += 1
    return out


def gemm_nt(x, w, bias=None, out=None, out_planes=False, **epi):
    """y[M,N] = x[M,K] @ w[N,K]^T (+ epilogue).  out_planes: the output is a later GEMM's operand - write its bf16 planes too."""
    M, K = x.shape
    N = w.shape[0]
    lnf = _lnf_take(x)                                                            # x may be the output of a parked LayerNorm forward
    if not (out_planes and _planes.split_activations()):
        y = _rowgemm_try(x, w, False, bias, out, epi, lnf_pending=lnf)             # (launches or resolves it on every path)
        if y is not None:
            return y
    elif lnf is not None:
        _resolve_lnf(lnf)
    if out is None:
        out = _new(M, N, like=x)
    ap, bp = _operand_planes(x, w)
    cp = _planes.new_output(out) if (out_planes and _planes.split_activations() and out.numel() >= _planes.MIN_SPLIT_ELEMS) else None
    hip.gemm(x, w, out, M, N, K, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=bias, a_planes=ap, b_planes=bp, c_planes=cp, **epi)
    return out


def gemm_nn(dy, w, out=None, out_planes=False, ln_epi=None, **epi):
    """dx[M,K] = dy[M,N] @ w[N,K].  out_planes: True = with the general pre-split policy, "always" = whenever pre-split
    operands exist at all (the conv front end's policy)."""
    M, N = dy.shape
    K = w.shape[1]
    pend = _PENDING_LN.pop(dy.data_ptr(), None) if _PENDING_LN else None         # dy is the g of a parked LayerNorm backward
    if not (out_planes and (_planes.available() if out_planes == "always" else _planes.split_activations())):
        y = _rowgemm_try(dy, w, True, None, out, epi, ln_pending=pend, ln_epi=ln_epi)      # (launches or resolves it on every path)
        if y is not None:
            return y
    elif pend is not None:
        _resolve_ln(pend)
    if out is None:
        out = _new(M, K, like=dy)
    ap, bp = _operand_planes(dy, w)
    want = _planes.available() if out_planes == "always" else (out_planes and _planes.split_activations())
    cp = _planes.new_output(out) if (want and out.numel() >= _planes.MIN_SPLIT_ELEMS) else None
    hip.gemm(dy, w, out, M, K, N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), b_kmajor=True, a_planes=ap, b_planes=bp, c_planes=cp, **epi)
    return out


def gemm_nn_deep(dy, w, alpha_dev=None):
    """dx[M,K] = dy[M,N] @ w[N,K] for a deep reduction into a narrow output (the activation gradient through a vocabulary
    projection: N = 3246, K = 256).  One 64x64 tile per block walks the whole reduction: with few tiles the chip is mostly
    idle and the launch is bound by the latency of ~100 dependent K-tiles.  Split-K with atomic accumulation into a zeroed
    output puts ~2000 blocks in flight (tools/deepk_scan.py: 992 rows 138 -> 27.5 us at 8 splits, 7936 rows 163 -> 94 us
    at 4; odd split counts measured slower)."""
    M, N = dy.shape
    K = w.shape[1]
    tiles = -(-M // 64) * -(-K // 64)
    sk = 1
    if hip.GEMM_PRECISION != 0 and N >= 2048 and K <= 512:
        want = -(-1984 // tiles)
        sk = 8 if want >= 8 else 4 if want >= 3 else 2
    if sk == 1:
        return gemm_nn(dy, w, alpha_dev=alpha_dev)
    out = _new(M, K, like=dy, zero=True)
    hip.gemm(dy, w, out, M, K, N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), b_kmajor=True, split_k=sk,
             atomic_out=True, alpha_dev=alpha_dev)
    return out


def wgrad_planes(dy, x, always=False):
    """Pre-split copies of both operands of a weight gradient dy^T x (both activations), or (None, None).
    always: under the conv front end's policy too (not only the general one)."""
    if not (_planes.available() if always else _planes.active()):
        return None, None
    make = always or _planes.split_activations()
    ap = _planes.of(dy, make=make)
    bp = _planes.of(x, make=make) if ap is not None else None
    return (ap, bp) if bp is not None else (None, None)


def gemm_tn(dy, x, out=None, alpha=1.0, alpha_dev=None, bias_out=None, planes=None):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]  (split-K, atomic accumulation into `out`).
    bias_out (optional, [N], accumulated): alpha * column sums of dy - fused into the GEMM on the
    bf16 paths, a separate column-sum launch on the exact-fp32 path.
    planes: (dy planes, x planes) resolved by the caller (deferred launches resolve them when the gradient is requested)."""
    M, N = dy.shape
    K = x.shape[1]
    if out is None:
        out = _new(N, K, like=dy, zero=True)
    fused = bias_out is not None and hip.GEMM_PRECISION != 0
    ap, bp = planes if planes is not None else wgrad_planes(dy, x)
    hip.gemm(dy, x, out, N, K, M, lda=dy.stride(0), ldb=x.stride(0), ldc=out.stride(0), a_kmajor=True, b_kmajor=True,
             split_k=_split_k(N, K, M), atomic_out=True, alpha=alpha, alpha_dev=alpha_dev,
             a_colsum=bias_out if fused else None, a_planes=ap, b_planes=bp)
    if bias_out is not None and not fused:
        colsum(dy, alpha, alpha_dev, N, out=bias_out)
    return out


def colsum(x, alpha=1.0, alpha_dev=None, n=None, out=None):
    M = x.shape[0]
    n = x.shape[1] if n is None else n
    acc = out is not None
    if out is None:
        out = _new(n, like=x)
    hip.call("oe_colsum_f32", x, x.stride(0), M, n, alpha, alpha_dev, out, int(acc))
    return out


# ---- weight-gradient GEMMs on a side stream -----------------------------------
# Nothing downstream of backward consumes a parameter gradient before the optimizer, so (with a
# ParamArena as the destination) the wgrad GEMMs can run on a second HIP stream beside the
# activation-gradient chain and fill its bubbles.  The engine joins the streams after backward.
ASYNC_WGRAD = False
_side_stream = None


def side_stream():
    global _side_stream
    if _side_stream is None:
        _side_stream = torch.cuda.Stream()
    return _side_stream


# ---- phase stamps (diagnostic, tools/phase_stamps.py): STAMPS = {"buf": int64 device tensor, "tags": [...]} or None ----------
STAMPS = None


def stamp(tag: str):
    """Record the wall clock when the current stream gets here (no-op unless a tool switched STAMPS on)."""
    if STAMPS is None:
        return
    slot = len(STAMPS["tags"])
    if slot >= STAMPS["buf"].numel():
        return
    STAMPS["tags"].append(tag)
    hip.call("oe_stamp", STAMPS["buf"], slot)


def stamp_grad(x: torch.Tensor, tag: str) -> None:
    """Stamp when backward has produced the gradient of x."""
    if STAMPS is not None and x.requires_grad:
        x.register_hook(lambda g, tag=tag: stamp(tag))


# ---- cut points for a step captured in segments (TrainEngine.capture with several ranks) ---------------------------------
# At a cut the forward value passes through unchanged, but the autograd tape is severed: the engine first differentiates
# from the loss down to the last cut (its gradient lands in `leaf.grad`), then resumes from each cut's upstream side
# with that gradient - one HIP graph per piece, a gradient all-reduce issued between the replays.
CUTS = None          # None: cut() is the identity.  A list: (name, upstream tensor, leaf) per cut, in forward order.


def cut(x: torch.Tensor, name: str) -> torch.Tensor:
    if CUTS is None or not x.requires_grad:
        return x
    leaf = x.detach().requires_grad_(True)
    CUTS.append((name, x, leaf))
    return leaf


def forked_streams():
    """Every stream the step may fork work onto (the engine checks that a capture has led them all back)."""
    return [s for s in (_side_stream, _decoder_stream, _ctc_stream) if s is not None] + list(_extra_streams)


_extra_streams = []          # streams registered by callers that fork on their own (tests; experiments)


def join_side_stream():
    flush_wgrads()
    if _side_stream is not None:
        torch.cuda.current_stream().wait_stream(_side_stream)
    if _decoder_stream is not None:
        torch.cuda.current_stream().wait_stream(_decoder_stream)
    if _ctc_stream is not None:
        torch.cuda.current_stream().wait_stream(_ctc_stream)


# ---- the right-to-left decoder beside the left-to-right one -----------------------
# The two decoders of the bi-decoder (decoder.py:278-309) share nothing but their inputs, and their launches are small
# (992 rows: 16 x 4 tiles of a GEMM, i.e. 64 blocks on 256 CUs, each bound by its own latency).  With this switch on,
# ASRModel._calc_att_loss runs the right decoder and its loss head on a second stream: ONE fork after the encoder, ONE
# join where the two losses are combined; autograd replays each node on the stream of its forward, so backward forks
# the same way.  (The per-layer weight-gradient fork above is a different matter: hundreds of cross-queue waits.)
PARALLEL_DECODERS = False
_decoder_stream = None


_ctc_stream = None


def ctc_stream():
    global _ctc_stream
    if _ctc_stream is None:
        _ctc_stream = torch.cuda.Stream()
    return _ctc_stream


def decoder_stream():
    global _decoder_stream
    if _decoder_stream is None:
        _decoder_stream = torch.cuda.Stream()
    return _decoder_stream


class _Side:
    """Run the enclosed launches on the side stream, ordered after everything enqueued so far."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if isinstance(t, torch.Tensor)]

    def __enter__(self):
        side = side_stream()
        side.wait_stream(torch.cuda.current_stream())
        self.cm = torch.cuda.stream(side)
        self.cm.__enter__()

    def __exit__(self, *exc):
        self.cm.__exit__(*exc)
        for t in self.tensors:                 # keep the allocator from recycling inputs the side stream still reads
            t.record_stream(_side_stream)
        return False


class _Inline:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def _wgrad_ctx(to_arena: bool, *tensors):
    return _Side(*tensors) if (ASYNC_WGRAD and to_arena) else _Inline()


# Deferred form for captured graphs: a cross-queue edge costs ~6 us at replay, so a fork per weight gradient (150 of them)
# loses more than the overlap gives (21.5 vs 19.6 ms/step).  With WGRAD_DEFER = n > 0 the launches are collected instead
# (the closures keep their operands alive) and every n of them go to the side stream behind ONE fork; the engine flushes
# the rest and joins after backward.  Operands are never modified in place after a weight gradient has been requested
# (the eager side stream relies on the same property).
WGRAD_DEFER = 0
_deferred = []


def _run_wgrad(to_arena: bool, tensors, fn, desc=None):
    """desc (optional): the launch as data - dict(dy, x, out, alpha, alpha_dev, bias_out) of a plain `out += alpha dy^T x` -
    so that a flush can run the small ones of a group as ONE grouped launch (WGRAD_GROUPED)."""
    if WGRAD_DEFER > 0 and to_arena:
        # the stream that produces this launch's operands: the flush must order the side stream after every one of them
        # (the decoders' backward runs on other streams than the encoder's)
        _deferred.append((fn, [t for t in tensors if isinstance(t, torch.Tensor)], torch.cuda.current_stream(), desc))
        if len(_deferred) >= WGRAD_DEFER:
            flush_wgrads()
        return
    with _wgrad_ctx(to_arena, *tensors):
        fn()


# The weight gradients of a flush as ONE launch of the bf16-planes kernel (oe_gemm_tn_grouped).  Alone each output of a few
# 128 x 128 tiles over K = B T' rows needs a 16-60-way split of the reduction to cover the chip and spends much of its time
# in the atomic epilogue (the four small ones of an encoder layer: 81 us one by one, 41 us together).  Measured in the step
# (one box, repeated): everything up to 12 tiles grouped 15.34 ms/step, up to 16 / 24 / 48 tiles 15.40 / 15.47 / 15.34-15.49,
# nothing grouped 15.86 - the feed-forward's 16-tile outputs fill the chip on their own and the 52-tile vocabulary
# projections (other reduction lengths) unbalance a group.
WGRAD_GROUPED = os.environ.get("OE_WGRAD_GROUPED", "1") == "1"
WGRAD_GROUP_MAX_TILES = int(os.environ.get("OE_WGRAD_GROUP_MAX_TILES", "12"))     # outputs of more tiles than this are launched on their own
WGRAD_GROUP_BLOCKS = int(os.environ.get("OE_WGRAD_GROUP_BLOCKS", "512"))           # blocks a grouped launch aims for (sets the split of the reductions)
_TN_TABLES = []                    # eager flushes: (device table, host bytes) kept until the stream has surely consumed them


def _group_wgrads(descs):
    """Launch the groupable ones among `descs` (current stream = the side stream); returns the set of indices it covered."""
    if hip.GEMM_PRECISION == 0:
        return set()
    idx = []
    for i, d in enumerate(descs):
        if d is None:
            continue
        dy, x, out = d["dy"], d["x"], d["out"]
        tiles = -(-dy.shape[1] // 128) * -(-x.shape[1] // 128)
        ok = (dy.dim() == 2 and x.dim() == 2 and out.dim() == 2 and dy.stride(1) == 1 and x.stride(1) == 1 and out.stride(1) == 1 and
              dy.shape[0] == x.shape[0] and dy.shape[0] >= 128 and tiles <= WGRAD_GROUP_MAX_TILES and dy.shape[1] >= 4 and x.shape[1] >= 4 and
              dy.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and
              dy.stride(0) >= -(-dy.shape[1] // 4) * 4 and x.stride(0) >= -(-x.shape[1] // 4) * 4)
        if ok:
            idx.append(i)
    if len(idx) < 2:
        return set()
    plan = hip.tn_grouped_plan([descs[i] for i in idx], WGRAD_GROUP_BLOCKS)
    if plan is None:
        return set()
    host, total = plan
    table = torch.empty(len(host), dtype=torch.uint8, device=descs[idx[0]]["dy"].device)
    if LN_TABLE is not None and not LN_TABLE.get("eager"):
        LN_TABLE.setdefault("uploads", []).append((table, host))       # inside a capture: filled by ln_table_end, kept with the graph
    else:
        # stream-ordered upload from pinned memory: the host never waits for the stream (bench.py's bracketed steps park the GPU
        # while the host enqueues the whole step)
        pinned = torch.frombuffer(bytearray(host), dtype=torch.uint8).pin_memory()
        table.copy_(pinned, non_blocking=True)
        _TN_TABLES.append((table, pinned))
        del _TN_TABLES[:-64]
    hip.tn_grouped_launch(table, len(idx), total,
                          flops=sum(2.0 * descs[i]["dy"].shape[0] * descs[i]["dy"].shape[1] * descs[i]["x"].shape[1] for i in idx))
    return set(idx)


def drop_deferred():
    """Forget deferred launches without running them (after a failed capture)."""
    global _deferred
    _deferred = []


# Measurement aid (bench.py's event-bracketed steps, `--single-stream` profile runs): run a flush on the CURRENT stream, so that
# an eager step launches exactly the kernels a captured step holds - the grouped weight gradients included - one after the other.
WGRAD_FLUSH_INLINE = False


def flush_wgrads():
    global _deferred
    if not _deferred:
        return
    if WGRAD_FLUSH_INLINE:
        grouped = _group_wgrads([d for _, _, _, d in _deferred]) if WGRAD_GROUPED else set()
        for i, (fn, _, _, _) in enumerate(_deferred):
            if i not in grouped:
                fn()
        _deferred = []
        return
    side = side_stream()
    origins = []
    for _, _, st, _ in _deferred:
        if all(st != o for o in origins):
            origins.append(st)
    for st in origins:
        side.wait_stream(st)
    with torch.cuda.stream(side):
        grouped = _group_wgrads([d for _, _, _, d in _deferred]) if WGRAD_GROUPED else set()
        for i, (fn, _, _, _) in enumerate(_deferred):
            if i not in grouped:
                fn()
    for _, ts, _, _ in _deferred:
        for t in ts:
            t.record_stream(side)
    _deferred = []


# ---- parameter-gradient sinks ------------------------------------------------
# With a ParamArena active the kernels accumulate into the flat gradient buffer and
# autograd is told "no gradient" (None); otherwise a fresh tensor is returned.
def wgrad(param, dy, x, alpha=1.0, alpha_dev=None):
    tgt = _arena.grad_target(param)
    if tgt is None:
        return gemm_tn(dy, x, alpha=alpha, alpha_dev=alpha_dev).view(param.shape)
    out = tgt.view(dy.shape[1], x.shape[1])
    pl = wgrad_planes(dy, x)
    _run_wgrad(True, (dy, x, alpha_dev) + _pl_tensors(pl), lambda: gemm_tn(dy, x, out=out, alpha=alpha, alpha_dev=alpha_dev, planes=pl),
               desc=dict(dy=dy, x=x, out=out, alpha=alpha, alpha_dev=alpha_dev, bias_out=None))
    return None


def _pl_tensors(pl):
    """The bf16 tensors behind a (dy planes, x planes) pair, for stream bookkeeping."""
    return tuple(q.t for q in pl if q is not None) if pl is not None else ()


def _sink_swapped(param, src, A, Bd, Cd):
    """param-grad = swap_last2(src viewed [A][Bd][Cd]) accumulated into the arena or returned."""
    tgt = _arena.grad_target(param)
    if tgt is None:
        out = torch.empty_like(param)
        hip.call("oe_swap_last2", src, A, Bd, Cd, out, 0)
        return out
    hip.call("oe_swap_last2", src, A, Bd, Cd, tgt, 1)
    return None


def wgrad_bias(w, b, dy, x, alpha=1.0, alpha_dev=None):
    """Weight and bias gradient of y = x w^T + b from dy in ONE GEMM launch (bias = fused column sums).
    Returns what autograd should see: (dw, db), each None when it went straight into the arena."""
    if b is None:
        return wgrad(w, dy, x, alpha, alpha_dev), None
    N, K = dy.shape[1], x.shape[1]
    tw, tb = _arena.grad_target(w), _arena.grad_target(b)
    ow = tw.view(N, K) if tw is not None else _new(N, K, like=dy, zero=True)
    ob = tb if tb is not None else _new(N, like=dy, zero=True)
    pl = wgrad_planes(dy, x)
    _run_wgrad(tw is not None and tb is not None, (dy, x, alpha_dev) + _pl_tensors(pl),
               lambda: gemm_tn(dy, x, out=ow, alpha=alpha, alpha_dev=alpha_dev, bias_out=ob, planes=pl),
               desc=dict(dy=dy, x=x, out=ow, alpha=alpha, alpha_dev=alpha_dev, bias_out=ob))
    return (None if tw is not None else ow.view(w.shape)), (None if tb is not None else ob)


def bgrad(param, dy, alpha=1.0, alpha_dev=None, n=None):
    if param is None:
        return None
    tgt = _arena.grad_target(param)
    if tgt is None:
        return colsum(dy, alpha, alpha_dev, n)
    colsum(dy, alpha, alpha_dev, n, out=tgt)
    return None


def grad_sink(param):
    """(buffer the kernel accumulates into, value to hand back to autograd)."""
    tgt = _arena.grad_target(param)
    if tgt is None:
        z = torch.zeros_like(param)
        return z, z
    return tgt.view(param.shape), None


def _fused_rows(parts, cat_dim0=True):
    """Weights that sit back to back in the arena are used in place; otherwise concatenated."""
    a = _arena.active()
    if a is not None and a.enabled and a.adjacent(*parts):
        rows = sum(p.shape[0] for p in parts)
        return torch.as_strided(parts[0], (rows,) + tuple(parts[0].shape[1:]), parts[0].stride()), True
    return torch.cat(list(parts), 0), False


def dropout_scale(x, alpha=1.0, p=0.0, seed=0, rowmask=None, cols=None):
    out = torch.empty_like(x)
    cols = x.shape[-1] if cols is None else cols
    hip.call("oe_dropout_scale", x, x.numel(), cols, alpha, p, seed, _seed_dev, rowmask, out)
    return out


# --------------------------------------------------------------------------- #
# LayerNorm
# --------------------------------------------------------------------------- #
def _ln_ws(like, rows, d):
    return _new(hip.lib().oe_layernorm_bwd_workspace_floats(rows, d), like=like)


def _ln_fwd(x, gamma, beta, eps, rows, d, rowmask, act, y, stats, planes_out=True):
    """LayerNorm forward; in the pre-split mode the output also leaves as bf16 planes (it feeds a GEMM)."""
    pl = _planes.new_output(y.view(rows, d)) if (planes_out and _planes.active() and rows * d >= _planes.MIN_SPLIT_ELEMS) else None
    if pl is None:
        hip.call("oe_layernorm_fwd", x, gamma, beta, eps, rows, d, rowmask, act, y, stats)
    else:
        hip.call("oe_layernorm_fwd_pl", x, gamma, beta, eps, rows, d, rowmask, act, y, stats, pl.t, pl.stride)


# While a training step is being captured into a HIP graph, LayerNorm backward leaves the reduction of its per-block
# parameter-gradient partials to ONE table-driven launch at the end of backward (ln_table_flush) instead of one small
# launch per call.  The table is a device tensor allocated before the capture; the host fills it after the capture (the
# workspaces' addresses inside the graph's memory pool are the same at every replay).
LN_TABLE = None
LN_TABLE_CAPACITY = 512


LN_EAGER_TABLE = os.environ.get("OE_LN_EAGER_TABLE", "1") == "1"
_LN_EAGER_DEV = {}          # device -> the eager table's device tensor (allocated once)
_LN_EAGER_PINNED = []       # pinned host tables of the last flushes (alive until their stream-ordered uploads have surely run)


def ln_table_begin(device, eager=False):
    """eager: table mode for an eager step too (TrainEngine) - the table is uploaded stream-ordered right before the launch that reads it
    (a captured step fills it once after the capture, ln_table_end)."""
    global LN_TABLE
    if eager:
        dev = _LN_EAGER_DEV.get(device)
        if dev is None:
            dev = _LN_EAGER_DEV[device] = torch.zeros(LN_TABLE_CAPACITY, 5, dtype=torch.int64, device=device)
    else:
        dev = torch.zeros(LN_TABLE_CAPACITY, 5, dtype=torch.int64, device=device)
    LN_TABLE = {"dev": dev, "entries": [], "keep": [], "max_rows": 1, "max_d": 4, "launched": 0, "eager": eager}
    return LN_TABLE


def ln_table_flush():
    """Enqueue the reduction of everything recorded since the last flush (current stream; call after the side streams
    have been joined)."""
    t = LN_TABLE
    if t is None or len(t["entries"]) == t["launched"]:
        return
    first, n = t["launched"], len(t["entries"]) - t["launched"]
    assert len(t["entries"]) <= LN_TABLE_CAPACITY, "LN_TABLE_CAPACITY exceeded"
    if t.get("eager"):
        host = torch.tensor(t["entries"][first:], dtype=torch.int64).pin_memory()
        t["dev"][first:first + n].copy_(host, non_blocking=True)
        _LN_EAGER_PINNED.append(host)
        del _LN_EAGER_PINNED[:-8]
        cur = torch.cuda.current_stream()
        for ws in t["keep"]:
            ws.record_stream(cur)               # (workspaces of norms that ran on the decoders' streams)
    hip.call("oe_layernorm_param_reduce_table", t["dev"][first:], n, t["max_rows"], t["max_d"])
    t["launched"] = len(t["entries"])
    t["keep"] = []                      # later allocations may reuse the workspaces: they come after this launch in stream order


def ln_table_end():
    """After the capture: upload the table (host -> device copy, not capturable) and leave table mode."""
    global LN_TABLE
    t, LN_TABLE = LN_TABLE, None
    if t is not None and t["entries"]:
        t["dev"][: len(t["entries"])].copy_(torch.tensor(t["entries"], dtype=torch.int64))
    if t is not None:
        for table, host in t.get("uploads", ()):        # grouped weight-gradient tables recorded during the capture
            table.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
    return t


# A block's output is `residual + out_scale * dropout(f(LN(x)))`, and the LayerNorm that consumes it belongs to the NEXT block:
# in backward that LayerNorm's dx is exactly the gradient the block starts from, which it first multiplies by its output
# dropout mask and scale.  The block tags its output (FUSE_OUT_DROP); the next LayerNorm backward then writes that product
# as a second output of its kernel (oe_layernorm_bwd_dx_drop) and the block's backward picks it up by the address of the
# gradient it receives - one elementwise launch per block (66 per step at config 2) less.  Anything in between (an adapter's
# add, a gradient autograd had to accumulate) changes the address and the block falls back to oe_dropout_scale.
# Entries live from the LayerNorm backward that makes them to the block backward that consumes them; whatever is left
# (a block that was never differentiated) is dropped when the next forward pass starts (ASRModel.forward /
# LanguageModel.forward / TrainEngine call predrop_clear).
FUSE_OUT_DROP = os.environ.get("OE_FUSE_OUT_DROP", "1") == "1"
_PREDROP = {}


def predrop_clear():
    """Called where a new step starts (ASRModel.forward / LanguageModel.forward / TrainEngine): drops the previous step's
    leftovers - dropped-gradient copies and the registry of pre-split GEMM operands."""
    if _PENDING_LN:
        resolve_pending_ln()
    if _PENDING_LNF:
        resolve_pending_lnf()
    _PREDROP.clear()
    _planes.clear()


def _tag_out_drop(out, out_scale, p_out, s_out, rowmask=None, ln_fuse=False):
    if FUSE_OUT_DROP and (p_out > 0 or out_scale != 1.0 or rowmask is not None):
        out._oe_outdrop = (float(out_scale), float(p_out), int(s_out), rowmask)
        if ln_fuse:
            out._oe_lnfuse = True          # this block's backward can take the next LayerNorm's backward into its first GEMM (_ln_bwd)
    return out


# The LayerNorm backward in front of a block's backward as the PROLOGUE of that block's first input-gradient GEMM (oe_rowgemm6's
# ln_* arguments): the block (attention: linear_out; conv module: pointwise_conv2) says at forward time that its backward starts with
# `g = _out_drop_grad(dy)` followed by `gemm_nn(g, w)` on the row-block kernel (_ln_fuse_ok); the pre-norm fork that consumes the
# block's output then does NOT launch its backward kernel but parks its arguments here, keyed by the address of g; gemm_nn finds
# them and launches the fused kernel, which writes dx and g before anything else can read them.  Every other path out of
# _out_drop_grad / gemm_nn resolves a parked entry by launching the LayerNorm backward on its own (_resolve_ln), and TrainEngine
# checks that none is left behind.  24 launches per step at config 2 (10.8 us + a dependent-launch gap each).
LN_BWD_FUSE = os.environ.get("OE_LN_BWD_FUSE", "1") == "1"
LN_BWD_FUSED_LAUNCHES = 0       # (tests)
LN_EPI_FUSE = os.environ.get("OE_LN_EPI_FUSE", "1") == "1"      # the conv module's norm backward as an epilogue (ConvModuleFn.backward)
LN_EPI_FUSED_LAUNCHES = 0
_PENDING_LN = {}


def _ln_fuse_ok(rows, d, w):
    """Forward-time promise of a block whose backward starts with gemm_nn(g, w), w (d, d): will that be the k = 256 row-block kernel?"""
    return (LN_BWD_FUSE and ROWGEMM and FUSE_OUT_DROP and hip.GEMM_PRECISION == 6 and d == 256 and rows >= ROWGEMM_MIN_ROWS and
            w.shape[0] == 256 and w.shape[1] == 256 and not _planes.active())


def _ln_reduce(pend):
    if pend["table"]:
        return                                   # (a captured step: the entries are in LN_TABLE, one launch at the end of backward)
    hip.call("oe_layernorm_param_reduce", pend["ws"], pend["rows"], pend["d"], pend["dg"], pend["db"])
    if pend.get("gamma2") is not None:
        hip.call("oe_layernorm_param_reduce", pend["ws2"], pend["rows"], pend["d"], pend["dg2"], pend["db2"])


def _ln_dict(pend):
    """A parked LayerNorm backward as the `ln` argument of hip.rowgemm6 / hip.ffn_bwd (LnPrologue.fill)."""
    alpha, p, seed, gmask = pend["spec"]
    return dict(dy=pend["dy"], x=pend["x"], stats=pend["stats"], gamma=pend["gamma"], add=pend["add"], dx=pend["dx"], g=pend["g"], ws=pend["ws"],
                alpha=alpha, p=p, seed=seed, rowmask=gmask, ln_rowmask=pend.get("rowmask"), beta=pend["beta"] if pend.get("gamma2") is not None else None,
                gamma2=pend.get("gamma2"), stats2=pend.get("stats2"), ws2=pend.get("ws2"))


def _resolve_ln(pend):
    """The parked LayerNorm backward as a launch of its own (exactly what _ln_bwd / NormPairFn.backward would have launched)."""
    if pend.get("done"):
        return
    alpha, p, seed, gmask = pend["spec"]
    if pend.get("gamma2") is not None:
        hip.call("oe_layernorm_pair_bwd_dx_drop", pend["dy"], pend["x"], pend["gamma"], pend["beta"], pend["stats"], pend["gamma2"], pend["stats2"],
                 pend["rows"], pend["d"], pend["add"], pend["dx"], pend["g"], alpha, p, seed, _seed_dev, gmask, pend["ws"], pend["ws2"])
    else:
        hip.call("oe_layernorm_bwd_dx_drop", pend["dy"], pend["x"], pend["gamma"], pend["beta"], 0, pend["stats"], pend["rows"], pend["d"],
                 pend.get("rowmask"), pend["add"], pend["dx"], pend["g"], alpha, p, seed, _seed_dev, gmask, pend["ws"])
    pend["done"] = True
    _ln_reduce(pend)


def resolve_pending_ln():
    """Safety net (TrainEngine after backward, predrop_clear): a parked LayerNorm backward nobody picked up.  Returns the count."""
    n = 0
    for pend in list(_PENDING_LN.values()):
        if not pend.get("done"):
            _resolve_ln(pend)
            n += 1
    _PENDING_LN.clear()
    return n


def _out_drop_grad(dy2, out_scale, p_out, s_out, rowmask=None):
    """dropout_scale(dy2, out_scale, p_out, s_out, rowmask) - or the copy the producing LayerNorm backward already made."""
    hit = _PREDROP.pop(dy2.data_ptr(), None)
    if hit is not None:
        # the entry keeps the producing dx alive, so no other tensor can have been given this address meanwhile; an
        # in-place change of dx since (autograd summing a second consumer's gradient into it) shows in its version
        g, spec, dx, version = hit
        pend = _PENDING_LN.get(g.data_ptr()) if _PENDING_LN else None
        if dx._version != version or dx.numel() != dy2.numel():
            if pend is not None:
                _resolve_ln(_PENDING_LN.pop(g.data_ptr()))
            return dropout_scale(dy2, out_scale, p_out, s_out, rowmask)
        same_mask = (spec[3] is None and rowmask is None) or (spec[3] is not None and rowmask is not None and
                                                               spec[3].data_ptr() == rowmask.data_ptr())
        if spec[:3] == (float(out_scale), float(p_out), int(s_out)) and same_mask and g.numel() == dy2.numel():
            return g.view(dy2.shape)              # (possibly still parked: the caller's gemm_nn comes next and launches it)
        if pend is not None:
            _resolve_ln(_PENDING_LN.pop(g.data_ptr()))
    return dropout_scale(dy2, out_scale, p_out, s_out, rowmask)


def _ln_bwd(dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, dg, db, to_arena, prev_drop=None, fuse=False):
    ws = _ln_ws(x, rows, d)
    g = None
    # (to_arena: the parameter gradients are written later than this function returns - fine for arena slices, which autograd never
    # sees, wrong for fresh tensors that PreNormFn.backward hands back to autograd right away)
    if (fuse and to_arena and prev_drop is not None and LN_BWD_FUSE and FUSE_OUT_DROP and act == ACT_NONE and d == 256 and
            hip.GEMM_PRECISION == 6 and rows >= ROWGEMM_MIN_ROWS and not _planes.active() and dy.is_contiguous() and x.is_contiguous() and
            (add is None or add.is_contiguous())):
        # parked: the consuming block's first input-gradient GEMM launches it as its prologue (see _PENDING_LN)
        g = torch.empty_like(dx)
        t = LN_TABLE
        table = t is not None and to_arena
        _PENDING_LN[g.data_ptr()] = dict(dy=dy, x=x, gamma=gamma, beta=beta, stats=stats, add=add, dx=dx, g=g, ws=ws, spec=prev_drop,
                                         rows=rows, d=d, dg=dg, db=db, table=table, done=False, rowmask=rowmask)
        _PREDROP[dx.data_ptr()] = (g, prev_drop, dx, dx._version)
        if table:
            t["entries"].append((ws.data_ptr(), rows, d, dg.data_ptr(), db.data_ptr()))
            t["keep"].append(ws)
            t["max_rows"], t["max_d"] = max(t["max_rows"], rows), max(t["max_d"], d)
        return
    if prev_drop is not None and FUSE_OUT_DROP and d % 8 == 0:
        g = torch.empty_like(dx)
        alpha, p, seed, gmask = prev_drop
        # g is what the previous block's backward GEMMs read: in the pre-split mode it leaves as bf16 planes too
        pl = _planes.new_output(g.view(rows, d)) if (_planes.active() and rows * d >= _planes.MIN_SPLIT_ELEMS) else None
        if pl is None:
            hip.call("oe_layernorm_bwd_dx_drop", dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, g, alpha, p, seed, _seed_dev,
                     gmask, ws)
        else:
            hip.call("oe_layernorm_bwd_dx_drop_pl", dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, g, alpha, p, seed, _seed_dev,
                     gmask, ws, pl.t, pl.stride)
        _PREDROP[dx.data_ptr()] = (g, prev_drop, dx, dx._version)
    else:
        hip.call("oe_layernorm_bwd_dx", dy, x, gamma, beta, act, stats, rows, d, rowmask, add, dx, ws)
    t = LN_TABLE
    if t is None or not to_arena:
        hip.call("oe_layernorm_param_reduce", ws, rows, d, dg, db)
        return
    t["entries"].append((ws.data_ptr(), rows, d, dg.data_ptr(), db.data_ptr()))
    t["keep"].append(ws)
    t["max_rows"], t["max_d"] = max(t["max_rows"], rows), max(t["max_d"], d)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rowmask, act, sole_consumer=False):
        x = _chk(x, "layer_norm")
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        stats = _new(rows, 2, like=x)
        _ln_fwd(x, gamma, beta, eps, rows, d, rowmask, act, y, stats)
        ctx.save_for_backward(x, gamma, beta, stats, rowmask)
        ctx.act = act
        # only when the caller vouches that x feeds nothing else: with a second consumer autograd SUMS the gradients of x
        # (possibly in place, keeping the address), and the fused copy would hold this branch's share only
        ctx.prev_drop = getattr(x, "_oe_outdrop", None) if sole_consumer else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats, rowmask = ctx.saved_tensors
        dy = dy.contiguous()
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        (dg, rg), (db, rb) = grad_sink(gamma), grad_sink(beta)
        _ln_bwd(dy, x, gamma, beta, ctx.act, stats, rows, d, rowmask, None, dx, dg, db, rg is None and rb is None, ctx.prev_drop)
        return dx, rg, rb, None, None, None, None


def layer_norm(x, gamma, beta, eps, rowmask=None, act=ACT_NONE, sole_consumer=False):
    """sole_consumer: x is used by nothing but this norm (lets backward hand the previous block its output-dropout gradient)."""
    return LayerNormFn.apply(x, gamma, beta, eps, rowmask, act, sole_consumer)


class PreNormFn(torch.autograd.Function):
    """The fork of a pre-norm residual block, x -> (x, LN(x)) (encoder_layer.py:79-83 etc.: `residual = x;
    x = norm(x)`).  Owning both branches lets backward produce d x = d residual + LN'(d y) in the LayerNorm
    backward kernel itself (its `add` input) instead of a separate elementwise add by autograd."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, rowmask, sole_consumer=False, fuse_fwd=False):
        x = _chk(x, "pre_norm")
        d = x.shape[-1]
        rows = x.numel() // d
        y = torch.empty_like(x)
        stats = _new(rows, 2, like=x)
        if (fuse_fwd and LN_FWD_FUSE and ROWGEMM and hip.GEMM_PRECISION == 6 and d == 256 and rows >= ROWGEMM_MIN_ROWS and not _planes.active() and
                gamma.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0):
            # parked: the consumer's first kernel makes y (see _PENDING_LNF)
            _PENDING_LNF[y.data_ptr()] = dict(x=x, gamma=gamma, beta=beta, eps=eps, rows=rows, d=d, rowmask=rowmask, y=y, stats=stats, done=False)
        else:
            _ln_fwd(x, gamma, beta, eps, rows, d, rowmask, ACT_NONE, y, stats)
        ctx.save_for_backward(x, gamma, beta, stats, rowmask)
        ctx.prev_drop = getattr(x, "_oe_outdrop", None) if sole_consumer else None
        ctx.ln_fuse = bool(sole_consumer and getattr(x, "_oe_lnfuse", False))
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma, beta, stats, rowmask = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None, None
        d = x.shape[-1]
        rows = x.numel() // d
        dx = torch.empty_like(x)
        (dg, rg), (db, rb) = grad_sink(gamma), grad_sink(beta)
        add = None if dres is None else dres.contiguous()
        _ln_bwd(dy.contiguous(), x, gamma, beta, ACT_NONE, stats, rows, d, rowmask, add, dx, dg, db, rg is None and rb is None,
                ctx.prev_drop, fuse=ctx.ln_fuse)
        return dx, rg, rb, None, None, None, None


def pre_norm(x, gamma, beta, eps, rowmask=None, sole_consumer=False, fuse_fwd=False):
    """-> (residual, normed): use `residual` for the skip connection of the block.  sole_consumer: see layer_norm.
    fuse_fwd: the caller hands `normed` straight to an op of this module and to nothing else (EncoderLayer) - its first kernel may
    then make it (_PENDING_LNF)."""
    return PreNormFn.apply(x, gamma, beta, eps, rowmask, sole_consumer, fuse_fwd)


# Two LayerNorms back to back - an encoder layer's norm_final followed by the next layer's first pre-norm fork, or by the
# encoder's after_norm (encoder_layer.py:109-110 -> :79-80; encoder.py after the last layer) - as ONE launch each way:
# 12 forward and 12 backward launches and a round trip of the (rows, d) activation each less per step at config 2.
LN_PAIR = os.environ.get("OE_LN_PAIR", "1") == "1"


def ln_pair_ok(x):
    return LN_PAIR and x.is_cuda and x.dtype == torch.float32 and not _planes.active()


class NormPairFn(torch.autograd.Function):
    """x -> (u, y) with u = LN1(x), y = LN2(u); u is also an output when the caller needs it (the next block's residual).
    Backward: d x = LN1'(d u + LN2'(d y)) in one kernel, u recomputed from x."""

    @staticmethod
    def forward(ctx, x, g1, b1, eps1, g2, b2, eps2, want_first, sole_consumer, fuse_fwd=False):
        x = _chk(x, "layer_norm_pair")
        d = x.shape[-1]
        rows = x.numel() // d
        u = torch.empty_like(x) if want_first else None
        y = torch.empty_like(x)
        st1, st2 = _new(rows, 2, like=x), _new(rows, 2, like=x)
        if (fuse_fwd and want_first and LN_FWD_FUSE and FUSED_FFN and hip.GEMM_PRECISION == 6 and d == 256 and rows >= ROWGEMM_MIN_ROWS and
                not _planes.active() and all(t.data_ptr() % 16 == 0 for t in (g1, b1, g2, b2))):
            # parked: the next layer's first feed-forward makes u and y on its rows' way in (_PENDING_LNF, pair form)
            _PENDING_LNF[y.data_ptr()] = dict(x=x, gamma=g1, beta=b1, eps=eps1, gamma2=g2, beta2=b2, eps2=eps2, rows=rows, d=d, rowmask=None,
                                              u=u, y=y, stats=st1, stats2=st2, done=False)
        else:
            hip.call("oe_layernorm_pair_fwd", x, g1, b1, eps1, g2, b2, eps2, rows, d, u, st1, y, st2)
        ctx.save_for_backward(x, g1, b1, st1, g2, b2, st2)
        ctx.prev_drop = getattr(x, "_oe_outdrop", None) if sole_consumer else None
        ctx.ln_fuse = bool(sole_consumer and getattr(x, "_oe_lnfuse", False))
        ctx.want_first = want_first
        if want_first:
            return u, y
        return y

    @staticmethod
    def backward(ctx, *grads):
        du, dy = grads if ctx.want_first else (None, grads[0])
        x, g1, b1, st1, g2, b2, st2 = ctx.saved_tensors
        d = x.shape[-1]
        rows = x.numel() // d
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        add = None if du is None else du.contiguous()
        dx = torch.empty_like(x)
        (dg1, rg1), (db1, rb1) = grad_sink(g1), grad_sink(b1)
        (dg2, rg2), (db2, rb2) = grad_sink(g2), grad_sink(b2)
        ws1, ws2 = _ln_ws(x, rows, d), _ln_ws(x, rows, d)
        gout, alpha, p, seed, gmask = None, 1.0, 0.0, 0, None
        arena_all = rg1 is None and rb1 is None and rg2 is None and rb2 is None
        if (ctx.ln_fuse and arena_all and ctx.prev_drop is not None and LN_BWD_FUSE and FUSE_OUT_DROP and d == 256 and hip.GEMM_PRECISION == 6 and
                rows >= ROWGEMM_MIN_ROWS and not _planes.active() and x.is_contiguous() and (add is None or add.is_contiguous())):
            # parked: the previous block's backward (the second feed-forward's oe_ffn_bwd) launches it as its prologue (_PENDING_LN)
            gout = torch.empty_like(dx)
            t = LN_TABLE
            _PENDING_LN[gout.data_ptr()] = dict(dy=dy, x=x, gamma=g1, beta=b1, stats=st1, gamma2=g2, stats2=st2, add=add, dx=dx, g=gout, ws=ws1,
                                                ws2=ws2, spec=ctx.prev_drop, rows=rows, d=d, dg=dg1, db=db1, dg2=dg2, db2=db2, table=t is not None,
                                                done=False)
            _PREDROP[dx.data_ptr()] = (gout, ctx.prev_drop, dx, dx._version)
            if t is not None:
                for ws, dg, db in ((ws1, dg1, db1), (ws2, dg2, db2)):
                    t["entries"].append((ws.data_ptr(), rows, d, dg.data_ptr(), db.data_ptr()))
                    t["keep"].append(ws)
                t["max_rows"], t["max_d"] = max(t["max_rows"], rows), max(t["max_d"], d)
            return dx, rg1, rb1, None, rg2, rb2, None, None, None, None
        if ctx.prev_drop is not None and FUSE_OUT_DROP and d % 8 == 0:
            gout = torch.empty_like(dx)
            alpha, p, seed, gmask = ctx.prev_drop
        hip.call("oe_layernorm_pair_bwd_dx_drop", dy, x, g1, b1, st1, g2, st2, rows, d, add, dx, gout, alpha, p, seed, _seed_dev, gmask, ws1, ws2)
        if gout is not None:
            _PREDROP[dx.data_ptr()] = (gout, ctx.prev_drop, dx, dx._version)
        t = LN_TABLE
        for ws, dg, db, to_arena in ((ws1, dg1, db1, rg1 is None and rb1 is None), (ws2, dg2, db2, rg2 is None and rb2 is None)):
            if t is None or not to_arena:
                hip.call("oe_layernorm_param_reduce", ws, rows, d, dg, db)
            else:
                t["entries"].append((ws.data_ptr(), rows, d, dg.data_ptr(), db.data_ptr()))
                t["keep"].append(ws)
                t["max_rows"], t["max_d"] = max(t["max_rows"], rows), max(t["max_d"], d)
        return dx, rg1, rb1, None, rg2, rb2, None, None, None, None


def layer_norm_pair(x, g1, b1, eps1, g2, b2, eps2, want_first=True, sole_consumer=False, fuse_fwd=False):
    """(LN1(x), LN2(LN1(x))) - or only the second with want_first=False.  sole_consumer: as layer_norm's.  fuse_fwd (with
    want_first): the caller hands the pair to a feed-forward as (residual, input) and to nothing else (pre_norm's fuse_fwd)."""
    return NormPairFn.apply(x, g1, b1, eps1, g2, b2, eps2, want_first, sole_consumer, fuse_fwd)


# --------------------------------------------------------------------------- #
# Linear (single GEMM with bias / activation)
# --------------------------------------------------------------------------- #
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        x = _chk(x, "linear")
        x2 = x.reshape(-1, x.shape[-1])
        pre = _new(x2.shape[0], w.shape[0], like=x) if act != ACT_NONE else None
        if act in GEMM_FUSED_ACTS:
            y = gemm_nt(x2, w, b, act=act, preact_out=pre, ld_aux=w.shape[0])
        else:
            gemm_nt(x2, w, b, out=pre)
            y = torch.empty_like(pre)
            hip.call("oe_act_fwd", pre, pre.numel(), act, y)
        ctx.save_for_backward(x2, w, pre)
        ctx.act, ctx.has_bias, ctx.in_shape = act, b is not None, x.shape
        ctx.bias_ref = b
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, w.shape[0])
        if ctx.act != ACT_NONE:
            dy2 = _act_grad(dy2, pre, ctx.act)
        dx = gemm_nn(dy2, w).view(ctx.in_shape) if ctx.needs_input_grad[0] else None
        dw, db = wgrad_bias(w, ctx.bias_ref if ctx.has_bias else None, dy2, x2)
        return dx, dw, db, None


def _act_grad(dy2, pre, act):
    """dy * act'(pre)."""
    out = torch.empty_like(dy2)
    hip.call("oe_act_grad", dy2, pre, dy2.numel(), act, out)
    return out


def linear(x, w, b=None, act=ACT_NONE):
    return LinearFn.apply(x, w, b, act)


class ActivationFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _chk(x, "activation")
        y = torch.empty_like(x)
        hip.call("oe_act_fwd", x, x.numel(), act, y)
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _act_grad(dy.contiguous(), x, ctx.act), None


def activation(x, act):
    return ActivationFn.apply(x, act)


class ScaleAddFn(torch.autograd.Function):
    """out = alpha * x (+ add broadcast over the batch): the x*sqrt(d) (+pe) of embedding.py:44-60,75-88 where it
    is not fused into a producing GEMM (LinearNoSubsampling)."""

    @staticmethod
    def forward(ctx, x, alpha, add):
        x = _chk(x, "scale_add input")
        out = torch.empty_like(x)
        add_b = None if add is None else _chk(add.expand_as(x), "scale_add addend")
        hip.call("oe_axpby", x, add_b, x.numel(), float(alpha), 1.0, None, out)
        ctx.alpha = float(alpha)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        hip.call("oe_axpby", dy, None, dy.numel(), ctx.alpha, 0.0, None, dx)
        return dx, None, None


def scale_add(x, alpha, add=None):
    return ScaleAddFn.apply(x, alpha, add)


class AddFn(torch.autograd.Function):
    """x + y with both gradients passed through (encoder_layer.py:108 / decoder_layer.py:106: `x = x + adapt_x`)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _chk(x, "add"), _chk(y, "add")
        out = torch.empty_like(x)
        hip.call("oe_axpby", x, y, x.numel(), 1.0, 1.0, None, out)
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(x, y):
    return AddFn.apply(x, y)


class LossCombineFn(torch.autograd.Function):
    """asr_model.py:150-157 + :196-198 on device scalars, one launch each way:
    att = l * (1 - r) + lr * r;  loss = wc * ctc + (1 - wc) * att  (absent terms: None)."""

    @staticmethod
    def forward(ctx, loss_att, loss_att_r, loss_ctc, ctc_weight, reverse_weight):
        f = lambda t: None if t is None else _chk(t.reshape(1), "loss")
        la, lr, lc = f(loss_att), f(loss_att_r), f(loss_ctc)
        out = _new(1, like=la)
        hip.call("oe_loss_combine", lc, la, lr, float(ctc_weight), float(reverse_weight), 1.0, out)
        ctx.cfg = (float(ctc_weight), float(reverse_weight), lr is not None, lc is not None)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        wc, r, has_r, has_c = ctx.cfg
        g = g.contiguous().reshape(1)
        d = _new(3, like=g)
        hip.call("oe_loss_combine_bwd", g, wc, r, 1.0, d[0:1] if has_c else None, d[1:2], d[2:3] if has_r else None)
        return d[1].view(()), (d[2].view(()) if has_r else None), (d[0].view(()) if has_c else None), None, None


def combine_losses(loss_att, loss_att_r=None, loss_ctc=None, ctc_weight=0.0, reverse_weight=0.0):
    return LossCombineFn.apply(loss_att, loss_att_r, loss_ctc, ctc_weight, reverse_weight)


class CmvnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, istd):
        x = _chk(x, "cmvn")
        y = torch.empty_like(x)
        hip.call("oe_global_cmvn", x, mean, istd, x.numel(), x.shape[-1], y)
        return y

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError("GlobalCMVN is applied to input features, which carry no gradient")


def global_cmvn(x, mean, istd):
    return CmvnFn.apply(x, mean, istd)


# --------------------------------------------------------------------------- #
# Position-wise feed forward (+ optional fused residual / outer dropout)
# --------------------------------------------------------------------------- #
# The fused feed-forward kernel (csrc/ffn.hip) consumes the weights pre-split to bf16 planes in MFMA-fragment order.  They
# are re-packed on every call (~3 us for the two 1 MB matrices): the optimizer moves the weights every step through a raw
# kernel, which no tensor version counter sees, so a cache could only be trusted for frozen models.
FUSED_FFN = os.environ.get("OE_FUSED_FFN", "1") == "1"
# One block of the fused kernel owns 32 rows and streams ALL of W1 and W2: it pays once there are enough rows to fill the
# chip (config 2: 7936 rows = 248 blocks, 71 us incl. the weight packing against 79 us for the two GEMMs); the decoders'
# 992 rows are 31 blocks that each still take the full ~56 us, against 41 us for their two small GEMMs.
FUSED_FFN_MIN_ROWS = int(os.environ.get("OE_FUSED_FFN_MIN_ROWS", "4096"))
# the input gradient in one launch too (oe_ffn_bwd).  None = automatic: on in precision 6 (csrc/ffn6.hip: 61 us against 92 for the
# two GEMMs at 7936 rows, profiles/r04_ffn6.md), off in precision 1 / 3 (measured slower in the step, profiles/r02_experiments.md)
FUSED_FFN_BWD = {"1": True, "0": False}.get(os.environ.get("OE_FUSED_FFN_BWD", ""), None)


def _ffn_bwd_fused() -> bool:
    return (hip.GEMM_PRECISION == 6) if FUSED_FFN_BWD is None else bool(FUSED_FFN_BWD)


def _ffn_fused_ok(x2, w1, w2, act, res2):
    d, ff = w1.shape[1], w1.shape[0]
    if hip.GEMM_PRECISION == 0 or x2.shape[0] < FUSED_FFN_MIN_ROWS or not hip.lib().oe_ffn_supported(d, ff, hip.GEMM_PRECISION, act):
        return False
    ok = lambda t: t is None or (t.data_ptr() % 16 == 0 and t.stride(-1) == 1)
    return (w1.is_contiguous() and w2.is_contiguous() and ok(x2) and ok(res2) and x2.stride(0) % 4 == 0 and
            (res2 is None or res2.stride(0) % 4 == 0) and w1.data_ptr() % 16 == 0 and w2.data_ptr() % 16 == 0)


# The packed weights of every feed-forward that has been seen, refreshed by ONE table-driven launch (oe_ffn_pack_weights_table) the
# first time any of them is needed after the weights may have changed (planes.weights_generation: a new forward pass, an optimizer
# step, a graph replay; plus the tensors' own version counters for writes through torch) - 48 pack launches per step at config 2
# otherwise (forward and backward orientation of 24 feed-forwards, 4.5 us each on the step's chain).

_FFN = _PackTable(9)        # rows: { W1, W2, w1p, w2p, w2tp, w1tp, d, ff, planes }; key = (W1 address, W2 address, precision)
FFN_PACK_TABLE = os.environ.get("OE_FFN_PACK_TABLE", "1") == "1"


def ffn_packs_clear():
    _FFN.clear()
    row_packs_clear()


def pack_tables_sweep():
    """Neutralise the table rows of weights that have died since the last look (planes.new_pass, TrainEngine.replay: before anything
    - a captured pack launch included - can touch them: their memory may be unmapped by now, torch.cuda.empty_cache)."""
    if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
        _FFN.sweep(lambda k, e: e["w1"]() is None or e["w2"]() is None)
        _FFN.upload()
        _ROW.sweep(lambda k, e: e["owner"]() is None)
        _ROW.upload()


def _ffn_packed(w1, w2, d, ff, bwd=False):
    """Packed W1 / W2 (forward) or W2^T / W1^T (bwd) of one feed-forward, fresh for the weights' current values."""
    prec = hip.GEMM_PRECISION
    nbytes = hip.lib().oe_ffn_packed_bytes(d, ff, prec)
    capturing = torch.cuda.is_current_stream_capturing()
    key = (w1.data_ptr(), w2.data_ptr(), prec)
    ent = _FFN.entries.get(key) if FFN_PACK_TABLE else None
    if ent is not None and ((ent["d"], ent["ff"]) != (d, ff) or ent["w1"]() is not w1 or ent["w2"]() is not w2):
        if capturing:
            ent = False                                          # (no table surgery inside a capture: a pack of this call's own)
        else:
            _FFN.drop(key)
            ent = None
    if ent is None and FFN_PACK_TABLE and not capturing and isinstance(w1, torch.nn.Parameter) and isinstance(w2, torch.nn.Parameter):
        # first sight (outside a capture: these buffers outlive every graph): persistent buffers, a table row
        bufs = [torch.empty(nbytes, dtype=torch.uint8, device=w1.device) for _ in range(4)]
        ent = dict(w1=weakref.ref(w1), w2=weakref.ref(w2), bufs=bufs, d=d, ff=ff, gen=[-1, -1], ver=[None, None])
        # (the row carries its own plane count: the table may hold feed-forwards registered under other precisions, whose
        # buffers are smaller - bench.py's extra mode-3 / mode-1 legs; a refresh in THIS precision must not overrun them)
        if not _FFN.add(key, [key[0], key[1]] + [b.data_ptr() for b in bufs] + [d, ff, {6: 3, 3: 2}.get(prec, 1)], ent, w1.device):
            ent = None
    if not ent:                                                  # unregistered (inside a capture, plain tensors): a pack of its own
        a = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
        b = torch.empty(nbytes, dtype=torch.uint8, device=w1.device)
        hip.call("oe_ffn_pack_weights_bwd" if bwd else "oe_ffn_pack_weights", w1, w2, d, ff, prec, a, b)
        return a, b
    gen, ver, kind = _planes.weights_generation(), (w1._version, w2._version), int(bwd)
    if ent["gen"][kind] != gen or ent["ver"][kind] != ver:
        if not capturing:
            _FFN.sweep(lambda k, e: e["w1"]() is None or e["w2"]() is None)
            _FFN.upload()
        if not _FFN.dirty and len(_FFN.entries) > 1 and ent["gen"][0] != gen and ent["gen"][1] != gen:
            # nobody has refreshed anything in this generation yet: all feed-forwards, both orientations, one launch
            alive = _FFN.entries.values()
            hip.call("oe_ffn_pack_weights_table", _FFN.dev, _FFN.n, max(e["d"] for e in alive), max(e["ff"] for e in alive), prec)
            _pack_done(_FFN)
            for e in alive:
                w1e, w2e = e["w1"](), e["w2"]()
                v = (w1e._version, w2e._version) if (w1e is not None and w2e is not None) else None
                e["gen"], e["ver"] = [gen, gen], [v, v]
        else:
            hip.call("oe_ffn_pack_weights_bwd" if bwd else "oe_ffn_pack_weights", w1, w2, d, ff, prec, ent["bufs"][2 * kind], ent["bufs"][2 * kind + 1])
            ent["gen"][kind], ent["ver"][kind] = gen, ver
    else:
        _pack_wait(_FFN)
    return ent["bufs"][2 * kind], ent["bufs"][2 * kind + 1]


class FeedForwardFn(torch.autograd.Function):
    """y = [residual + out_scale * drop_out(] w_2(drop_in(act(w_1 x + b_1))) + b_2 [)]
    positionwise_feed_forward.py:36-43 and encoder_layer.py:81-83,104-106 / decoder_layer.py:104-106."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, p_in, residual, out_scale, p_out):
        x = _chk(x, "feed_forward", keep_pending=True)
        d = x.shape[-1]
        x2 = x.reshape(-1, d)
        M, ff = x2.shape[0], w1.shape[0]
        s_in, s_out = (next_seed() if p_in > 0 else 0), (next_seed() if p_out > 0 else 0)
        res2 = None if residual is None else _chk(residual, "residual").reshape(-1, w2.shape[0])
        lnf_p = _lnf_take(x2)                     # x may be the output of a parked LayerNorm forward (_PENDING_LNF)
        fused_now = FUSED_FFN and _ffn_fused_ok(x2, w1, w2, act, res2)
        lnf = None
        if lnf_p is not None:
            if (fused_now and hip.GEMM_PRECISION == 6 and d == 256 and ff % 256 == 0 and hip.lib().oe_ffn6_config(-1) in (0, 3) and
                    x2.is_contiguous() and M == lnf_p["rows"]):
                lnf = dict(x=lnf_p["x"], gamma=lnf_p["gamma"], beta=lnf_p["beta"], eps=lnf_p["eps"], y=lnf_p["y"], stats=lnf_p["stats"],
                           rowmask=lnf_p["rowmask"], gamma2=lnf_p.get("gamma2"), beta2=lnf_p.get("beta2"), eps2=lnf_p.get("eps2"),
                           u=lnf_p.get("u"), stats2=lnf_p.get("stats2"))
                if lnf["gamma2"] is not None and (res2 is None or lnf["u"] is None or res2.data_ptr() != lnf["u"].data_ptr()):
                    lnf = None                    # the pair form is for `residual = u` (EncoderLayer's pre): anything else, the plain way
                    _resolve_lnf(lnf_p)
            else:
                _resolve_lnf(lnf_p)
        if fused_now:
            # one kernel (csrc/ffn.hip): the (M, ff) intermediate stays in registers; pre / a are written only when a
            # backward will read them
            need = any(ctx.needs_input_grad)
            pre = _new(M, ff, like=x) if need else None
            a = _new(M, ff, like=x) if need else None
            w1p, w2p = _ffn_packed(w1, w2, d, ff)
            y = _new(M, d, like=x)
            hip.ffn_fwd(x2, w1p, b1, w2p, b2, M, d, ff, act, drop_in=p_in, seed_in=s_in, drop_out=p_out, seed_out=s_out,
                        seed_dev=_seed_dev, pre_out=pre, act_out=a, residual=res2, ldr=0 if res2 is None else res2.stride(0),
                        beta=out_scale, y=y, lnf=lnf)
            if lnf is not None:
                global LN_FWD_FUSED_LAUNCHES
                lnf_p["done"] = True
                LN_FWD_FUSED_LAUNCHES += 1
            ctx.save_for_backward(x2, w1, w2, pre, a)
            ctx.biases = (b1, b2)
            ctx.cfg = (act, p_in, s_in, out_scale, p_out, s_out, residual is not None, x.shape)
            ctx.fused = True
            # (the backward's fused kernel can take the next LayerNorm's backward as its prologue: _PENDING_LN)
            lnf = (LN_BWD_FUSE and FUSE_OUT_DROP and hip.GEMM_PRECISION == 6 and d == 256 and M >= ROWGEMM_MIN_ROWS and ff % 256 == 0 and
                   _ffn_bwd_fused() and hip.lib().oe_ffn6_config(-1) in (0, 3) and not _planes.active())
            return _tag_out_drop(y.view(*x.shape[:-1], w2.shape[0]), out_scale, p_out, s_out, ln_fuse=lnf)
        ctx.fused = False
        pre = _new(M, ff, like=x)
        if act in GEMM_FUSED_ACTS:
            a = gemm_nt(x2, w1, b1, act=act, preact_out=pre, ld_aux=ff, drop_p=p_in, seed=s_in, seed_dev=_seed_dev, out_planes=True)
        else:
            gemm_nt(x2, w1, b1, out=pre)
            a = torch.empty_like(pre)
            hip.call("oe_act_fwd", pre, pre.numel(), act, a)
            if p_in > 0:
                a = dropout_scale(a, 1.0, p_in, s_in)
        y = gemm_nt(a, w2, b2, drop_p=p_out, seed=s_out, seed_dev=_seed_dev, residual=res2,
                    ldr=0 if res2 is None else res2.stride(0), beta=out_scale)
        ctx.save_for_backward(x2, w1, w2, pre, a)
        ctx.biases = (b1, b2)
        ctx.cfg = (act, p_in, s_in, out_scale, p_out, s_out, residual is not None, x.shape)
        return _tag_out_drop(y.view(*x.shape[:-1], w2.shape[0]), out_scale, p_out, s_out)

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, pre, a = ctx.saved_tensors
        act, p_in, s_in, out_scale, p_out, s_out, has_res, in_shape = ctx.cfg
        dy = dy.contiguous()
        dy2 = dy.view(-1, w2.shape[0])
        g2 = dy2 if (p_out == 0 and out_scale == 1.0) else _out_drop_grad(dy2, out_scale, p_out, s_out)
        b1, b2 = ctx.biases
        pend = _PENDING_LN.pop(g2.data_ptr(), None) if _PENDING_LN else None     # g2 may be a parked LayerNorm backward
        if ctx.fused and _ffn_bwd_fused() and g2.stride(0) % 4 == 0:
            # both input-gradient GEMMs in one launch (csrc/ffn.hip, oe_ffn_bwd): dH is written once and never re-read here
            M, d, ff = g2.shape[0], w2.shape[0], w1.shape[0]
            prec = hip.GEMM_PRECISION
            nbytes = hip.lib().oe_ffn_packed_bytes(d, ff, prec)
            w2tp, w1tp = _ffn_packed(w1, w2, d, ff, bwd=True)
            dh, dx = _new(M, ff, like=g2), _new(M, d, like=g2)
            ln = None
            if pend is not None and not pend.get("done"):
                if (prec == 6 and d == 256 and ff % 256 == 0 and hip.lib().oe_ffn6_config(-1) in (0, 3) and g2.is_contiguous() and
                        g2.data_ptr() == pend["g"].data_ptr() and M == pend["rows"]):
                    ln = _ln_dict(pend)
                else:
                    _resolve_ln(pend)
            # FIRST launch of this backward: it makes g2 when that is a parked LayerNorm backward (the weight gradient reads it after)
            hip.ffn_bwd(g2, w2tp, w1tp, M, d, ff, act, drop_in=p_in, seed_in=s_in, seed_dev=_seed_dev, pre=pre, dh=dh, dx=dx, ln=ln)
            if ln is not None:
                global LN_BWD_FUSED_LAUNCHES
                pend["done"] = True
                LN_BWD_FUSED_LAUNCHES += 1
                _ln_reduce(pend)
            dw2, db2 = wgrad_bias(w2, b2, g2, a)
            dw1, db1 = wgrad_bias(w1, b1, dh, x2)
            return dx.view(in_shape), dw1, db1, dw2, db2, None, None, (dy if has_res else None), None, None
        if pend is not None:
            _resolve_ln(pend)
        dw2, db2 = wgrad_bias(w2, b2, g2, a)
        if act in GEMM_FUSED_ACTS:
            dh = gemm_nn(g2, w2, act=act, actgrad_in=pre, ld_aux=pre.stride(0), drop_p=p_in, seed=s_in, seed_dev=_seed_dev, out_planes=True)
        else:
            dh = gemm_nn(g2, w2)
            if p_in > 0:
                dh = dropout_scale(dh, 1.0, p_in, s_in)
            dh = _act_grad(dh, pre, act)
        dw1, db1 = wgrad_bias(w1, b1, dh, x2)
        dx = gemm_nn(dh, w1).view(in_shape)
        return dx, dw1, db1, dw2, db2, None, None, (dy if has_res else None), None, None


def feed_forward(x, w1, b1, w2, b2, act, p_in=0.0, residual=None, out_scale=1.0, p_out=0.0):
    return FeedForwardFn.apply(x, w1, b1, w2, b2, act, p_in, residual, out_scale, p_out)


# --------------------------------------------------------------------------- #
# Multi-head attention (plain and relative-position)
# --------------------------------------------------------------------------- #
def mask_bytes(mask: torch.Tensor) -> torch.Tensor:
    """A boolean / integer mask as contiguous uint8; a bool tensor is reinterpreted in place (no launch)."""
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)
    return (mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)).contiguous()


def _mask_u8(mask: Optional[torch.Tensor]):
    if mask is None:
        return None, (0, 0)
    m = mask_bytes(mask)
    assert m.dim() == 3, "mask must be (B, 1|T1, T2)"
    return m, (m.shape[1] * m.shape[2], 0 if m.shape[1] == 1 else m.shape[2])


# Set (by the caller that built the masks: decode paths, the LM) while every self-attention mask in flight is causal - a
# (B, L, L) mask that is zero above the diagonal: the forward kernel then skips the key blocks above it.  A hint only: the
# mask is still applied, the results are identical.
CAUSAL_SELF_ATTENTION = False


class causal_self_attention:
    def __enter__(self):
        global CAUSAL_SELF_ATTENTION
        self.old, CAUSAL_SELF_ATTENTION = CAUSAL_SELF_ATTENTION, True

    def __exit__(self, *exc):
        global CAUSAL_SELF_ATTENTION
        CAUSAL_SELF_ATTENTION = self.old
        return False


class AttentionFn(torch.autograd.Function):
    """attention.py:99-117 (MultiHeadedAttention.forward) and :166-209
    (RelPositionMultiHeadedAttention.forward) incl. linear_q/k/v/out, with an
    optional fused `residual + dropout(.)` of the caller (encoder_layer.py:89)."""

    @staticmethod
    def forward(ctx, xq, xkv, wq, bq, wk, bk, wv, bv, wo, bo, mask, pos_emb, wpos, pu, pv, H, p_attn, residual, p_out, pp_in=None):
        xq = _chk(xq, "attention query", keep_pending=True)        # (a parked LayerNorm forward: the q / k / v projection below makes it)
        self_attn = xkv is None
        B, T1, d = xq.shape
        D = d // H
        scale = 1.0 / math.sqrt(D)
        xq2 = xq.view(-1, d)
        if self_attn:
            T2 = T1
            wqkv, _ = _fused_rows((wq, wk, wv))
            bqkv, _ = _fused_rows((bq, bk, bv))
            qkv = gemm_nt(xq2, wqkv, bqkv)                       # (B*T, 3d)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            qs = ks = vs = (T1 * 3 * d, 3 * d)
            xkv2 = xq2
        else:
            xkv = _chk(xkv, "attention key/value")
            T2 = xkv.shape[1]
            xkv2 = xkv.view(-1, d)
            q = gemm_nt(xq2, wq, bq)
            wkv, _ = _fused_rows((wk, wv))
            bkv, _ = _fused_rows((bk, bv))
            kv = gemm_nt(xkv2, wkv, bkv)                         # (B*T2, 2d)
            k, v = kv[:, :d], kv[:, d:]
            qs, ks, vs = (T1 * d, d), (T2 * 2 * d, 2 * d), (T2 * 2 * d, 2 * d)
            qkv = None
        rel = pos_emb is not None
        kp = keybias = pp = None
        k_att, k_att_s = k, ks
        if rel:
            assert self_attn and pos_emb.shape[-2] == T2
            pe2 = _chk(pos_emb, "pos_emb").reshape(-1, d)
            pp = pp_in if pp_in is not None else gemm_nt(pe2, wpos)     # (T, d): linear_pos has no bias
            kp = _new(B, T2, d, like=xq)
            keybias = _new(B, H, T2, like=xq)
            hip.call("oe_relpos_prepare", k, ks[0], ks[1], pp, d, pu, pv, B, T2, H, D, scale, kp, keybias)
            k_att, k_att_s = kp, (T2 * d, d)
        m8, mstr = _mask_u8(mask)
        s_att, s_out = (next_seed() if p_attn > 0 else 0), (next_seed() if p_out > 0 else 0)
        att = _new(B, T1, d, like=xq)
        lse = _new(B, H, T1, like=xq)
        a = hip.attn_args(q, k_att, v, att, lse, B, H, T1, T2, D, scale, q_strides=qs, k_strides=k_att_s, v_strides=vs,
                          o_strides=(T1 * d, d), mask=m8, mask_strides=mstr, keybias=keybias, drop_p=p_attn, seed=s_att,
                          seed_dev=_seed_dev, causal=CAUSAL_SELF_ATTENTION and self_attn)
        hip.attention_fwd(a)
        res2 = None if residual is None else _chk(residual, "residual").view(-1, d)
        y = gemm_nt(att.view(-1, d), wo, bo, drop_p=p_out, seed=s_out, seed_dev=_seed_dev, residual=res2,
                    ldr=0 if res2 is None else d)
        ctx.save_for_backward(xq2, xkv2 if not self_attn else None, wq, wk, wv, wo, qkv, None if self_attn else q,
                              None if self_attn else kv, kp, keybias, pp, None if not rel else pe2, wpos, pu, pv, m8, att, lse)
        ctx.cfg = (self_attn, rel, B, T1, T2, d, H, D, scale, p_attn, s_att, p_out, s_out, residual is not None, mstr)
        ctx.biases = (bq, bk, bv, bo)
        ctx.pp_external = pp_in is not None
        return _tag_out_drop(y.view(B, T1, d), 1.0, p_out, s_out, ln_fuse=_ln_fuse_ok(B * T1, d, wo))

    @staticmethod
    def backward(ctx, dy):
        (xq2, xkv2, wq, wk, wv, wo, qkv, q_x, kv_x, kp, keybias, pp, pe2, wpos, pu, pv, m8, att, lse) = ctx.saved_tensors
        self_attn, rel, B, T1, T2, d, H, D, scale, p_attn, s_att, p_out, s_out, has_res, mstr = ctx.cfg
        dy = dy.contiguous()
        dy2 = dy.view(-1, d)
        g = dy2 if p_out == 0 else _out_drop_grad(dy2, 1.0, p_out, s_out)
        att2 = att.view(-1, d)
        bq, bk, bv, bo = ctx.biases
        datt = gemm_nn(g, wo)                    # FIRST: g may be a parked LayerNorm backward that this launch makes (_PENDING_LN)
        dwo, dbo = wgrad_bias(wo, bo, g, att2)
        delta = _new(B, H, T1, like=dy)
        if self_attn:
            dqkv = _new(B * T1, 3 * d, like=dy)
            q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
            dq, dk, dv = dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:]
            qs = ks = vs = (T1 * 3 * d, 3 * d)
        else:
            q, k, v = q_x, kv_x[:, :d], kv_x[:, d:]
            dq = _new(B * T1, d, like=dy)
            dkv = _new(B * T2, 2 * d, like=dy)
            dk, dv = dkv[:, :d], dkv[:, d:]
            qs, ks, vs = (T1 * d, d), (T2 * 2 * d, 2 * d), (T2 * 2 * d, 2 * d)
        k_att, k_att_s, dk_att, dkb = k, ks, dk, None
        if rel:
            dkp = _new(B, T2, d, like=dy)
            dkb = _new(B, H, T2, like=dy)
            k_att, k_att_s, dk_att = kp, (T2 * d, d), dkp
        a = hip.attn_args(q, k_att, v, att, lse, B, H, T1, T2, D, scale, q_strides=qs, k_strides=k_att_s, v_strides=vs,
                          o_strides=(T1 * d, d), mask=m8, mask_strides=mstr, keybias=keybias, drop_p=p_attn, seed=s_att,
                          seed_dev=_seed_dev, d_out=datt, dq=dq, dk=dk_att, dv=dv, dkeybias=dkb, delta=delta)
        hip.attention_bwd(a)
        dwpos = rpu = rpv = None
        if rel:
            dpp = _new(T2, d, like=dy)
            (dpu, rpu), (dpv, rpv) = grad_sink(pu), grad_sink(pv)
            hip.call("oe_relpos_backward", dkp, dkb, k, ks[0], ks[1], pp, d, pu, pv, B, T2, H, D, scale, dk, dpp, d, dpu, dpv)
            if not ctx.pp_external:
                dwpos = wgrad(wpos, dpp, pe2)

        def split_or_sink(parts_w, parts_b, dy_fused, x_in):
            """Weight/bias gradients of a fused projection: one GEMM / one column sum."""
            fw, in_place = _fused_rows(parts_w)
            dx_in = gemm_nn(dy_fused, fw)
            if in_place and _arena.grad_target(parts_w[0]) is not None:
                n = dy_fused.shape[1]
                gw = _arena.grad_target(parts_w[0])
                gw_all = torch.as_strided(gw, (n, x_in.shape[1]), (x_in.shape[1], 1))
                gb = _arena.grad_target(parts_b[0])
                gb_all = torch.as_strided(gb, (n,), (1,))
                pl = wgrad_planes(dy_fused, x_in)
                _run_wgrad(True, (dy_fused, x_in) + _pl_tensors(pl), lambda: gemm_tn(dy_fused, x_in, out=gw_all, bias_out=gb_all, planes=pl),
                           desc=dict(dy=dy_fused, x=x_in, out=gw_all, alpha=1.0, alpha_dev=None, bias_out=gb_all))
                return dx_in, [None] * len(parts_w), [None] * len(parts_b)
            dwf, dbf = gemm_tn(dy_fused, x_in), colsum(dy_fused)
            ws, bs, o = [], [], 0
            for pw, pb in zip(parts_w, parts_b):
                r = pw.shape[0]
                tw, tb = _arena.grad_target(pw), _arena.grad_target(pb)
                if tw is not None:
                    tw.add_(dwf[o:o + r]); tb.add_(dbf[o:o + r]); ws.append(None); bs.append(None)
                else:
                    ws.append(dwf[o:o + r]); bs.append(dbf[o:o + r])
                o += r
            return dx_in, ws, bs

        if self_attn:
            dx, (dwq, dwk, dwv), (dbq, dbk, dbv) = split_or_sink((wq, wk, wv), (bq, bk, bv), dqkv, xq2)
            dx = dx.view(B, T1, d)
            dxkv = None
        else:
            dx = gemm_nn(dq, wq).view(B, T1, d)
            dwq, dbq = wgrad_bias(wq, bq, dq, xq2)
            dxkv, (dwk, dwv), (dbk, dbv) = split_or_sink((wk, wv), (bk, bv), dkv, xkv2)
            dxkv = dxkv.view(B, T2, d)
        return (dx, dxkv, dwq, dbq, dwk, dbk, dwv, dbv, dwo, dbo, None, None, dwpos, rpu, rpv, None, None,
                (dy if has_res else None), None, (dpp if (rel and ctx.pp_external) else None))


def attention(xq, xkv, wq, bq, wk, bk, wv, bv, wo, bo, mask, H, p_attn=0.0, pos_emb=None, wpos=None, pu=None, pv=None,
              residual=None, p_out=0.0, pp=None):
    return AttentionFn.apply(xq, xkv, wq, bq, wk, bk, wv, bv, wo, bo, mask, pos_emb, wpos, pu, pv, H, p_attn, residual, p_out, pp)


class ScoresAttentionFn(torch.autograd.Function):
    """attention.py:65-97 up to (not including) linear_out, on MATERIALISED scores: masked softmax -> 0-fill -> dropout ->
    attn @ value -> heads merged.  Module-API surface only (MultiHeadedAttention.forward_attention); forward() uses
    AttentionFn, which never builds the (B,H,T1,T2) tensor.  value (B,H,T2,dk) in any strides with a contiguous last
    dim (forward_qkv returns transposed views); returns (B,T1,H*dk)."""

    @staticmethod
    def forward(ctx, value, scores, mask, p):
        scores = _chk(scores, "scores")
        if not value.is_cuda or value.dtype != torch.float32:
            raise TypeError("value: openeat_amd ops need float32 CUDA tensors; there is no CPU fallback")
        if value.stride(-1) != 1:
            value = value.contiguous()
        B, H, T1, T2 = scores.shape
        dk = value.shape[-1]
        m8, mstr = _mask_u8(mask)
        seed = next_seed() if p > 0 else 0
        y = torch.empty_like(scores)
        att = torch.empty_like(scores) if p > 0 else y
        hip.call("oe_masked_softmax_fwd", scores, m8, mstr[0], mstr[1], B, H, T1, T2, float(p), seed, _seed_dev, y, att)
        out = _new(B, T1, H, dk, like=scores)
        for b in range(B):
            for h in range(H):
                v = value[b, h]
                hip.gemm(att[b, h], v, out[b, :, h], T1, dk, T2, lda=T2, ldb=v.stride(0), ldc=H * dk, b_kmajor=True)
        ctx.save_for_backward(value, y, att)
        ctx.cfg = (p, seed)
        return out.view(B, T1, H * dk)

    @staticmethod
    def backward(ctx, dout):
        value, y, att = ctx.saved_tensors
        p, seed = ctx.cfg
        B, H, T1, T2 = y.shape
        dk = value.shape[-1]
        dout = dout.contiguous().view(B, T1, H, dk)
        datt = torch.empty_like(y)
        dv = torch.zeros(B, H, T2, dk, device=y.device, dtype=torch.float32)
        for b in range(B):
            for h in range(H):
                g, v = dout[b, :, h], value[b, h]
                hip.gemm(g, v, datt[b, h], T1, T2, dk, lda=H * dk, ldb=v.stride(0), ldc=T2)                       # dO @ V^T
                hip.gemm(att[b, h], g, dv[b, h], T2, dk, T1, lda=T2, ldb=H * dk, ldc=dk, a_kmajor=True, b_kmajor=True,
                         atomic_out=True)                                                                         # P^T @ dO
        ds = torch.empty_like(y)
        hip.call("oe_masked_softmax_bwd", y, datt, B * H * T1, T2, float(p), seed, _seed_dev, ds)
        return dv, ds, None, None


def scores_attention(value, scores, mask, p=0.0):
    return ScoresAttentionFn.apply(value, scores, mask, p)


# linear_pos(pos_emb) (attention.py:185) depends on the positional table and one weight only: with POS_PROJ_AHEAD the
# encoder computes it for every layer on the side stream when it starts, as an autograd node of its own - so the
# projection AND its weight gradient (which autograd runs on the same stream) leave the layers' critical chain.
# Off for eager multi-rank steps: there the backward hooks hand finished slices of the gradient arena to the collective,
# and this node's weight gradient would run after its layer's slice had gone.
POS_PROJ_AHEAD = False


class PosProjFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos_emb, wpos):
        pe2 = _chk(pos_emb, "pos_emb").reshape(-1, wpos.shape[1])
        ctx.save_for_backward(pe2, wpos)
        return gemm_nt(pe2, wpos)

    @staticmethod
    def backward(ctx, dpp):
        pe2, wpos = ctx.saved_tensors
        return None, wgrad(wpos, dpp.contiguous(), pe2)


def pos_proj(pos_emb, wpos):
    return PosProjFn.apply(pos_emb, wpos)


# --------------------------------------------------------------------------- #
# Conformer convolution module
# --------------------------------------------------------------------------- #
DWCONV_LN_FUSED = os.environ.get("OE_DWCONV_LN", "1") == "1"


class ConvModuleFn(torch.autograd.Function):
    """convolution.py:72-120: mask -> pw1 -> GLU -> depthwise -> LayerNorm -> act -> pw2 -> mask
    (+ optional fused `residual + dropout(.)`, encoder_layer.py:95)."""

    @staticmethod
    def forward(ctx, x, rowmask, w1, b1, wd, bd, g, b, w2, b2, K, causal, act, residual, p_out, input_masked):
        x = _chk(x, "conv_module", keep_pending=not (rowmask is not None and not input_masked))    # (pointwise_conv1 below makes a parked LayerNorm)
        B, T, d = x.shape
        x2 = x.view(-1, d)
        xm = x2 if (rowmask is None or input_masked) else dropout_scale(x2, 1.0, 0.0, 0, rowmask)
        w1m, w2m = w1.view(2 * d, d), w2.view(d, d)
        a = gemm_nt(xm, w1m, b1)                                  # (B*T, 2d)
        yc = _new(B * T, d, like=x)
        gpad = None
        if causal:   # the reference pads before pointwise_conv1: padded frames hold GLU(bias)
            gpad = _new(d, like=x)
            hip.call("oe_glu_fwd", b1, 1, d, gpad)
        z = torch.empty_like(yc)
        stats = _new(B * T, 2, like=x)
        if DWCONV_LN_FUSED and not _planes.active():
            # the depthwise convolution normalises its own rows: one launch and one pass over (B*T, d) less per layer
            hip.call("oe_dwconv_glu_ln_fwd", a, wd, bd, gpad, B, T, d, K, int(causal), yc, g, b, 1e-5, act, z, stats)
        else:
            hip.call("oe_dwconv_glu_fwd", a, wd, bd, gpad, B, T, d, K, int(causal), yc)
            _ln_fwd(yc, g, b, 1e-5, B * T, d, None, act, z, stats)
        s_out = next_seed() if p_out > 0 else 0
        res2 = None if residual is None else _chk(residual, "residual").view(-1, d)
        y = gemm_nt(z, w2m, b2, drop_p=p_out, seed=s_out, seed_dev=_seed_dev, rowmask=rowmask, residual=res2,
                    ldr=0 if res2 is None else d)
        ctx.save_for_backward(xm, rowmask, w1, wd, g, b, w2, a, yc, stats, z, gpad)
        ctx.biases = (b1, bd, b2)
        ctx.cfg = (B, T, d, K, causal, act, p_out, s_out, residual is not None)
        return _tag_out_drop(y.view(B, T, d), 1.0, p_out, s_out, rowmask, ln_fuse=_ln_fuse_ok(B * T, d, w2m))

    @staticmethod
    def backward(ctx, dy):
        xm, rowmask, w1, wd, g, b, w2, a, yc, stats, z, gpad = ctx.saved_tensors
        b1, bd, b2 = ctx.biases
        B, T, d, K, causal, act, p_out, s_out, has_res = ctx.cfg
        dy = dy.contiguous()
        dy2 = dy.view(-1, d)
        w1m, w2m = w1.view(2 * d, d), w2.view(d, d)
        gq = dy2 if (p_out == 0 and rowmask is None) else _out_drop_grad(dy2, 1.0, p_out, s_out, rowmask)
        dyc = torch.empty_like(yc)
        (dg, rg), (dbeta, rbeta) = grad_sink(g), grad_sink(b)
        to_arena = rg is None and rbeta is None
        # the norm + activation behind the depthwise convolution: its backward as the EPILOGUE of pointwise_conv2's input gradient
        # (oe_rowgemm6's lne arguments: the block owns whole rows), when that launch is the 256 <- 256 row-block kernel
        epi_ln = None
        if (LN_EPI_FUSE and ROWGEMM and hip.GEMM_PRECISION == 6 and d == 256 and B * T >= ROWGEMM_MIN_ROWS and act in GEMM_FUSED_ACTS and
                not _planes.active() and to_arena):
            epi_ln = dict(x=yc, stats=stats, gamma=g, beta=b, act=act, dx=dyc, ws=_ln_ws(yc, B * T, d), done=False)
        dz = gemm_nn(gq, w2m, ln_epi=epi_ln)     # FIRST: gq may be a parked LayerNorm backward that this launch makes (_PENDING_LN)
        dw2, db2 = wgrad_bias(w2, b2, gq, z)
        if epi_ln is not None and epi_ln["done"]:
            global LN_EPI_FUSED_LAUNCHES
            LN_EPI_FUSED_LAUNCHES += 1
            t = LN_TABLE
            if t is None:
                hip.call("oe_layernorm_param_reduce", epi_ln["ws"], B * T, d, dg, dbeta)
            else:
                t["entries"].append((epi_ln["ws"].data_ptr(), B * T, d, dg.data_ptr(), dbeta.data_ptr()))
                t["keep"].append(epi_ln["ws"])
                t["max_rows"], t["max_d"] = max(t["max_rows"], B * T), max(t["max_d"], d)
        else:
            _ln_bwd(dz, yc, g, b, act, stats, B * T, d, None, None, dyc, dg, dbeta, to_arena)
        da = torch.empty_like(a)
        (dwd, rwd), (dbd, rbd) = grad_sink(wd), grad_sink(bd)
        dgpad = torch.zeros(d, device=dy.device) if causal else None
        ws = _new(hip.lib().oe_dwconv_glu_bwd_workspace_floats(B, T, d, K), like=dy)
        hip.call("oe_dwconv_glu_bwd", a, dyc, wd, gpad, B, T, d, K, int(causal), da, dwd, dbd, dgpad, ws)
        (db1, rb1) = grad_sink(b1)
        tw1 = _arena.grad_target(w1)
        ow1 = tw1.view(2 * d, d) if tw1 is not None else _new(2 * d, d, like=dy, zero=True)
        pl = wgrad_planes(da, xm)
        _run_wgrad(tw1 is not None and rb1 is None and not causal, (da, xm) + _pl_tensors(pl), lambda: gemm_tn(da, xm, out=ow1, bias_out=db1, planes=pl),
                   desc=dict(dy=da, x=xm, out=ow1, alpha=1.0, alpha_dev=None, bias_out=db1))
        dw1 = None if tw1 is not None else ow1.view(w1.shape)
        if causal:
            db1_pad = torch.empty_like(db1)
            hip.call("oe_glu_bwd", b1, dgpad, 1, d, db1_pad)
            hip.call("oe_axpby", db1_pad, db1, 2 * d, 1.0, 1.0, None, db1)
        dx = gemm_nn(da, w1m, rowmask=rowmask).view(B, T, d)
        return dx, None, dw1, rb1, rwd, rbd, rg, rbeta, dw2, db2, None, None, None, (dy if has_res else None), None, None


def conv_module(x, rowmask, w1, b1, wd, bd, g, b, w2, b2, K, causal, act, residual=None, p_out=0.0, input_masked=False):
    return ConvModuleFn.apply(x, rowmask, w1, b1, wd, bd, g, b, w2, b2, K, causal, act, residual, p_out, input_masked)


# --------------------------------------------------------------------------- #
# Conv2d subsampling (1/4) + output Linear + positional scaling
# --------------------------------------------------------------------------- #
CONV_DGRAD_IMPLICIT = os.environ.get("OE_CONV_DGRAD", "implicit") != "col2im"
# pre-split conv GEMMs walk the reduction channel-chunk major (oe_gemm_args.conv_korder): the taps of one 32-channel chunk back to
# back, so the overlapping windows of neighbouring output positions hit L2.  Only for problems the pre-split kernel surely takes.
CONV_KORDER = os.environ.get("OE_CONV_KORDER", "1") == "1"
CONV_KORDER_MIN_ROWS = 32768


def _korder_cols(w2d, taps, C):
    """(rows, taps * C) with columns (tap, c) -> columns (c // 32, tap, c % 32)."""
    r = w2d.shape[0]
    return w2d.view(r, taps, C // 32, 32).permute(0, 2, 1, 3).contiguous().view(r, taps * C)
# the conv weight gradient on pre-split operands too (dy leaves the Linear's input-gradient GEMM with planes): 256 x 256 tiles
# with the kernel's own split of the 150784-deep reduction, 1454 -> 817 us at config 2 (same-box A/B, step 21.9 -> 21.55 ms)
CONV_WGRAD_PLANES = os.environ.get("OE_CONV_WGRAD_PLANES", "1") == "1"
# the conv1 activation as bf16 planes only, its ReLU mask read from plane 0 (ConvSubsamplingFn.forward)
CONV1_PLANES_ONLY = os.environ.get("OE_CONV1_PLANES_ONLY", "1") == "1"
# the zero-padded dy of the stride-2 input gradient written as bf16 planes directly (_conv_dgrad_k3s2)
CONV_PAD_PLANES = os.environ.get("OE_CONV_PAD_PLANES", "1") == "1"


def _conv_dgrad_k3s2(dy, wk, yin, B, Ti, Fi, To, Fo, C, yin_planes=None):
    """Input gradient of Conv2d(C, C, 3, stride 2) (+ the ReLU mask of its input) without a column buffer.
    dx[b, t1, f1, ci] = sum over the taps (kh, kw) with t1 - kh and f1 - kw even of dy[b, (t1-kh)/2, (f1-kw)/2, :] . W[:, ci, kh, kw]:
    per parity class (t1 % 2, f1 % 2) that is a stride-1 convolution of dy with a 2x2 / 1x2 / 2x1 / 1x1 kernel, i.e. an
    implicit GEMM whose A rows are gathered from dy padded with one zero position on every side (so every tap is in
    bounds and the gather needs no predicate) and whose epilogue scatters the rows to dx[:, t1 % 2 :: 2, f1 % 2 :: 2]
    and applies the mask.  The same 2*M*N*K flops as the column-buffer form, 1.4 GB less written and 2.7 GB less gathered
    at config 2."""
    # every class on pre-split operands in channel-chunk order (the launch fails rather than fall back): the padded dy is made
    # as bf16 planes directly - no padded fp32 tensor, no split pass over it
    ni_min, nj_min = Ti // 2, Fi // 2
    direct = bool(_planes.available() and CONV_PAD_PLANES and CONV_KORDER and C % 32 == 0 and B * ni_min * nj_min >= CONV_KORDER_MIN_ROWS)
    dyp_pl = None
    if direct:
        dyp = None
        dyp_pl = _planes.alloc(B * (To + 2) * (Fo + 2), C, dy.device)
        hip.call("oe_pad1_nhwc_planes", dy, B, To, Fo, C, None, dyp_pl.t, dyp_pl.stride)
    else:
        dyp = _new(B, To + 2, Fo + 2, C, like=dy)
        hip.call("oe_pad1_nhwc", dy, B, To, Fo, C, dyp)
    wcls = _new(9 * C * C, like=dy)                    # the four classes' B operands [ci][(window row, window col, co)], one launch
    hip.call("oe_conv_dgrad_k3s2_weights", wk, C, wcls)
    # yin_planes: the stage's input exists as bf16 planes only - the ReLU mask is read from plane 0 (same sign, half the bytes)
    dyin = torch.empty_like(yin) if yin_planes is None else _new(B, Ti, Fi, C, like=dy)
    flat_in, flat_out = (None if dyp is None else dyp.view(-1)), dyin.view(-1)
    flat_y = yin.reshape(-1) if yin_planes is None else yin_planes.t[0].reshape(-1)
    if dyp_pl is None:
        dyp_pl = _planes.of(dyp.view(-1, C)) if _planes.available() else None   # pre-split mode: one pass over the padded dy
    w_off = 0
    for pt in (0, 1):
        KH = 2 if pt == 0 else 1                       # window row 0 is dy row i - 1 (tap 2), row 1 is dy row i (tap 0); odd t1: tap 1
        ni = (Ti + 1 - pt) // 2
        for pf in (0, 1):
            KW = 2 if pf == 0 else 1
            nj = (Fi + 1 - pf) // 2
            wsel = wcls[w_off:w_off + KH * KW * C * C].view(C, KH * KW * C)
            w_off += KH * KW * C * C
            a_off = (pt * (Fo + 2) + pf) * C           # odd classes start one padded row / column further
            o_off = (pt * Fi + pf) * C
            ap = bp = None
            korder = 0
            if dyp_pl is not None:
                korder = int(CONV_KORDER and C % 32 == 0 and B * ni * nj >= CONV_KORDER_MIN_ROWS)
                bp = _planes.of(_korder_cols(wsel, KH * KW, C) if korder else wsel, force=True)
                if bp is not None:             # the class's window into the padded dy: the same planes, a_off elements in
                    ap = _planes.Planes(dyp_pl.t, dyp_pl.ptr + 2 * a_off, dyp_pl.stride, C, dyp_pl.rows, C)
                else:
                    korder = 0
            assert flat_in is not None or (korder and ap is not None)
            hip.gemm(None if flat_in is None else flat_in[a_off:], wsel, flat_out[o_off:], B * ni * nj, C, KH * KW * C, lda=0, ldb=KH * KW * C, ldc=C,
                     act=ACT_RELU, actgrad_in=flat_y[o_off:], ld_aux=C, conv=(To + 2, Fo + 2, ni, nj, C, KW, 1), conv_gather=hip.GATHER_A,
                     conv_kh=KH, scatter=(Ti, Fi, ni, nj, 2), a_planes=ap, b_planes=bp, conv_korder=korder, actgrad_bf16=yin_planes is not None)
    return dyin


class ConvSubsamplingFn(torch.autograd.Function):
    """subsampling.py:110-116 (Conv2dSubsampling4) / :176-182 (6) / :248-253 (8) + embedding.py:44-60/75-88:
    Conv2d(1,C,3,2)+ReLU -> n x [Conv2d(C,C,k,s)+ReLU] -> channel-major flatten -> Linear -> x*sqrt(d) (+pe);
    `geoms` = ((k, s), ...) of the n C->C stages ((3,2) for 1/4, (5,3) for 1/6, (3,2),(3,2) for 1/8).
    Activations are kept NHWC so every C->C conv is an implicit GEMM (im2col gather in the operand load) and the
    flatten before the Linear is free; the checkpoint's OIHW / channel-major weights are re-laid-out on the fly (tiny).
    Arguments after the fixed ones: (w_k, b_k) of the n C->C conv stages, in forward order."""

    @staticmethod
    def forward(ctx, x, w1, b1, wl, bl, pe, xscale, geoms, *stage_params):
        x = _chk(x, "subsampling input")
        B, T, Fd = x.shape
        C = w1.shape[0]
        d = wl.shape[0]
        n = len(stage_params) // 2
        assert len(geoms) == n
        dims = [((T - 3) // 2 + 1, (Fd - 3) // 2 + 1)]
        T1, F1 = dims[0]
        # The conv1 activation as bf16 planes ONLY (no fp32 copy: 636 MB less written here and 318 MB less read by the input
        # gradient's ReLU mask at config 2) when every reader is sure to take the pre-split kernel: the first C -> C stage is the
        # 3x3 stride-2 one in channel-chunk order forward and in all four parity classes backward (oe_gemm_f32 fails loudly
        # otherwise), and its weight gradient runs on planes too.
        planes_only = False
        if _planes.available() and n > 0 and CONV1_PLANES_ONLY and geoms[0] == (3, 2) and CONV_DGRAD_IMPLICIT and CONV_WGRAD_PLANES \
                and CONV_KORDER and C % 32 == 0 and C % 128 == 0 and T1 >= 3 and F1 >= 3:
            To0, Fo0 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
            # (the weight gradient reduces over the B To0 Fo0 output positions and has no other kernel to fall back to: any count
            # goes since planes buffers carry zero pad rows to the next multiple of 16, oe_gemm_args.planes_k_padded)
            planes_only = (B * To0 * Fo0 >= CONV_KORDER_MIN_ROWS and B * (T1 // 2) * (F1 // 2) >= CONV_KORDER_MIN_ROWS)
        if planes_only:
            y = None
            y_pl = _planes.alloc(B * T1 * F1, C, x.device)
            hip.call("oe_conv1_fwd_pl", x, w1, b1, B, T, Fd, C, None, y_pl.t, y_pl.stride)
            acts = [y_pl.t]
        else:
            y = _new(B, T1, F1, C, like=x)
            y_pl = _planes.new_output(y.view(-1, C)) if (_planes.available() and n > 0) else None     # conv2 reads it pre-split
            if y_pl is not None:
                hip.call("oe_conv1_fwd_pl", x, w1, b1, B, T, Fd, C, y, y_pl.t, y_pl.stride)
            else:
                hip.call("oe_conv1_fwd", x, w1, b1, B, T, Fd, C, y)
            acts = [y]
        wgs = []
        for k in range(n):
            wk, bk = stage_params[2 * k], stage_params[2 * k + 1]
            Ti, Fi = dims[-1]
            ks, st = geoms[k]
            To, Fo = (Ti - ks) // st + 1, (Fi - ks) // st + 1
            assert Ti >= ks and Fi >= ks and To > 0 and Fo > 0, "input too short for the subsampling stack"
            kk = ks * ks
            wg = _new(C, kk * C, like=x)                            # [co][kh][kw][ci]
            hip.call("oe_swap_last2", wk, C, C, kk, wg, 0)
            yo = _new(B * To * Fo, C, like=x)
            ap = bp = cp = None
            korder = 0
            if _planes.available():            # pre-split mode: the NHWC activation and the re-laid weights as bf16 planes
                ap = y_pl if (k == 0 and planes_only) else _planes.of(acts[-1].view(-1, C))
                korder = int(ap is not None and CONV_KORDER and C % 32 == 0 and B * To * Fo >= CONV_KORDER_MIN_ROWS)
                bp = _planes.of(_korder_cols(wg, kk, C) if korder else wg, force=True) if ap is not None else None
                if bp is None:
                    ap, korder = None, 0
                cp = _planes.new_output(yo) if (k + 1 < n or _planes.split_activations()) else None    # a next conv stage reads it as an operand
            assert not (k == 0 and planes_only) or (korder and bp is not None)
            hip.gemm(None if (k == 0 and planes_only) else acts[-1], wg, yo, B * To * Fo, C, kk * C, lda=0, ldb=kk * C, ldc=C, bias=bk, act=ACT_RELU,
                     conv=(Ti, Fi, To, Fo, C, ks, st), conv_gather=hip.GATHER_A, a_planes=ap, b_planes=bp, c_planes=cp, conv_korder=korder)
            dims.append((To, Fo))
            acts.append(yo.view(B, To, Fo, C))
            wgs.append(wg)
        TL, FL = dims[-1]
        assert wl.shape[1] == C * FL, "Linear input size does not match the conv output"
        wlg = _new(d, FL * C, like=x)                               # columns reordered to (f, c)
        hip.call("oe_swap_last2", wl, d, C, FL, wlg, 0)
        ylv = acts[-1].view(B * TL, FL * C)
        pe2 = None if pe is None else _chk(pe, "pe").reshape(-1, d)[:TL]
        out = gemm_nt(ylv, wlg, bl, beta=xscale, residual=pe2, ldr=0 if pe2 is None else d, res_row_mod=0 if pe2 is None else TL)
        ctx.save_for_backward(x, wlg, *acts, *wgs)
        ctx.params = (w1, b1, wl, bl, stage_params)
        ctx.cfg = (B, T, Fd, C, d, xscale, n, dims, geoms)
        ctx.planes_only = planes_only
        return out.view(B, TL, d)

    @staticmethod
    def backward(ctx, dout):
        B, T, Fd, C, d, xscale, n, dims, geoms = ctx.cfg
        x, wlg = ctx.saved_tensors[0], ctx.saved_tensors[1]
        acts = ctx.saved_tensors[2:2 + n + 1]
        wgs = ctx.saved_tensors[2 + n + 1:]
        w1, b1, wl, bl, stage_params = ctx.params
        TL, FL = dims[-1]
        do2 = dout.contiguous().view(B * TL, d)
        ylv = acts[-1].view(B * TL, FL * C)
        (dbl_buf, dbl) = grad_sink(bl)
        dwlg = gemm_tn(do2, ylv, alpha=xscale, bias_out=dbl_buf)
        dwl = _sink_swapped(wl, dwlg, d, FL, C)
        # gradient w.r.t. the last conv's pre-activation: the Linear's dgrad with the ReLU mask fused
        dy = gemm_nn(do2, wlg, alpha=xscale, act=ACT_RELU, actgrad_in=ylv, ld_aux=FL * C,
                     out_planes="always" if CONV_WGRAD_PLANES else True).view(B * TL * FL, C)
        fused = hip.GEMM_PRECISION != 0
        stage_grads = [None] * (2 * n)
        for k in range(n - 1, -1, -1):
            wk, bk = stage_params[2 * k], stage_params[2 * k + 1]
            (Ti, Fi), (To, Fo) = dims[k], dims[k + 1]
            ks, st = geoms[k]
            kk = ks * ks
            Mo = B * To * Fo
            conv = (Ti, Fi, To, Fo, C, ks, st)
            yin = acts[k]
            dwg = _new(C, kk * C, like=do2, zero=True)
            (dbk_buf, dbk) = grad_sink(bk)
            yin_pl = None
            if k == 0 and ctx.planes_only:          # acts[0] is the (3, rows, C) bf16 planes tensor: there is no fp32 copy
                yin_pl = _planes.Planes(yin, yin.data_ptr(), yin.shape[1] * C, C, yin.shape[1], C, kpad=True)     # (planes.alloc's buffer)
                ap, bp = _planes.of(dy, make=True, force=True), yin_pl
                assert ap is not None, "the conv stage's output gradient must be dense, 16-byte aligned fp32"
                yin = None
            else:
                ap, bp = wgrad_planes(dy, yin.view(-1, C), always=CONV_WGRAD_PLANES)
            hip.gemm(dy, yin, dwg, C, kk * C, Mo, lda=C, ldb=0, ldc=kk * C, a_kmajor=True, b_kmajor=True,
                     split_k=_split_k(C, kk * C, Mo), atomic_out=True, conv=conv, conv_gather=hip.GATHER_B,
                     a_colsum=dbk_buf if fused else None, a_planes=ap, b_planes=bp)
            if not fused:
                colsum(dy, out=dbk_buf)
            stage_grads[2 * k], stage_grads[2 * k + 1] = _sink_swapped(wk, dwg, C, kk, C), dbk
            if yin_pl is not None:
                dyin = _conv_dgrad_k3s2(dy, wk, None, B, Ti, Fi, To, Fo, C, yin_planes=yin_pl)
            elif (ks, st) == (3, 2) and CONV_DGRAD_IMPLICIT:
                dyin = _conv_dgrad_k3s2(dy, wk, yin, B, Ti, Fi, To, Fo, C)
            else:
                dcol = gemm_nn(dy, wgs[k])                           # (Mo, k*k*C)
                dyin = torch.empty_like(yin)
                hip.call("oe_col2im_relu_ks", dcol, yin, B, Ti, Fi, C, ks, st, dyin)   # col2im + the ReLU mask of this stage's input
                del dcol
            dy = dyin.view(B * Ti * Fi, C)
        (dw1, rw1), (db1, rb1) = grad_sink(w1), grad_sink(b1)
        hip.call("oe_conv1_wgrad", x, dy.view(B, dims[0][0], dims[0][1], C), B, T, Fd, C, dw1, db1)
        return (None, rw1, rb1, dwl, dbl, None, None, None) + tuple(stage_grads)


def subsampling4(x, w1, b1, w2, b2, wl, bl, pe, xscale):
    return ConvSubsamplingFn.apply(x, w1, b1, wl, bl, pe, xscale, ((3, 2),), w2, b2)


def subsampling6(x, w1, b1, w2, b2, wl, bl, pe, xscale):
    return ConvSubsamplingFn.apply(x, w1, b1, wl, bl, pe, xscale, ((5, 3),), w2, b2)


def subsampling8(x, w1, b1, w2, b2, w3, b3, wl, bl, pe, xscale):
    return ConvSubsamplingFn.apply(x, w1, b1, wl, bl, pe, xscale, ((3, 2), (3, 2)), w2, b2, w3, b3)


# --------------------------------------------------------------------------- #
# heads
# --------------------------------------------------------------------------- #
def _vpad(V: int) -> int:
    return (V + 3) // 4 * 4


class CTCHeadFn(torch.autograd.Function):
    """ctc.py:38-45: ctc_lo -> log_softmax -> CTCLoss(sum, zero_infinity) / B, fused with its
    gradient; the logits buffer is overwritten by d loss/d logits (never read again)."""

    @staticmethod
    def forward(ctx, hs, w, b, hlens, ys, ylens, length_normalized=False):
        hs = _chk(hs, "ctc input")
        B, T, d = hs.shape
        V = w.shape[0]
        Vp = _vpad(V)
        hs2 = hs.view(-1, d)
        logits = _new(B * T, Vp, like=hs)
        hip.gemm(hs2, w, logits, B * T, V, d, lda=d, ldb=d, ldc=Vp, bias=b)
        if Vp != V:
            logits[:, V:].zero_()
        Lmax = max(int(ys.shape[1]), 1)
        ys32 = ys.to(torch.int32).contiguous()
        if ys32.shape[1] == 0:
            ys32 = torch.zeros(B, 1, dtype=torch.int32, device=hs.device)
        hl32, yl32 = hlens.to(torch.int32).contiguous(), ylens.to(torch.int32).contiguous()
        ws = _new(hip.lib().oe_ctc_workspace_floats(B, T, Lmax), like=hs)
        nll = _new(B, like=hs)
        tot = _new(1, like=hs)
        # reduction 'sum' then / B (ctc.py:43-44); 'mean' = mean_b(nll_b / max(len_b, 1)), then / B as well
        uw = (1.0 / yl32.clamp(min=1).to(torch.float32)) if length_normalized else None
        denom = float(B * B) if length_normalized else float(B)
        hip.call("oe_ctc_loss_fused", logits, Vp, B, T, V, hl32, ys32, Lmax, yl32, 1.0 / denom, uw, nll, tot, logits, ws)
        ctx.save_for_backward(hs2, w, logits)
        ctx.shape = (B, T, d, V)
        ctx.bias_ref = b
        return tot[0] / denom

    @staticmethod
    def backward(ctx, g):
        stamp("bwd: ctc head backward starts")
        hs2, w, dlogits = ctx.saved_tensors
        B, T, d, V = ctx.shape
        g = g.contiguous().view(1)
        dl = dlogits[:, :V]
        dhs = gemm_nn_deep(dl, w, alpha_dev=g).view(B, T, d)
        dw, db = wgrad_bias(w, ctx.bias_ref, dl, hs2, alpha_dev=g)
        return dhs, dw, db, None, None, None, None


def ctc_head(hs, w, b, hlens, ys, ylens, length_normalized=False):
    return CTCHeadFn.apply(hs, w, b, hlens, ys, ylens, length_normalized)


class LSMHeadFn(torch.autograd.Function):
    """decoder.py:192 (output_layer) + label_smoothing_loss.py:58-91 + common.py:135-157, fused.
    Returns (loss, n_correct, n_valid)."""

    @staticmethod
    def forward(ctx, x, w, b, target, smoothing, normalize_length, ignore_id):
        x = _chk(x, "decoder output")
        B = x.shape[0]
        d = x.shape[-1]
        V = w.shape[0]
        Vp = _vpad(V)
        x2 = x.reshape(-1, d)
        rows = x2.shape[0]
        logits = _new(rows, Vp, like=x)
        hip.gemm(x2, w, logits, rows, V, d, lda=d, ldb=d, ldc=Vp, bias=b)
        tgt = target.reshape(-1).to(torch.int64).contiguous()
        ws = torch.empty(hip.lib().oe_lsm_workspace_bytes(rows), dtype=torch.uint8, device=x.device)
        out3 = _new(3, like=x)
        hip.call("oe_lsm_loss_fused", logits, Vp, rows, V, tgt, ignore_id, smoothing, int(normalize_length), float(B), 1.0,
                 1, out3, ws)
        ctx.save_for_backward(x2, w, logits)
        ctx.shape = (x.shape, V)
        ctx.bias_ref = b
        ctx.mark_non_differentiable(out3[1], out3[2])
        return out3[0], out3[1], out3[2]

    @staticmethod
    def backward(ctx, g, _g1, _g2):
        stamp("bwd: a decoder's backward starts")
        x2, w, dlogits = ctx.saved_tensors
        shape, V = ctx.shape
        g = g.contiguous().view(1)
        dl = dlogits[:, :V]
        dx = gemm_nn_deep(dl, w, alpha_dev=g).view(shape)
        dw, db = wgrad_bias(w, ctx.bias_ref, dl, x2, alpha_dev=g)
        return dx, dw, db, None, None, None, None


def lsm_head(x, w, b, target, smoothing, normalize_length=False, ignore_id=-1):
    return LSMHeadFn.apply(x, w, b, target, smoothing, normalize_length, ignore_id)


class LSMLossFn(torch.autograd.Function):
    """label_smoothing_loss.py:58-91 on materialised logits (module API)."""

    @staticmethod
    def forward(ctx, x, target, smoothing, normalize_length, ignore_id):
        x = _chk(x, "logits")
        B, V = x.shape[0], x.shape[-1]
        Vp = _vpad(V)
        rows = x.numel() // V
        buf = _new(rows, Vp, like=x, zero=(Vp != V))
        buf[:, :V].copy_(x.reshape(rows, V))
        tgt = target.reshape(-1).to(torch.int64).contiguous()
        ws = torch.empty(hip.lib().oe_lsm_workspace_bytes(rows), dtype=torch.uint8, device=x.device)
        out3 = _new(3, like=x)
        hip.call("oe_lsm_loss_fused", buf, Vp, rows, V, tgt, ignore_id, smoothing, int(normalize_length), float(B), 1.0, 1,
                 out3, ws)
        ctx.save_for_backward(buf)
        ctx.shape = (x.shape, V)
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        (buf,) = ctx.saved_tensors
        shape, V = ctx.shape
        src = buf[:, :V].contiguous()
        out = torch.empty_like(src)
        hip.call("oe_axpby", src, None, out.numel(), 1.0, 0.0, g.contiguous().view(1), out)
        return out.view(shape), None, None, None, None


class EmbedFn(torch.autograd.Function):
    """decoder.py:144-147,186: Embedding -> x*sqrt(d) + pe."""

    @staticmethod
    def forward(ctx, tokens, table, pe, xscale):
        B, L = tokens.shape
        V, d = table.shape
        tok = tokens.to(torch.int64).contiguous()
        out = _new(B, L, d, like=table)
        hip.call("oe_embed_fwd", tok, table, pe, B * L, L, d, V, xscale, out)
        ctx.save_for_backward(tok)
        ctx.table_ref = table
        ctx.cfg = (V, d, xscale)
        return out

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        V, d, xscale = ctx.cfg
        dout = dout.contiguous()
        dt, rt = grad_sink(ctx.table_ref)
        hip.call("oe_embed_bwd", tok, dout, tok.numel(), d, V, xscale, dt)
        stamp("bwd: a decoder's backward done")
        return None, rt, None, None


def embed(tokens, table, pe, xscale):
    return EmbedFn.apply(tokens, table, pe, xscale)


# --------------------------------------------------------------------------- #
# inference helpers
# --------------------------------------------------------------------------- #
def ctc_greedy(logits, ldv, B, T, V, hlens, eos):
    fb = torch.empty(B, T, dtype=torch.int32, device=logits.device)
    ot = torch.empty(B, T, dtype=torch.int32, device=logits.device)
    ol = torch.empty(B, dtype=torch.int32, device=logits.device)
    hl = hlens.to(torch.int32).contiguous()
    hip.call("oe_ctc_greedy", logits, ldv, B, T, V, hl, eos, fb, ot, ol)
    return ot, ol


def topk_rows(x, k: int, log_softmax: bool = False):
    """`x.topk(k)` over the last dim - of log_softmax(x) when asked - in one kernel (asr_model.py:251, 258, 358).
    Returns (values float32, indices int64), sorted descending; ties go to the lowest index."""
    x = _chk(x, "topk_rows")
    V = x.shape[-1]
    rows = x.numel() // V
    vals = torch.empty(*x.shape[:-1], k, dtype=torch.float32, device=x.device)
    idx = torch.empty(*x.shape[:-1], k, dtype=torch.int64, device=x.device)
    hip.call("oe_topk_rows", x, rows, V, int(k), int(bool(log_softmax)), vals, idx)
    return vals, idx


def logprob_gather(logits, idx, also: int = None):
    """log_softmax(logits)[..., idx] (+ log_softmax(logits)[..., also] for one fixed column) without materialising the
    log-probabilities: logits (..., V) float32, idx (...) int64 (out-of-range entries give 0).  Returns tensors shaped like idx."""
    x = _chk(logits, "logprob_gather")
    V = x.shape[-1]
    rows = x.numel() // V
    idx = idx.to(torch.int64).contiguous()
    assert idx.numel() == rows
    out_a = torch.empty(idx.shape, dtype=torch.float32, device=x.device)
    out_b = torch.empty(idx.shape, dtype=torch.float32, device=x.device) if also is not None else None
    hip.call("oe_logprob_gather", x, rows, V, idx, 0 if also is None else int(also), out_a, out_b)
    return out_a if also is None else (out_a, out_b)


def log_softmax_rows(x):
    """log_softmax over the last dim (ctc.py:56-64, asr_model.py:484-488) on device."""
    x = _chk(x, "log_softmax")
    out = torch.empty_like(x)
    V = x.shape[-1]
    hip.call("oe_log_softmax", x, x.numel() // V, V, out)
    return out
