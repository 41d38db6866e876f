"""Pre-split GEMM operands (precision 6).

A matrix product in the reference-precision mode is six bf16 MFMA products of three exact bf16 pieces per operand
(csrc/oe_common.h).  Splitting fp32 -> pieces inside every consuming GEMM block costs more vector instructions than the
matrix work itself, so tensors that feed GEMMs carry a second copy, three bf16 PLANES p0 + p1 + p2 = x, written once:

* activations / gradients: by the kernel that produces them (LayerNorm forward / backward, GEMM epilogues:
  ``c_planes``) or by one pass of ``oe_split_planes``;
* weights: the whole parameter arena is split by the first reader of every pass (``ParamArena.step_planes``); weights
  outside an arena are split on first use and cached by (address, version) under no_grad.

``csrc/gemm_pl.hip`` then moves tiles global -> LDS by LDS-DMA and its loop is DMA issue, LDS reads and MFMAs only.
Everything here is optional: an operand without planes sends the GEMM to the kernels that split in the loop.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import hip

# Where operands are pre-split (measured on MI355X, tools/pl_bench.py + bench.py, profiles/r03_experiments.md):
#   "conv" (default): the subsampling front end only - the conv1 output feeds conv2's forward gather 9/4 times over and its
#                     weight gradient again, the padded dy feeds four parity-class input gradients: one split pass pays;
#   "all":            every Linear of the encoder too (LayerNorm / GEMM epilogues write planes).  The GEMMs gain 10-25 %
#                     each, but a feed-forward's (rows, ff) intermediate costs 1.5 x its fp32 bytes again as planes: a wash
#                     at config 2 (the step as a whole measured slower), so not the default;
#   "ln":             conv + the operands that come pre-split for (almost) nothing: LayerNorm outputs (forward: x of W1 / QKV /
#                     pointwise-1; backward: the block-input gradient g that feeds the previous block's input-gradient
#                     GEMMs) against the arena's weight planes.  No split passes, no planes of GEMM outputs;
#   "0":              never.
POLICY = os.environ.get("OE_PLANES", "conv")
# below this many elements a tensor is not worth a split pass of its own (the decoders' 992-row activations)
MIN_SPLIT_ELEMS = int(os.environ.get("OE_PLANES_MIN", str(1 << 19)))


_DEBUG = bool(os.environ.get("OE_PLANES_DEBUG"))


def available() -> bool:
    """Pre-split operands exist at all in the current arithmetic (explicit oe_split_planes users, the conv front end)."""
    return POLICY != "0" and hip.GEMM_PRECISION == 6


def active() -> bool:
    """The general Linear / LayerNorm plumbing of ops.py uses pre-split operands."""
    return POLICY in ("all", "ln") and hip.GEMM_PRECISION == 6


# The weights of the parameter arena as planes under EVERY policy (one split pass over the arena per optimizer step): csrc/gemm_hyb.hip
# takes a Linear's weight operand pre-split and splits only the activation on the fragment - half the ring kernel's vector work.
WEIGHT_PLANES = os.environ.get("OE_WEIGHT_PLANES", "1") == "1"
# ... from this many activation rows on (measured, same-box A/B of the whole step: 25472 rows 51.3 -> 49.8 ms/step, 7936 rows
# 18.51 -> 18.67: there the launches are bound by what they write and by cold operands, and the arena's split pass costs 50 us)
HYB_MIN_ROWS = int(os.environ.get("OE_HYB_MIN_ROWS", "12288"))


def weights_presplit() -> bool:
    """The arena keeps bf16 planes of the weights (ParamArena.step_planes: split by the first reader of every pass)."""
    return hip.GEMM_PRECISION == 6 and (POLICY in ("all", "ln") or (WEIGHT_PLANES and POLICY != "0"))


def arena_weight(w2d: torch.Tensor) -> Optional[Planes]:
    """Planes of a weight matrix that lives in the parameter arena (a window of the arena's planes), else None - never a split pass."""
    if not weights_presplit() or w2d.dim() != 2 or w2d.stride(1) != 1 or w2d.stride(0) % 8 or w2d.shape[1] % 8:
        return None
    from . import arena as _arena
    a = _arena.active()
    if a is None:
        return None
    ptr = w2d.data_ptr()
    off = (ptr - a.flat.data_ptr()) // 4
    if not (0 <= off < a.numel) or ptr % 32:
        return None
    a.step_planes()                      # split on this pass's first use
    return Planes(a.planes, a.planes.data_ptr() + 2 * off, a.planes_stride, w2d.stride(0), w2d.shape[0], w2d.shape[1])


def split_activations() -> bool:
    """Activations whose producer wrote no planes get a split pass of their own / GEMM epilogues write planes of their
    outputs ("all"); under "ln" only what LayerNorm forward / backward wrote on its way (and the weights) is pre-split."""
    return POLICY == "all" and hip.GEMM_PRECISION == 6


class Planes:
    """Three bf16 planes of a (rows, cols) fp32 matrix: tensor `t` of shape (3, rows, cols) or a view into a larger buffer."""
    __slots__ = ("t", "ptr", "stride", "ld", "rows", "cols", "kpad")

    def __init__(self, t: torch.Tensor, ptr: int, stride: int, ld: int, rows: int, cols: int, kpad: bool = False):
        self.t, self.ptr, self.stride, self.ld, self.rows, self.cols = t, ptr, stride, ld, rows, cols
        self.kpad = kpad            # the buffer holds ZERO rows from `rows` up to the next multiple of 16 (oe_gemm_args.planes_k_padded)


def alloc(rows: int, cols: int, device) -> Planes:
    """Planes buffer of a (rows, cols) matrix.  The row count of the ALLOCATION is rounded up to a multiple of 16 and the pad rows
    are zero: as the k-major operand of a weight gradient the matrix can then be reduced over whole K-tiles whatever `rows` is
    (ragged batches: rows = the step's valid positions; without it the pre-split kernel declines and configs[4]'s conv2 weight
    gradient ran at 18 TFLOP/s)."""
    rp = (rows + 15) // 16 * 16
    t = torch.empty(3, rp, cols, dtype=torch.bfloat16, device=device)
    if rp != rows:
        t[:, rows:].zero_()
    return Planes(t, t.data_ptr(), rp * cols, cols, rows, cols, kpad=True)


# ---- activations: registry by device address ----------------------------------------------------------------------------
# An entry keeps its source tensor alive, so the address cannot be handed to another tensor while the entry exists; the
# registry is emptied when a new step starts (ops.predrop_clear's call sites).  Planes that backward needs are kept by the
# autograd functions themselves (ctx), not looked up again.
import contextlib
from collections import OrderedDict

_REG: "OrderedDict[int, tuple]" = OrderedDict()
_NO_GRAD_KEEP = 64     # inference: a tensor's planes are consumed right after they are made - keep a short FIFO only
_CAPTURE_DEPTH = 0     # > 0 inside capture_scope(): no planes made outside the capture may be handed to a captured launch
ISOLATE_CAPTURES = True   # tools/capture_probe.py switches this off to show what round 3's red test was


def clear():
    _REG.clear()
    new_pass()


def clear_all():
    """Everything keyed by a device address: the activation registry AND the cached weight splits (test teardown; a model
    that goes away must not leave planes of its weights behind under an address the next model may be given)."""
    _WCACHE.clear()
    clear()
    try:
        from . import ops as _ops
        _ops.ffn_packs_clear()
    except ImportError:                  # (planes is imported by ops: only a half-initialised package gets here)
        pass


@contextlib.contextmanager
def capture_scope():
    """Around the capture of a HIP graph.  A capture executes nothing and bakes device addresses into its launches, so it
    must neither READ planes that eager code owns nor leave its own behind for eager code to find:

    * the registry is a FIFO under no_grad (_NO_GRAD_KEEP) and is emptied at every step: an entry that a captured GEMM
      picked up - e.g. the planes of the position table slice, whose address is the same at every call - is evicted a few
      registrations later, its memory goes back to the allocator, and the replay reads whatever was put there since
      (GPUTEST_r03: first replay of the decode stage-1 graph under OE_PLANES=all returned empty hypotheses);
    * a cached weight split (_WCACHE) handed to a capture would be read by every replay even after the weights changed;
    * an entry registered DURING the capture points into the graph's private pool, which holds nothing before the first
      replay: an eager reader would multiply garbage.

    Inside the scope the registry starts empty and weights outside an arena are split by a captured launch (the replay
    then follows the weights); on exit the capture's entries are dropped and the outer registry is back.  Planes made and
    freed INSIDE one capture need no owner: their blocks are only ever reused by later allocations of the same capture,
    i.e. by launches that the replay orders behind their last reader.  The arena's planes buffer is allocated BEFORE the
    capture so that it never lives in a graph's pool."""
    global _REG, _CAPTURE_DEPTH
    if not ISOLATE_CAPTURES:
        yield
        return
    from . import arena as _arena
    a = _arena.active()
    if a is not None and weights_presplit():
        a.alloc_planes()
    outer, _REG = _REG, OrderedDict()
    _CAPTURE_DEPTH += 1
    try:
        yield
    finally:
        _CAPTURE_DEPTH -= 1
        _REG = outer
        bump_generation()                # whatever the capture "refreshed" (weight planes, packed feed-forward weights) was only
        if a is not None:                # recorded: the next eager reader refreshes for real
            a.mark_step()


_GEN = 0


def weights_generation() -> int:
    """Changes whenever weights may have moved without torch noticing (a new forward pass announces itself, FusedAdam steps, a captured
    step is replayed): consumers of derived copies (ops' packed feed-forward weights) compare it with the value they packed at."""
    return _GEN


def bump_generation():
    global _GEN
    _GEN += 1


def new_pass():
    """A forward pass starts (training step, decode, LM scoring): the arena's weight planes are split again by their first reader
    (whatever wrote the weights since - Adam's raw kernel, load_state_dict, a broadcast - is picked up)."""
    bump_generation()
    from . import arena as _arena
    from . import ops as _ops
    _ops.pack_tables_sweep()
    a = _arena.active()
    if a is not None:
        a.mark_step()


def _dense2d(t: torch.Tensor) -> bool:
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) == t.shape[1] and t.shape[1] % 8 == 0 and t.data_ptr() % 16 == 0


def register(t2d: torch.Tensor, pl: Planes):
    _REG[t2d.data_ptr()] = (pl, t2d, t2d._version)
    if not torch.is_grad_enabled():
        while len(_REG) > _NO_GRAD_KEEP:
            _REG.popitem(last=False)


def lookup(t2d: torch.Tensor) -> Optional[Planes]:
    e = _REG.get(t2d.data_ptr())
    if e is None:
        return None
    pl, src, ver = e
    if src._version != ver or not _dense2d(t2d) or pl.rows * pl.cols != t2d.numel():
        return None
    if pl.rows != t2d.shape[0]:          # another dense 2-D view of the same memory: same planes, other row length
        r, c = t2d.shape
        # the buffer's zero pad rows serve the view too if they cover ITS rows up to the next multiple of 16
        kpad = pl.kpad and pl.stride - r * c >= ((r + 15) // 16 * 16 - r) * c
        return Planes(pl.t, pl.ptr, pl.stride, c, r, c, kpad=kpad)
    return pl


def of(t2d: torch.Tensor, make: bool = True, force: bool = False) -> Optional[Planes]:
    """Planes of a dense (rows, cols) fp32 CUDA tensor: the producer's, or (make) one oe_split_planes pass."""
    if not available() or not _dense2d(t2d):
        return None
    pl = lookup(t2d)
    if pl is not None or not make:
        return pl
    if t2d.numel() < MIN_SPLIT_ELEMS and not force:
        return None
    pl = alloc(t2d.shape[0], t2d.shape[1], t2d.device)
    if _DEBUG:
        import traceback
        print("split_planes", tuple(t2d.shape), [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]], flush=True)
    hip.call("oe_split_planes", t2d, t2d.stride(0), t2d.shape[0], t2d.shape[1], pl.t, pl.ld, pl.stride)
    register(t2d, pl)
    return pl


def new_output(out2d: torch.Tensor) -> Optional[Planes]:
    """Planes buffer for a GEMM / LayerNorm output about to be written (registered now; the producer fills it)."""
    if not available() or not _dense2d(out2d):
        return None
    pl = alloc(out2d.shape[0], out2d.shape[1], out2d.device)
    register(out2d, pl)
    return pl


# ---- weights ----------------------------------------------------------------------------------------------------------
_WCACHE = {}       # data_ptr -> (Planes, weight tensor, version)
_WCACHE_MAX = 4096


def weight(w2d: torch.Tensor) -> Optional[Planes]:
    """Planes of a weight matrix (rows, cols) with row stride `w2d.stride(0)`: a window of the arena's planes, or a cached split."""
    if not available() or w2d.dim() != 2 or w2d.stride(1) != 1 or w2d.stride(0) % 8 or w2d.shape[1] % 8:
        return None
    from . import arena as _arena
    a = _arena.active()
    ptr = w2d.data_ptr()
    if a is not None and (a.planes is not None or active()):
        off = (ptr - a.flat.data_ptr()) // 4
        if 0 <= off < a.numel and ptr % 32 == 0:
            a.step_planes()             # split on this pass's first use
            return Planes(a.planes, a.planes.data_ptr() + 2 * off, a.planes_stride, w2d.stride(0), w2d.shape[0], w2d.shape[1])
    if ptr % 16 or w2d.stride(0) != w2d.shape[1]:
        return None
    cacheable = not torch.is_grad_enabled() and not _CAPTURE_DEPTH      # capture_scope(): a captured launch splits, every replay
    e = _WCACHE.get(ptr)
    if e is not None and e[2] == w2d._version and e[0].rows == w2d.shape[0] and e[0].cols == w2d.shape[1] and cacheable:
        return e[0]
    # training outside an arena: parameters change through optimizers whose in-place updates bump the version, but raw
    # kernels do not - only trust the cache under no_grad (decode / eval), else split afresh
    pl = alloc(w2d.shape[0], w2d.shape[1], w2d.device)
    if _DEBUG:
        print("split_planes weight", tuple(w2d.shape), flush=True)
    hip.call("oe_split_planes", w2d, w2d.stride(0), w2d.shape[0], w2d.shape[1], pl.t, pl.ld, pl.stride)
    if cacheable:
        while len(_WCACHE) >= _WCACHE_MAX:           # oldest first (no capture holds a cached split: capture_scope)
            _WCACHE.pop(next(iter(_WCACHE)))
        _WCACHE[ptr] = (pl, w2d, w2d._version)
    return pl
