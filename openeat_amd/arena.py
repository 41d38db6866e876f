"""Flat parameter / gradient arenas.

All trainable parameters of a model are re-homed into ONE contiguous fp32
buffer (and their gradients into a second one of the same layout):

* the weight-gradient kernels accumulate straight into the gradient arena
  (``ops`` looks the target up by the parameter's device address), so there is
  no per-parameter gradient tensor, no autograd accumulation kernel and
  ``zero_grad`` is a single memset;
* the optimiser (global norm + clip + Adam) and the data-parallel all-reduce
  each touch one flat buffer instead of ~620 tensors;
* q/k/v projection weights of every attention module are laid out back to
  back, so the fused QKV GEMM reads them in place (no concatenation copy).

``nn.Parameter`` objects, names and shapes are untouched: ``state_dict()`` and
checkpoints keep the reference's layout.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

_ACTIVE: Optional["ParamArena"] = None


def active() -> Optional["ParamArena"]:
    return _ACTIVE


def grad_target(param: torch.Tensor) -> Optional[torch.Tensor]:
    """Gradient-arena view for a parameter tensor (by device address), or None."""
    a = _ACTIVE
    if a is None or not a.enabled:
        return None
    return a.by_ptr.get(param.data_ptr())


def arena_units(model: torch.nn.Module):
    """(unit name, module) in forward order for the ASR model; any other module is one unit."""
    enc = getattr(model, "encoder", None)
    if enc is None or not hasattr(enc, "encoders") or not hasattr(enc, "embed"):
        return [("all", model)]
    units = []
    if getattr(enc, "global_cmvn", None) is not None:
        units.append(("embed", enc.global_cmvn))
    units.append(("embed", enc.embed))
    for i, layer in enumerate(enc.encoders):
        units.append((f"enc{i}", layer))
    units.append(("enc_norm", enc.after_norm))
    for name in ("ctc", "decoder"):
        if getattr(model, name, None) is not None:
            units.append(("heads", getattr(model, name)))
    return units


class ParamArena:
    ALIGN = 8  # floats (32 bytes): a parameter's bf16 planes (planes.py) then start on 16-byte boundaries too

    def __init__(self, model: torch.nn.Module):
        from openeat_amd.modules.attention import MultiHeadedAttention
        params: List[torch.nn.Parameter] = []
        seen = set()
        self.unit_of: List[str] = []          # unit name of every parameter, in arena order

        def add(p, unit):
            if p is not None and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
                self.unit_of.append(unit)

        # The arena is laid out in units that follow the forward order (input layer, encoder layer 0..n-1, final norm,
        # heads): backward finishes them in reverse, so "everything from unit u on" is a contiguous tail that can go
        # to the gradient all-reduce while the earlier units are still being differentiated (ddp.GradAllReduce).
        # Inside a unit every attention module first lays out [Wq Wk Wv][bq bk bv] contiguously (fused QKV GEMM).
        for unit, mod in arena_units(model):
            for m in mod.modules():
                if isinstance(m, MultiHeadedAttention):
                    for lin in (m.linear_q, m.linear_k, m.linear_v):
                        add(lin.weight, unit)
                    for lin in (m.linear_q, m.linear_k, m.linear_v):
                        add(lin.bias, unit)
            for p_ in mod.parameters():
                add(p_, unit)
        for p_ in model.parameters():           # anything the unit walk did not reach
            add(p_, "rest")
        assert params, "model has no trainable parameters"
        dev = params[0].device
        assert dev.type == "cuda", "move the model to the GPU before building the arena"
        offs, total = [], 0
        for p in params:
            assert p.dtype == torch.float32 and p.device == dev
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = total
        self.flat = torch.zeros(total, device=dev)
        self.grad = torch.zeros(total, device=dev)
        self.params = params
        self.unit_start: Dict[str, int] = {}      # first float of each unit (units are contiguous, in forward order)
        for u, o in zip(self.unit_of, offs):
            self.unit_start.setdefault(u, o)
        self.by_ptr: Dict[int, torch.Tensor] = {}
        self.enabled = True
        # the weights as three bf16 planes (precision 6 GEMM operands, planes.py): allocated on first use, split again by the
        # first reader of every pass (step_planes) - after an optimizer step, a graph replay, a torch write
        self.planes: Optional[torch.Tensor] = None
        self.planes_stride = total
        self._planes_version = -1
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                g = self.grad[o:o + p.numel()].view(p.shape)
                p.grad = g
                self.by_ptr[view.data_ptr()] = g

    def activate(self):
        global _ACTIVE
        _ACTIVE = self
        return self

    def refresh_planes(self):
        """fp32 arena -> bf16 planes (one launch over all parameters, 49 us for 31 M; capturable)."""
        from . import hip, planes as _planes
        if not _planes.weights_presplit():
            return
        self.alloc_planes()
        hip.call("oe_split_planes", self.flat, self.numel, 1, self.numel, self.planes, self.numel, self.planes_stride)
        self._planes_version = self.flat._version
        self._planes_fresh = True

    _planes_fresh = False

    def alloc_planes(self):
        """The planes buffer outlives every graph that reads it: never allocate it from a capture's private pool
        (planes.capture_scope allocates it ahead of the capture)."""
        if self.planes is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ParamArena: the weights' bf16 planes must be allocated before a graph capture "
                                   "(wrap the capture in openeat_amd.planes.capture_scope())")
            self.planes = torch.empty(3, self.numel, dtype=torch.bfloat16, device=self.flat.device)

    def mark_step(self):
        """The weights may have moved without torch noticing - a raw kernel wrote them (FusedAdam.step calls this), a graph
        replay ran its Adam without any Python (TrainEngine.replay), or a new forward pass starts and nobody can vouch for what
        happened since the last one (planes.new_pass: ASRModel.forward, the decode entry points, LanguageModel.forward) - so the
        next reader splits again, under EVERY policy, and that launch is part of whatever graph the pass is captured into.
        planes.capture_scope() calls this on exit: a split that was only RECORDED has refreshed nothing."""
        self._planes_fresh = False
        from . import planes as _planes
        _planes.bump_generation()

    def step_planes(self):
        """Planes valid for the current pass: split on the pass's first use, reused until mark_step() or a torch write."""
        if not self._planes_fresh or self.planes is None or self._planes_version != self.flat._version:
            self.refresh_planes()

    ensure_planes = step_planes      # (round 3 had a second form that trusted the tensor version alone: stale after any
    #                                   optimizer step taken outside TrainEngine)

    def deactivate(self):
        global _ACTIVE
        if _ACTIVE is self:
            _ACTIVE = None

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:               # someone may have set p.grad = None (optimizer.zero_grad default)
            if p.grad is None or p.grad.data_ptr() != self.by_ptr[p.data_ptr()].data_ptr():
                p.grad = self.by_ptr[p.data_ptr()]

    def adjacent(self, *tensors) -> bool:
        """True if the tensors sit back to back in the weight arena (in this order)."""
        ptr = tensors[0].data_ptr()
        for t in tensors:
            if t.data_ptr() != ptr or t.data_ptr() not in self.by_ptr:
                return False
            ptr += t.numel() * 4
        return True
