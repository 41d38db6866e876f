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


class ParamArena:
    ALIGN = 4  # floats (16 bytes)

    def __init__(self, model: torch.nn.Module):
        from openeat_amd.modules.attention import MultiHeadedAttention
        params: List[torch.nn.Parameter] = []
        seen = set()

        def add(p):
            if p is not None and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)

        # attention modules first lay out [Wq Wk Wv][bq bk bv] contiguously
        for mod in model.modules():
            if isinstance(mod, MultiHeadedAttention):
                for lin in (mod.linear_q, mod.linear_k, mod.linear_v):
                    add(lin.weight)
                for lin in (mod.linear_q, mod.linear_k, mod.linear_v):
                    add(lin.bias)
        for p in model.parameters():
            add(p)
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
        self.by_ptr: Dict[int, torch.Tensor] = {}
        self.enabled = True
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
