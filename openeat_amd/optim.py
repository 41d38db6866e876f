"""Fused clip_grad_norm_ + Adam over the flat arenas
(executor.py:58-63 + torch.optim.Adam defaults of /root/reference/openeat/bin/train.py:195).
Three kernel launches per step regardless of the number of parameters."""
import torch

from openeat_amd import hip
from openeat_amd.arena import ParamArena


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, arena: ParamArena, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: float = 0.0):
        super().__init__(arena.params, dict(lr=lr, betas=betas, eps=eps))
        self.arena = arena
        self.max_grad_norm = max_grad_norm
        dev = arena.flat.device
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.step_state = torch.zeros(2, device=dev)            # [step count, 0]
        self.lr_dev = torch.full((1,), lr, device=dev)
        self.total_norm = torch.zeros(1, device=dev)
        self._ws = torch.empty(hip.lib().oe_grad_norm_workspace_floats(), device=dev)

    def set_lr(self, lr: float):
        self.param_groups[0]["lr"] = lr

    def compute_grad_norm(self) -> torch.Tensor:
        hip.call("oe_grad_norm", self.arena.grad, self.arena.numel, self._ws, self.total_norm)
        return self.total_norm

    @torch.no_grad()
    def step(self, closure=None, lr_from_device: bool = False):
        """Global-norm clip (if max_grad_norm > 0) + Adam.  The update is skipped on device when the
        gradient norm is not finite (no host sync).  ``lr_from_device``: use ``self.lr_dev`` as is
        (the caller updates it, e.g. around a captured graph)."""
        g = self.param_groups[0]
        if not lr_from_device:
            self.lr_dev.fill_(float(g["lr"]))
        self.compute_grad_norm()
        b1, b2 = g["betas"]
        hip.call("oe_adam_step", self.arena.flat, self.arena.grad, self.exp_avg, self.exp_avg_sq, self.arena.numel,
                 self.lr_dev, 0.0, b1, b2, g["eps"], float(self.max_grad_norm), self.total_norm, self.step_state)
        self.arena.mark_step()          # the raw kernel bumps no version counter: the weights' bf16 planes are stale (any training loop)
        return None

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()
