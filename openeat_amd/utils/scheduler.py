"""WarmupLR (/root/reference/openeat/utils/scheduler.py:42-49):
lr * warmup^0.5 * min(step^-0.5, step * warmup^-1.5)."""
from typing import Union

import torch
from torch.optim.lr_scheduler import _LRScheduler


class WarmupLR(_LRScheduler):
    def __init__(self, optimizer: torch.optim.Optimizer, warmup_steps: Union[int, float] = 25000, last_epoch: int = -1):
        self.warmup_steps = warmup_steps
        super().__init__(optimizer, last_epoch)

    def __repr__(self):
        return f"{self.__class__.__name__}(warmup_steps={self.warmup_steps})"

    def factor(self, step_num: int) -> float:
        w = self.warmup_steps
        return w ** 0.5 * min(step_num ** -0.5, step_num * w ** -1.5)

    def get_lr(self):
        f = self.factor(self.last_epoch + 1)
        return [base * f for base in self.base_lrs]

    def set_step(self, step: int):
        self.last_epoch = step
