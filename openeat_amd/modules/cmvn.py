"""GlobalCMVN (/root/reference/openeat/modules/cmvn.py:18-46): (x - mean) * istd with
``mean``/``istd`` registered buffers (they are part of the checkpoint)."""
import torch

from openeat_amd import ops


class GlobalCMVN(torch.nn.Module):
    def __init__(self, mean: torch.Tensor, istd: torch.Tensor, norm_var: bool = True):
        super().__init__()
        assert mean.shape == istd.shape
        self.norm_var = norm_var
        self.register_buffer("mean", mean)
        self.register_buffer("istd", istd)

    def forward(self, x: torch.Tensor):
        istd = self.istd if self.norm_var else torch.ones_like(self.istd)
        return ops.global_cmvn(x, self.mean, istd)
