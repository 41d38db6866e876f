"""Input layers (/root/reference/openeat/modules/subsampling.py): the 1/4 conv2d subsampling of every shipped
config, the 1/6 (a 5x5 stride-3 second conv) and 1/8 (one more 3x3 stride-2 stage) variants on the same implicit-GEMM
op, and the linear (no subsampling) layer."""
from typing import Tuple

import torch

from openeat_amd import ops


class BaseSubsampling(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.right_context = 0
        self.subsampling_rate = 1

    def position_encoding(self, offset: int, size: int) -> torch.Tensor:
        """subsampling.py:19-20: forwarded to the positional-encoding object (whose class defines no such method in
        the reference, embedding.py:14-88: calling it there is an AttributeError, and so it is here)."""
        return self.pos_enc.position_encoding(offset, size)


class Conv2dSubsampling4(BaseSubsampling):
    """subsampling.py:65-116.  Parameters keep the reference's names/shapes:
    conv.0 (d,1,3,3), conv.2 (d,d,3,3), out.0 (d, d*((idim-1)//2-1)//2)."""

    def __init__(self, idim: int, odim: int, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.conv = torch.nn.Sequential(
            torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(), torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU())
        self.out = torch.nn.Sequential(torch.nn.Linear(odim * (((idim - 1) // 2 - 1) // 2), odim))
        self.pos_enc = pos_enc_class
        self.subsampling_rate = 4
        self.right_context = 6

    def forward(self, x: torch.Tensor, x_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        t_out = ((x.size(1) - 1) // 2 - 1) // 2
        pos = self.pos_enc.table(x.device, t_out)
        pe = pos if self.pos_enc.kind == "abs_pos" else None
        c0, c2, lin = self.conv[0], self.conv[2], self.out[0]
        y = ops.subsampling4(x, c0.weight, c0.bias, c2.weight, c2.bias, lin.weight, lin.bias, pe, self.pos_enc.xscale)
        return y, x_mask[:, :, :-2:2][:, :, :-2:2], pos


class Conv2dSubsampling6(BaseSubsampling):
    """subsampling.py:119-182.  Parameter names as in the reference: conv.0 (d,1,3,3), conv.2 (d,d,5,5),
    linear (d, d*(((idim-1)//2-2)//3))."""

    def __init__(self, idim: int, odim: int, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.conv = torch.nn.Sequential(
            torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(), torch.nn.Conv2d(odim, odim, 5, 3), torch.nn.ReLU())
        self.linear = torch.nn.Linear(odim * (((idim - 1) // 2 - 2) // 3), odim)
        self.pos_enc = pos_enc_class
        self.subsampling_rate = 6
        self.right_context = 14

    def forward(self, x: torch.Tensor, x_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        t_out = ((x.size(1) - 1) // 2 - 2) // 3
        pos = self.pos_enc.table(x.device, t_out)
        pe = pos if self.pos_enc.kind == "abs_pos" else None
        c0, c2 = self.conv[0], self.conv[2]
        y = ops.subsampling6(x, c0.weight, c0.bias, c2.weight, c2.bias, self.linear.weight, self.linear.bias, pe, self.pos_enc.xscale)
        return y, x_mask[:, :, :-2:2][:, :, :-4:3], pos


class Conv2dSubsampling8(BaseSubsampling):
    """subsampling.py:185-253.  Parameter names as in the reference: conv.0 (d,1,3,3), conv.2, conv.4 (d,d,3,3),
    linear (d, d*((((idim-1)//2-1)//2-1)//2))."""

    def __init__(self, idim: int, odim: int, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.conv = torch.nn.Sequential(
            torch.nn.Conv2d(1, odim, 3, 2), torch.nn.ReLU(), torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU(),
            torch.nn.Conv2d(odim, odim, 3, 2), torch.nn.ReLU())
        self.linear = torch.nn.Linear(odim * ((((idim - 1) // 2 - 1) // 2 - 1) // 2), odim)
        self.pos_enc = pos_enc_class
        self.subsampling_rate = 8
        self.right_context = 14

    def forward(self, x: torch.Tensor, x_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        t_out = (((x.size(1) - 1) // 2 - 1) // 2 - 1) // 2
        pos = self.pos_enc.table(x.device, t_out)
        pe = pos if self.pos_enc.kind == "abs_pos" else None
        c0, c2, c4 = self.conv[0], self.conv[2], self.conv[4]
        y = ops.subsampling8(x, c0.weight, c0.bias, c2.weight, c2.bias, c4.weight, c4.bias, self.linear.weight, self.linear.bias,
                             pe, self.pos_enc.xscale)
        return y, x_mask[:, :, :-2:2][:, :, :-2:2][:, :, :-2:2], pos


class LinearNoSubsampling(BaseSubsampling):
    """subsampling.py:23-62: Linear(idim, odim) -> LayerNorm(odim, eps 1e-12) -> positional encoding; the mask and
    the time axis pass through unchanged.  Parameter names as in the reference: out.0 (Linear), out.1 (LayerNorm)."""

    def __init__(self, idim: int, odim: int, pos_enc_class: torch.nn.Module):
        super().__init__()
        self.out = torch.nn.Sequential(torch.nn.Linear(idim, odim), torch.nn.LayerNorm(odim, eps=1e-12))
        self.pos_enc = pos_enc_class
        self.right_context = 0
        self.subsampling_rate = 1

    def forward(self, x: torch.Tensor, x_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        lin, norm = self.out[0], self.out[1]
        y = ops.linear(x, lin.weight, lin.bias)
        y = ops.layer_norm(y, norm.weight, norm.bias, norm.eps)
        pos = self.pos_enc.table(x.device, x.size(1))
        y = ops.scale_add(y, self.pos_enc.xscale, pos if self.pos_enc.kind == "abs_pos" else None)
        return y, x_mask, pos
