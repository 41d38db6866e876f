"""Position-wise feed forward (/root/reference/openeat/modules/positionwise_feed_forward.py)."""
import torch

from openeat_amd import ops


def act_id_of(activation: torch.nn.Module) -> int:
    if isinstance(activation, torch.nn.ReLU):
        return ops.ACT_RELU
    if getattr(activation, "act_id", None) is not None:
        return activation.act_id
    raise NotImplementedError(f"activation {type(activation).__name__} has no gfx950 kernel yet")


class PositionwiseFeedForward(torch.nn.Module):
    """w_2(dropout(activation(w_1(x)))); both GEMMs, the activation and the dropout are one fused op."""

    def __init__(self, idim: int, hidden_units: int, dropout_rate: float, activation: torch.nn.Module = torch.nn.ReLU()):
        super().__init__()
        self.w_1 = torch.nn.Linear(idim, hidden_units)
        self.activation = activation
        self.dropout = torch.nn.Dropout(dropout_rate)
        self.w_2 = torch.nn.Linear(hidden_units, idim)
        self._act = act_id_of(activation)

    def forward(self, xs: torch.Tensor, residual: torch.Tensor = None, out_scale: float = 1.0,
                out_dropout: float = 0.0) -> torch.Tensor:
        """``residual``/``out_scale``/``out_dropout`` let the caller fuse
        ``residual + out_scale * dropout(ff(xs))`` into the second GEMM's epilogue."""
        p_in = self.dropout.p if self.training else 0.0
        return ops.feed_forward(xs, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias, self._act, p_in,
                                residual, out_scale, out_dropout if self.training else 0.0)
