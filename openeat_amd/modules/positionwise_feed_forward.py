"""Position-wise feed forward (/root/reference/openeat/modules/positionwise_feed_forward.py)."""
import torch

from openeat_amd import ops


def act_id_of(activation: torch.nn.Module) -> int:
    """The kernel id of one of the reference's activation modules (utils/common.py:160-173)."""
    if getattr(activation, "act_id", None) is not None:          # Swish
        return activation.act_id
    table = ((torch.nn.ReLU, ops.ACT_RELU), (torch.nn.Tanh, ops.ACT_TANH), (torch.nn.SELU, ops.ACT_SELU))
    for cls, aid in table:
        if isinstance(activation, cls):
            return aid
    if isinstance(activation, torch.nn.Hardtanh) and activation.min_val == -1.0 and activation.max_val == 1.0:
        return ops.ACT_HARDTANH
    if isinstance(activation, torch.nn.GELU) and getattr(activation, "approximate", "none") == "none":
        return ops.ACT_GELU
    raise NotImplementedError(f"activation {activation!r} has no gfx950 kernel")


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
