"""CTC head (/root/reference/openeat/modules/ctc.py)."""
import torch

from openeat_amd import ops


class CTC(torch.nn.Module):
    def __init__(self, odim: int, encoder_output_size: int, length_normalized_loss: bool = False):
        super().__init__()
        self.length_normalized_loss = bool(length_normalized_loss)      # ctc.py:24: CTCLoss(reduction='mean')
        self.ctc_lo = torch.nn.Linear(encoder_output_size, odim)
        self.odim = odim

    def forward(self, hs_pad: torch.Tensor, hlens: torch.Tensor, ys_pad: torch.Tensor, ys_lens: torch.Tensor):
        """ctc.py:27-45: projection, log-softmax, CTC loss (sum or length-normalised mean, zero_infinity) / batch -
        one fused op whose backward is already computed when forward returns."""
        return ops.ctc_head(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias, hlens, ys_pad, ys_lens, self.length_normalized_loss)

    def logits(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return ops.linear(hs_pad, self.ctc_lo.weight, self.ctc_lo.bias)

    def log_softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        """ctc.py:56-64."""
        return ops.log_softmax_rows(self.logits(hs_pad))

    def softmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        return self.log_softmax(hs_pad).exp()

    def argmax(self, hs_pad: torch.Tensor) -> torch.Tensor:
        """ctc.py:66-74 (lowest index on ties)."""
        lg = self.logits(hs_pad)
        B, T, V = lg.shape
        full = torch.full((B,), T, dtype=torch.int32, device=lg.device)
        fb = torch.empty(B, T, dtype=torch.int32, device=lg.device)
        ot, ol = torch.empty_like(fb), torch.empty(B, dtype=torch.int32, device=lg.device)
        from openeat_amd import hip
        hip.call("oe_ctc_greedy", lg, V, B, T, V, full, V - 1, fb, ot, ol)
        return fb.long()
