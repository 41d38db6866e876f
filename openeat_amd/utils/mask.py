"""Mask helpers with the reference's names and semantics
(/root/reference/openeat/utils/mask.py).  Index bookkeeping only - runs on
whatever device the lengths live on."""
import torch


def subsequent_mask(size: int, device: torch.device = torch.device("cpu")) -> torch.Tensor:
    """(size, size) bool, True where key index <= query index (mask.py:9-39)."""
    idx = torch.arange(size, device=device)
    return idx[None, :] <= idx[:, None]


def make_pad_mask(lengths: torch.Tensor, max_len: int = 0) -> torch.Tensor:
    """(B, max_len) bool, True on padding (mask.py:43-69).  max_len=0 -> lengths.max()."""
    n = int(max_len) if max_len > 0 else int(lengths.max().item())
    steps = torch.arange(n, dtype=torch.int64, device=lengths.device)
    return steps.unsqueeze(0) >= lengths.unsqueeze(-1)


def make_non_pad_mask(lengths: torch.Tensor) -> torch.Tensor:
    return ~make_pad_mask(lengths)


def mask_finished_scores(score: torch.Tensor, flag: torch.Tensor) -> torch.Tensor:
    """Finished beams keep exactly one live branch with score 0 (mask.py:100-128); in place."""
    beam = score.size(-1)
    first = torch.zeros(1, beam, dtype=torch.bool, device=score.device)
    first[0, 0] = True
    done = flag.view(-1, 1)
    score.masked_fill_(done & ~first, -float("inf"))
    score.masked_fill_(done & first, 0)
    return score


def mask_finished_preds(pred: torch.Tensor, flag: torch.Tensor, eos: int) -> torch.Tensor:
    """Finished beams emit <eos> on every branch (mask.py:131-146); in place."""
    return pred.masked_fill_(flag.view(-1, 1).expand_as(pred), eos)
