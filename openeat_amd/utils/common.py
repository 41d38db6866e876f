"""Host-side helpers with the reference's names
(/root/reference/openeat/utils/common.py).  Token bookkeeping is vectorised
(no per-utterance Python loops, no host sync unless a shape depends on it)."""
import logging
import math
from typing import List, Tuple

import torch

IGNORE_ID = -1
# When True, helpers assume targets are left-aligned and padded to the longest
# utterance of the batch (what the reference's collate function produces), so
# output shapes follow from input shapes and no device->host sync is needed.
STATIC_SHAPES = False


def init_logger(log_file=None):
    fmt = logging.Formatter("[%(asctime)s %(levelname)s] %(message)s")
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    console = logging.StreamHandler()
    console.setFormatter(fmt)
    logger.handlers = [console]
    if log_file:
        fh = logging.FileHandler(log_file)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    return logger


def map_to_device(tensor_dict, device):
    return {k: v.to(device) for k, v in tensor_dict.items()}


def pad_list(xs: List[torch.Tensor], pad_value: int):
    n = max(x.size(0) for x in xs)
    out = xs[0].new_full((len(xs), n) + tuple(xs[0].shape[1:]), pad_value)
    for i, x in enumerate(xs):
        out[i, : x.size(0)] = x
    return out


def _compact(ys_pad: torch.Tensor, ignore_id: int):
    keep = ys_pad != ignore_id
    lens = keep.sum(1)
    slot = torch.cumsum(keep, 1) - 1                      # destination column of each kept token
    return keep, lens, slot


def add_sos_eos(ys_pad: torch.Tensor, sos: int, eos: int, ignore_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """common.py:89-132: ys_in = [sos, y...] padded with eos; ys_out = [y..., eos] padded with ignore_id."""
    B, L = ys_pad.shape
    keep, lens, slot = _compact(ys_pad, ignore_id)
    width = (L if STATIC_SHAPES else (int(lens.max().item()) if B > 0 else 0)) + 1
    ys = ys_pad.to(torch.long)
    # sync-free compaction: dropped entries are scattered into a spare last column that is sliced off
    spare = width
    buf_in = ys.new_full((B, width + 1), eos)
    buf_in[:, 0] = sos
    buf_in.scatter_(1, torch.where(keep, slot + 1, torch.full_like(slot, spare)), ys)
    buf_out = ys.new_full((B, width + 1), ignore_id)
    buf_out.scatter_(1, torch.where(keep, slot, torch.full_like(slot, spare)), ys)
    buf_out.scatter_(1, lens.unsqueeze(1), torch.full((B, 1), eos, dtype=torch.long, device=ys.device))
    return buf_in[:, :width].contiguous(), buf_out[:, :width].contiguous()


def reverse_pad_list(ys_pad: torch.Tensor, ys_lens: torch.Tensor, pad_value: float = -1.0) -> torch.Tensor:
    """common.py:61-86: reverse the first ys_lens[b] entries of each row (int32, like the reference)."""
    B, L = ys_pad.shape
    lens = torch.clamp(ys_lens.to(ys_pad.device).long(), max=L)
    width = L if STATIC_SHAPES else (int(lens.max().item()) if B > 0 else 0)
    col = torch.arange(width, device=ys_pad.device).unsqueeze(0)
    src = (lens.unsqueeze(1) - 1 - col).clamp(min=0)
    out = torch.gather(ys_pad.int(), 1, src) if L > 0 else ys_pad.int()[:, :0]
    return out.masked_fill(col >= lens.unsqueeze(1), int(pad_value))


def th_accuracy(pad_outputs: torch.Tensor, pad_targets: torch.Tensor, ignore_label: int) -> torch.Tensor:
    """common.py:135-157 on materialised logits."""
    pred = pad_outputs.view(pad_targets.size(0), pad_targets.size(1), pad_outputs.size(1)).argmax(-1)
    valid = pad_targets != ignore_label
    return torch.true_divide(((pred == pad_targets) & valid).sum(), valid.sum())


def remove_duplicates_and_blank(hyp: List[int]) -> List[int]:
    """common.py:187-196."""
    out, prev = [], None
    for tok in hyp:
        if tok != prev and tok != 0:
            out.append(tok)
        prev = tok
    return out


def log_add(args: List[float]) -> float:
    """common.py:198-206."""
    if all(a == -float("inf") for a in args):
        return -float("inf")
    top = max(args)
    return top + math.log(sum(math.exp(a - top) for a in args))


def get_subsample(config):
    """common.py:176-184: the input layer's frame-rate reduction."""
    input_layer = config["encoder_conf"]["input_layer"]
    assert input_layer in ["conv2d", "conv2d6", "conv2d8"]
    return {"conv2d": 4, "conv2d6": 6, "conv2d8": 8}[input_layer]


def get_activation(act):
    """common.py:160-173 (same table; unknown names raise KeyError as there)."""
    from openeat_amd.modules.swish import Swish
    table = {"hardtanh": torch.nn.Hardtanh, "tanh": torch.nn.Tanh, "relu": torch.nn.ReLU, "selu": torch.nn.SELU, "swish": Swish,
             "gelu": torch.nn.GELU}
    return table[act]()
