"""CMVN statistics files -> (mean, istd)  (/root/reference/openeat/utils/cmvn.py)."""
import json
import math

import numpy as np


def _finish(sums, sqsums, count):
    mean = np.asarray(sums, dtype=np.float64) / count
    var = np.maximum(np.asarray(sqsums, dtype=np.float64) / count - mean * mean, 1.0e-20)
    return np.array([mean, 1.0 / np.sqrt(var)])


def _load_json_cmvn(json_cmvn_file):
    """cmvn.py:21-43: {'mean_stat','var_stat','frame_num'}."""
    with open(json_cmvn_file) as f:
        st = json.load(f)
    return _finish(st["mean_stat"], st["var_stat"], st["frame_num"])


def _load_kaldi_cmvn(kaldi_cmvn_file):
    """cmvn.py:46-85: text matrix '[ sums count \\n sqsums 0 ]'."""
    with open(kaldi_cmvn_file, "r") as f:
        if f.read(2) == "\0B":
            raise ValueError("binary kaldi cmvn is not supported; recompute with --binary=false")
        f.seek(0)
        arr = f.read().split()
    assert arr[0] == "[" and arr[-2] == "0" and arr[-1] == "]"
    dim = (len(arr) - 4) // 2
    sums = [float(v) for v in arr[1: dim + 1]]
    count = float(arr[dim + 1])
    sq = [float(v) for v in arr[dim + 2: 2 * dim + 2]]
    return _finish(sums, sq, count)


def load_cmvn(cmvn_file, is_json):
    cm = _load_json_cmvn(cmvn_file) if is_json else _load_kaldi_cmvn(cmvn_file)
    return cm[0], cm[1]
