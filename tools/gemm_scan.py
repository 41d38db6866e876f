#!/usr/bin/env python3
"""K-scan of one GEMM shape: intercept = prologue+epilogue cost, slope = cost per K-tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import run  # noqa: E402

if __name__ == "__main__":
    for kind, m, n in (("nt", 7936, 1024), ("nt", 7936, 256), ("nn", 7936, 1024), ("tn", 1024, 256)):
        for prec in (3, 1):
            row = []
            ks = (32, 64, 128, 256, 512, 1024, 2048) if kind != "tn" else (512, 1024, 2048, 4096, 7936, 15872)
            for k in ks:
                us, tf = run(kind, m, n, k, prec, reps=10)
                row.append(f"k={k}:{us:7.1f}us")
            print(f"{kind} m={m} n={n} p{prec}  " + "  ".join(row), flush=True)
