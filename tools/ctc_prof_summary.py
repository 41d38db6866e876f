#!/usr/bin/env python3
"""Median duration of the CTC kernels per launch shape from a rocprofv3 kernel trace of tools/ctc_bench.py."""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "ctc" in n:
        d[(n.split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, blocks), v in sorted(d.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    v = sorted(v)
    print(f"{n:48s} {blocks:7d} blocks  {len(v):3d} launches  median {v[len(v) // 2] / 1e3:8.1f} us")
