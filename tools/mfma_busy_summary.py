#!/usr/bin/env python3
"""MFMA-busy share per kernel from one rocprofv3 PMC pass (`--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace`).

SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs (32 per
v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is the sum of the 8 XCDs' active cycles of the dispatch.
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)

usage: mfma_busy_summary.py <counter_collection.csv> [title]
"""
import collections
import csv
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    key = (re.sub(r"\(.*", "", r["Kernel_Name"])[:80], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[key] += 1
        dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
title = sys.argv[2] if len(sys.argv) > 2 else "MFMA busy per kernel"
print(f"# {title}\n")
print("`MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)`; one PMC pass with `--kernel-trace` only; "
      "durations are those of the profiled pass.\n")
print("| kernel | blocks | launches | avg us | MFMA busy % | share of all MFMA-busy cycles % |\n|---|---|---|---|---|---|")
tot = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in agg.values()) or 1.0
cls = collections.defaultdict(lambda: [0.0, 0.0])
for key, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
    busy, act = v["SQ_VALU_MFMA_BUSY_CYCLES"], v["GRBM_GUI_ACTIVE"]
    c = "attention" if "attn_" in key[0] else "gemm" if ("gemm_" in key[0] or "ffn6_kernel" in key[0] or ("rowgemm6" in key[0] or "rowtile6" in key[0]) or "ffn_fwd_kernel" in key[0]) else "other"
    cls[c][0] += busy
    cls[c][1] += act
    if busy <= 0:
        continue
    print(f"| `{key[0]}` | {key[1]} | {cnt[key]} | {dur[key] / max(cnt[key], 1) / 1e3:.1f} | {100 * busy / (act / 8 * 1024):.1f} | {100 * busy / tot:.1f} |")
print()
for c, (busy, act) in cls.items():
    if act > 0:
        print(f"* class `{c}`: MFMA busy {100 * busy / (act / 8 * 1024):.1f} % of its kernels' active cycles")
