#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) of
`bench.py --no-graph` into profiles/<tag>_pmc_hbm_traffic.json + a per-kernel markdown table.

Corrections as MI355X_MICROARCH.md prescribes: counter values are KB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads, so it is doubled.

usage: pmc_summary.py <dir with pmc_FETCH_SIZE_counter_collection.csv / pmc_WRITE_SIZE_...> <steps in trace> <precision> <tag>
"""
import collections
import csv
import json
import os
import re
import sys

d, steps, prec, tag = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(os.path.join(d, f"pmc_{counter}_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    return agg


fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
# optimizer steps in the trace: the Adam kernel runs once per step (the command-line figure is only a fallback)
adam = [v[0] for k, v in fetch.items() if "adam_kernel" in k]
if adam:
    steps = float(adam[0])
is_gemm = lambda n: ("gemm_" in n and "kernel" in n) or "ffn_fwd_kernel" in n or "ffn6_kernel" in n or ("rowgemm6" in n or "rowtile6" in n)
rows = []
for name in sorted(set(fetch) | set(write)):
    n = max(fetch[name][0], write[name][0]) / steps
    rows.append((name, n, 2 * fetch[name][1] * 1024 / steps, write[name][1] * 1024 / steps))
g = [r for r in rows if is_gemm(r[0])]
gl = sum(r[1] for r in g)
gb = sum(r[2] + r[3] for r in g)
out = {
    "round": int(tag[1:]), "precision": prec, "steps_in_trace": steps,
    "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --no-graph "
               "--no-cpu-baseline --no-decode (two separate passes)",
    "correction": "KB -> bytes x1024; FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads, MI355X_MICROARCH.md HBM section)",
    "gemm": {"kernels": sorted({r[0] for r in g}), "launches_per_step": gl, "read_bytes_per_step": sum(r[2] for r in g),
             "write_bytes_per_step": sum(r[3] for r in g), "hbm_bytes_per_step": gb, "hbm_bytes_per_launch": gb / max(gl, 1)},
    "all_kernels": {"read_bytes_per_step": sum(r[2] for r in rows), "write_bytes_per_step": sum(r[3] for r in rows),
                    "hbm_bytes_per_step": sum(r[2] + r[3] for r in rows)},
}
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_per_kernel.md"), "w") as f:
    f.write(f"# HBM traffic per kernel from PMC counters (precision {prec}, {steps:g} steps in the trace)\n\n")
    f.write("Two separate rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--kernel-trace` only). FETCH_SIZE doubled "
            "(gfx950 half-count of wide coalesced reads); MB per optimizer step.\n\n| kernel | launches/step | read MB/step | write MB/step |\n|---|---|---|---|\n")
    for name, n, rd, wr in sorted(rows, key=lambda r: -(r[2] + r[3]))[:40]:
        f.write(f"| `{name[:100]}` | {n:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} |\n")
    f.write(f"\nGEMM kernels: {gl:.0f} launches/step, {gb / 1e9:.2f} GB/step = {gb / max(gl, 1) / 1e6:.1f} MB/launch; "
            f"all kernels {out['all_kernels']['hbm_bytes_per_step'] / 1e9:.2f} GB/step.\n")
print(json.dumps(out["gemm"])[:300])
