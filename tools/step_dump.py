#!/usr/bin/env python3
"""Per-launch listing of the LAST optimizer step in a rocprofv3 kernel trace of `bench.py` (graph or eager mode):
start relative to the step's first kernel, duration, queue, how many kernels were running when it started, short name.
Used to read the concurrency of the heads phase (CTC + the two decoders) and the gaps on the main chain.

    python tools/step_dump.py <kernel_trace.csv> [step index from the end; default: the last step without a spin kernel]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name"),
                     r.get("Queue_Id", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
if k == 0:
    # default: the last step that is NOT one of bench.py's event-bracketed eager steps (those park the GPU behind a spin kernel
    # while the host enqueues) - i.e. the last replay of the captured graph in a default bench run
    k = 1
    while k + 1 < len(adam) and any("spin_kernel" in r[2] for r in rows[adam[-k - 1] + 1:adam[-k] + 1]):
        k += 1
lo, hi = adam[-k - 1] + 1, adam[-k] + 1
seg = rows[lo:hi]
t0 = seg[0][0]
queues = {}
ends = []
print(f"# {len(seg)} kernels, span {(max(r[1] for r in seg) - t0) / 1e3:.1f} us")
print("# start_us  dur_us  q  running  grid/wg  name")
for s, e, name, q, grid, wg in seg:
    qi = queues.setdefault(q, len(queues))
    ends = [x for x in ends if x > s]
    short = re.sub(r"^void ", "", name)
    short = re.sub(r"\(.*$", "", short)[:90]
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:7.1f}  {qi}  {len(ends)}  {grid}/{wg}  {short}")
    ends.append(e)
