#!/usr/bin/env python3
"""Timeline of the captured training step from a rocprofv3 kernel trace of `bench.py` in graph mode: for the last replays,
the step's span, the union of the intervals in which at least one kernel runs, how much of it has two or more kernels
running, and the idle gaps (count, total, largest) - i.e. whether the step is bound by kernel time or by the gaps between
dependent launches.

    python tools/graph_timeline.py <kernel_trace.csv> [steps to analyse]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name") or r.get("Name")))
rows.sort()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
for k in range(nsteps, 0, -1):
    lo, hi = adam[-k - 1] + 1, adam[-k] + 1            # kernels after the previous step's Adam up to and including this step's
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], max(e for _, e, _ in seg)
    ev = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
    depth, last, busy, multi, gaps = 0, t0, 0, 0, []
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d
        last = t
    tot = sum(e - s for s, e, _ in seg)
    big = sorted(gaps)[-5:]
    print(f"step -{k}: {len(seg)} kernels, span {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {tot / 1e6:.2f} ms, some kernel running "
          f"{busy / 1e6:.2f} ms ({multi / 1e6:.2f} ms with >= 2 at once), idle {sum(gaps) / 1e6:.2f} ms in {len(gaps)} gaps "
          f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.1f} us, largest {[round(g / 1e3, 1) for g in big]} us)")
