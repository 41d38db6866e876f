#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (with --kernel-trace only) of a command, per-kernel averages of every counter.
# Usage: bash tools/pmc.sh TAG "COUNTER1 COUNTER2 ..." FILTER -- python3 script.py args...
TAG=$1; CTRS=$2; FILT=$3; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_$TAG
timeout -k 10 600 rocprofv3 --pmc $CTRS --kernel-trace -d gpurun_out/pmc_$TAG -o $TAG --output-format csv -- "$@" > gpurun_out/pmc_$TAG.log 2>&1 || { tail -5 gpurun_out/pmc_$TAG.log; exit 1; }
python3 - "$(ls gpurun_out/pmc_$TAG/*counter_collection.csv | head -1)" "$FILT" <<'PY' | tee gpurun_out/pmc_${TAG}_summary.txt
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    if sys.argv[2] not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    d = max(len(n[k]), 1)
    print(k, f"({d} dispatches)")
    for c, x in sorted(v.items()): print(f"    {c:32s} {x / d:16.0f}")
PY
