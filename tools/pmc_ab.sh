#!/bin/bash
# GPU box: FETCH_SIZE per kernel under two settings of one environment variable.  Usage: pmc_ab.sh VAR a b kernel-substring
set -o pipefail
VAR=$1; A=$2; B=$3; PAT=$4
OUT=gpurun_out/pmcab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in $A $B; do
  export $VAR=$V
  rm -rf $OUT/p_$V
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/p_$V -o p --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-other-modes --no-graph --single-stream --steps 2 --warmup 1 > $OUT/log_$V.txt 2>&1 || { tail -5 $OUT/log_$V.txt; exit 1; }
  python - "$OUT/p_$V/p_counter_collection.csv" "$PAT" "$VAR=$V" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"][:90]
        tot[k][0] += 1; tot[k][1] += float(r["Counter_Value"])
for k, (n, v) in tot.items():
    print(sys.argv[3], k, "launches", n, "FETCH_SIZE x2 (MB per launch):", round(2 * v * 1024 / n / 1e6, 1))   # counter in KiB, gfx950 half-count
PY
  rm -rf $OUT/p_$V
done
