#!/bin/bash
# GPU box: kernel traces of the step under two settings of one environment variable.  Usage: trace_ab.sh VAR valueA valueB [steps]
set -o pipefail
VAR=$1; A=$2; B=$3; STEPS=${4:-5}
OUT=gpurun_out/ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in $A $B; do
  export $VAR=$V
  rm -rf $OUT/t_$V
  timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/t_$V -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-other-modes --no-graph --single-stream --steps $STEPS --warmup 2 > $OUT/log_$V.txt 2>&1 || { tail -5 $OUT/log_$V.txt; exit 1; }
  python tools/summarize_kernel_trace.py $(ls $OUT/t_$V/*kernel_trace.csv | head -1) $STEPS > $OUT/summary_${VAR}_$V.txt
  rm -rf $OUT/t_$V
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-decode --no-other-modes --steps 20 --warmup 3 > $OUT/bench_${VAR}_$V.json 2>/dev/null || exit 1
done
grep -H ms_per_step $OUT/bench_${VAR}_*.json | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'
