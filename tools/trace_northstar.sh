#!/bin/bash
# GPU box: kernel trace of the north-star shape (B=64 x 16 s), per-kernel table.  Usage: trace_northstar.sh [tag]
set -o pipefail
TAG=${1:-ns}
OUT=gpurun_out/ns; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $OUT/t_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace -d $OUT/t_$TAG -o t --output-format csv -- python3 bench.py --batch 64 --seconds 16 --target-len 48 --no-cpu-baseline --no-decode --no-other-modes --no-graph --single-stream --steps 3 --warmup 1 > $OUT/log_$TAG.txt 2>&1 || { tail -5 $OUT/log_$TAG.txt; exit 1; }
python tools/summarize_kernel_trace.py $(ls $OUT/t_$TAG/*kernel_trace.csv | head -1) 3 > $OUT/summary_$TAG.txt
rm -rf $OUT/t_$TAG
timeout -k 10 300 python bench.py --batch 64 --seconds 16 --target-len 48 --no-cpu-baseline --no-decode --no-other-modes --steps 10 --warmup 2 > $OUT/bench_$TAG.json 2>/dev/null || exit 1
