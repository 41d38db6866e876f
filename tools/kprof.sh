#!/bin/bash
# GPU box: per-kernel durations of one command (rocprofv3 --kernel-trace --stats, csv) summarised per kernel.
# Usage: bash tools/kprof.sh TAG [steps-divisor] -- python3 script.py args...     (the program itself after --, never a wrapper)
TAG=$1; shift
DIV=1
if [ "$1" != "--" ]; then DIV=$1; shift; fi
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$TAG
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o $TAG --output-format csv -- "$@" > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
python tools/summarize_kernel_trace.py $(ls gpurun_out/prof_$TAG/*kernel_trace.csv | head -1) $DIV | tee gpurun_out/prof_${TAG}_summary.txt
