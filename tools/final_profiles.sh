#!/bin/bash
# GPU box: the measurement bundle committed under profiles/ (round tag as $1, default r01).
# Usage: gpurun -- 'bash tools/final_profiles.sh r01'
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "[1/6] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
echo "[2/6] bench (default flags)"; timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_p3.json 2> $OUT/${TAG}_bench_p3.err || { tail -5 $OUT/${TAG}_bench_p3.err; exit 1; }
echo "[3/6] kernel trace + stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --steps 5 --warmup 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python tools/summarize_kernel_trace.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 10 > $OUT/${TAG}_kernel_trace_summary.txt
echo "[4/6] pmc FETCH_SIZE"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o pmc_FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --steps 2 --warmup 1 > $OUT/pmc_f.log 2>&1 || { tail -5 $OUT/pmc_f.log; exit 1; }
echo "[5/6] pmc WRITE_SIZE"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc -o pmc_WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --steps 2 --warmup 1 > $OUT/pmc_w.log 2>&1 || { tail -5 $OUT/pmc_w.log; exit 1; }
echo "[6/6] gemm microbench"; timeout -k 10 300 python tools/gemm_bench.py 3,1,0 > $OUT/${TAG}_gemm_bench.txt 2>/dev/null
ls $OUT $OUT/trace $OUT/pmc
