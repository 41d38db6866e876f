#!/bin/bash
# GPU box: the measurement bundle committed under profiles/ (round tag as $1, default r01).
# Usage: gpurun -- 'bash tools/final_profiles.sh r01'
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "[1/8] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
echo "[2/8] bench (default flags)"; timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_p3.json 2> $OUT/${TAG}_bench_p3.err || { tail -5 $OUT/${TAG}_bench_p3.err; exit 1; }
echo "[3/8] kernel trace + stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 5 --warmup 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python tools/summarize_kernel_trace.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 10 > $OUT/${TAG}_kernel_trace_summary.txt
echo "[4/8] pmc FETCH_SIZE"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o pmc_FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/pmc_f.log 2>&1 || { tail -5 $OUT/pmc_f.log; exit 1; }
echo "[5/8] pmc WRITE_SIZE"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc -o pmc_WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/pmc_w.log 2>&1 || { tail -5 $OUT/pmc_w.log; exit 1; }
echo "[6/8] gemm microbench"; timeout -k 10 300 python tools/gemm_bench.py 3,1,0 > $OUT/${TAG}_gemm_bench.txt 2>/dev/null
echo "[7/8] CTC head alone (config-2 and north-star shapes)"; timeout -k 10 200 python tools/ctc_bench.py > $OUT/${TAG}_ctc_bench.txt 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/ctc -o ctc --output-format csv -- python3 tools/ctc_bench.py > $OUT/ctc.log 2>&1 || { tail -5 $OUT/ctc.log; exit 1; }
python tools/ctc_prof_summary.py $OUT/ctc/ctc_kernel_trace.csv >> $OUT/${TAG}_ctc_bench.txt
echo "[8/8] MFMA busy at the north-star shape (B=64 x 16 s)"; timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o mfma --output-format csv -- python3 bench.py --batch 64 --seconds 16 --target-len 48 --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/mfma.log 2>&1 || { tail -5 $OUT/mfma.log; exit 1; }
python tools/mfma_busy_summary.py $OUT/mfma/mfma_counter_collection.csv "MFMA busy per kernel, north-star shape (B=64 x 16 s, precision 3), 4 optimizer steps" > $OUT/${TAG}_mfma_busy_northstar.md
ls $OUT $OUT/trace $OUT/pmc
