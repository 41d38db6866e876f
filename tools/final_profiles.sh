#!/bin/bash
# GPU box: the measurement bundle committed under profiles/ (round tag as $1, default r01).
# Usage: gpurun -- 'bash tools/final_profiles.sh r01'
set -o pipefail
TAG=${1:-r01}
P=${2:-6}            # arithmetic mode of the headline (oe_gemm_args.precision): 6 since round 3
export OE_GEMM_PRECISION=$P
OUT=gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "[1] smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 || { tail -5 $OUT/smoke.log; exit 1; }
echo "[2] bench (default flags)"; timeout -k 10 600 python bench.py > $OUT/${TAG}_bench_p${P}.json 2> $OUT/${TAG}_bench_p${P}.err || { tail -5 $OUT/${TAG}_bench_p${P}.err; exit 1; }
echo "[3] kernel trace + stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 5 --warmup 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python tools/summarize_kernel_trace.py $(ls $OUT/trace/*kernel_trace.csv | head -1) 10 > $OUT/${TAG}_kernel_trace_summary.txt
echo "[4] pmc FETCH_SIZE"; timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc -o pmc_FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/pmc_f.log 2>&1 || { tail -5 $OUT/pmc_f.log; exit 1; }
echo "[5] pmc WRITE_SIZE"; timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc -o pmc_WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/pmc_w.log 2>&1 || { tail -5 $OUT/pmc_w.log; exit 1; }
echo "[6] gemm microbench"; timeout -k 10 300 python tools/gemm_bench.py 6,3,1,0 > $OUT/${TAG}_gemm_bench.txt 2>/dev/null
echo "[7] CTC head alone (config-2 and north-star shapes)"; timeout -k 10 200 python tools/ctc_bench.py > $OUT/${TAG}_ctc_bench.txt 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/ctc -o ctc --output-format csv -- python3 tools/ctc_bench.py > $OUT/ctc.log 2>&1 || { tail -5 $OUT/ctc.log; exit 1; }
python tools/ctc_prof_summary.py $OUT/ctc/ctc_kernel_trace.csv >> $OUT/${TAG}_ctc_bench.txt
echo "[8] MFMA busy at the north-star shape (B=64 x 16 s)"; timeout -k 10 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o mfma --output-format csv -- python3 bench.py --batch 64 --seconds 16 --target-len 48 --no-cpu-baseline --no-decode --no-graph --single-stream --steps 2 --warmup 1 > $OUT/mfma.log 2>&1 || { tail -5 $OUT/mfma.log; exit 1; }
python tools/mfma_busy_summary.py $OUT/mfma/mfma_counter_collection.csv "MFMA busy per kernel, north-star shape (B=64 x 16 s, precision $P), 4 optimizer steps" > $OUT/${TAG}_mfma_busy_northstar.md
echo "[9] attention kernels alone"; timeout -k 10 300 python tools/attn_bench.py $P > $OUT/${TAG}_attn_bench.txt 2>/dev/null
echo "[10] weight-gradient GEMMs alone"; timeout -k 10 300 python tools/tn_bench.py $P > $OUT/${TAG}_tn_bench.txt 2>/dev/null
echo "[11] north-star shape bench line"; timeout -k 10 400 python bench.py --batch 64 --seconds 16 --target-len 48 --no-decode --no-cpu-baseline --no-other-modes > $OUT/${TAG}_bench_northstar_shape.json 2> $OUT/northstar.err || { tail -5 $OUT/northstar.err; exit 1; }
echo "[12] configs[4] on one GPU: per-shape graph cache vs eager; the same ragged data through the 12L d=256 model"
timeout -k 10 500 python tools/config5_bench.py --mode cached --utts 2200 --steps 40 > $OUT/c5_cached.json 2> $OUT/c5_cached.err || { tail -5 $OUT/c5_cached.err; exit 1; }
timeout -k 10 300 python tools/config5_bench.py --mode eager --utts 2200 --steps 40 > $OUT/c5_eager.json 2> $OUT/c5_eager.err || { tail -5 $OUT/c5_eager.err; exit 1; }
timeout -k 10 300 python tools/config5_bench.py --model 12L256 --budget 32000 --mode cached --utts 1500 --steps 40 > $OUT/c5s_cached.json 2> $OUT/c5s_cached.err || { tail -5 $OUT/c5s_cached.err; exit 1; }
timeout -k 10 300 python tools/config5_bench.py --model 12L256 --budget 32000 --mode eager --utts 1500 --steps 40 > $OUT/c5s_eager.json 2> $OUT/c5s_eager.err || { tail -5 $OUT/c5s_eager.err; exit 1; }
echo "[12b] pre-split GEMM kernel alone"; timeout -k 10 300 python tools/pl_bench.py > $OUT/${TAG}_pl_bench.txt 2>/dev/null
echo "[13] decode breakdown"; timeout -k 10 300 python tools/decode_breakdown.py > $OUT/${TAG}_decode_breakdown.txt 2>/dev/null
echo "[14] un-profiled phase timeline of the captured step"; timeout -k 10 300 python tools/phase_stamps.py > $OUT/${TAG}_phase_stamps.txt 2>/dev/null
ls $OUT $OUT/trace $OUT/pmc
