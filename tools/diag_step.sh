#!/bin/bash
# GPU box: quick bench line and a per-launch dump of one replay of the captured step (tools/step_dump.py, tools/graph_timeline.py).
set -o pipefail
OUT=gpurun_out/diag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-decode --no-other-modes --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python -c "import json;d=json.load(open('$OUT/bench.json'));print('ms_per_step', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], 'loss', d['loss'])"
if [ "${1:-}" = "trace" ]; then
rm -rf $OUT/tg
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/tg -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-decode --no-other-modes --force-graph --steps 4 --warmup 2 > $OUT/tg.log 2>&1 || { tail -5 $OUT/tg.log; exit 1; }
python tools/step_dump.py $(ls $OUT/tg/*kernel_trace.csv | head -1) > $OUT/step_dump_graph.txt
rm -rf $OUT/tg
head -1 $OUT/step_dump_graph.txt
fi
