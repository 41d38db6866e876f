#!/bin/bash
# GPU box: configs[4]'s model (24L d=512) on one GPU, eager steps, under the pre-split-operand policies (planes.POLICY).
set -o pipefail
OUT=gpurun_out/c5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  IFS='|' read -r label envs <<< "$spec"
  env $envs timeout -k 10 400 python tools/config5_bench.py --mode eager --utts 900 --steps 16 --warmup 3 > $OUT/$label.json 2> $OUT/$label.err || { tail -5 $OUT/$label.err; exit 1; }
  echo "$label: $(python -c "import json;d=json.load(open('$OUT/$label.json'));print('%.1f ms/step, gemm %.1f ms at %.1f TFLOP/s, loss %.3f' % (d['ms_per_step'], d['gemm_class_roofline']['gemm_ms'], d['gemm_class_roofline']['achieved_tflops'], d['loss']))")"
done
