#!/bin/bash
# GPU box: step time at the north-star shape (B=64 x 16 s) under the pre-split-operand policies and 256 x 256 tile thresholds.
set -o pipefail
OUT=gpurun_out/ns; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for P in conv ln all; do
  for T in 1024 300; do
    OE_PLANES=$P OE_PL_T44_MIN=$T timeout -k 10 300 python bench.py --batch 64 --seconds 16 --target-len 48 --no-decode --no-cpu-baseline --no-other-modes --steps 8 --warmup 3 > $OUT/ns_${P}_$T.json 2> $OUT/ns_${P}_$T.err || { tail -5 $OUT/ns_${P}_$T.err; exit 1; }
    echo "north-star planes=$P t44min=$T $(python -c "import json;d=json.load(open('$OUT/ns_${P}_$T.json'));print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'])")"
  done
done
