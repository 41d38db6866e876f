#!/bin/bash
# GPU box: same-box A/B/A/B of the step at the north-star shape (B=64 x 16 s) with and without the arena's weight planes (gemm_hyb.hip).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for r in 1 2; do for v in 1 0; do
OE_WEIGHT_PLANES=$v timeout -k 10 300 python bench.py --batch 64 --seconds 16 --target-len 48 --no-decode --no-cpu-baseline --no-other-modes --steps 8 --warmup 3 > gpurun_out/ns_$v.json 2>/dev/null || exit 1
echo "north-star weight_planes=$v: $(python -c "import json;d=json.load(open('gpurun_out/ns_$v.json'));print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'])")"
done; done
