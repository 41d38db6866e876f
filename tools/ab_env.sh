#!/bin/bash
# GPU box: same-box A/B/A/B of the captured step under environment settings / bench flags.
# Usage: ab_env.sh "<label>|<env assignments>|<extra bench flags>" ...   (each variant runs twice, interleaved)
set -o pipefail
OUT=gpurun_out/ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for spec in "$@"; do
    IFS='|' read -r label envs flags <<< "$spec"
    env $envs timeout -k 10 300 python bench.py --no-cpu-baseline --no-decode --no-other-modes --steps 30 --warmup 3 $flags > $OUT/${label}_$round.json 2> $OUT/${label}_$round.err || { tail -5 $OUT/${label}_$round.err; exit 1; }
    echo "$label round $round: $(python -c "import json;d=json.load(open('$OUT/${label}_$round.json'));print('%.3f ms/step, gemm %.2f, loss %.4f' % (d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['loss']))")"
  done
done
