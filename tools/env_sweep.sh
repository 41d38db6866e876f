#!/bin/bash
# GPU box: bench.py's step time under values of one environment variable, interleaved twice.  Usage: env_sweep.sh VAR v1 v2 ...
VAR=$1; shift
for rep in 1 2; do for V in "$@"; do
  export $VAR=$V
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-decode --no-other-modes --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$V', round(d['ms_per_step'],3))" || exit 1
done; done
