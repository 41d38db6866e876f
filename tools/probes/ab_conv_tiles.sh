B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-other-modes --gemm-table"
for e in "OE_PL_HYBRID=0" "OE_PL_HYBRID=1"; do
  echo "== $e"; env $e $B 2>&1 >/dev/null | grep -E "\((15[0-9]{4}|256, 2304), " | cut -c1-120
done
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-other-modes"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "OE_PL_HYBRID=0" && run "OE_PL_HYBRID=1" && run "OE_PL_HYBRID=0" && run "OE_PL_HYBRID=1"
