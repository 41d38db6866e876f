B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-other-modes --gemm-table"
for e in "OE_X=0" "OE_PL_TILE=22" "OE_PL_TILE=44"; do
  echo "== $e"; env $e $B 2>&1 >/dev/null | grep -E "\((15[0-9]{4}|256, 2304), " | cut -c1-120
done
