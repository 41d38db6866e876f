B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-other-modes"
run() { echo "== $1"; env $1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
run "OE_X=0" && run "OE_WGRAD_GROUP_MAX_TILES=16 OE_WGRAD_GROUP_BLOCKS=1024" && run "OE_WGRAD_GROUP_MAX_TILES=16 OE_WGRAD_GROUP_BLOCKS=1536" && run "OE_WGRAD_GROUP_MAX_TILES=16 OE_WGRAD_GROUP_BLOCKS=2048" && run "OE_WGRAD_GROUP_MAX_TILES=12 OE_WGRAD_GROUP_BLOCKS=1024" && run "OE_X=0" && run "OE_WGRAD_GROUP_MAX_TILES=16 OE_WGRAD_GROUP_BLOCKS=768"
