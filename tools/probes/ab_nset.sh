for l in libopeneat_hip.so libopeneat_hip_fr1n4.so libopeneat_hip_fr1n8.so; do
  echo "== $l"; OE_HIP_LIB=openeat_amd/lib/$l timeout -k 10 200 python tools/probes/ring_depth.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
  OE_HIP_LIB=openeat_amd/lib/$l timeout -k 10 200 python tools/ffn6_bench.py 12000 512 2048 25472 256 1024 2>&1 | grep -E "^rows|warm" | cut -c1-260
done
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-other-modes"
run() { echo "== $1"; OE_HIP_LIB=openeat_amd/lib/$1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for i in 1 2; do run libopeneat_hip_fr1n4.so && run libopeneat_hip_fr1n8.so || exit 1; done
