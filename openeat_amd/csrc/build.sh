#!/bin/bash
# Build libopeneat_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" obj
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=fast"
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ oe_common.h -nt "$o" ] || [ gemm_common.h -nt "$o" ] || [ attn_common.h -nt "$o" ] || [ ../../include/openeat_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for f in *.cpp; do
  o=obj/${f%.cpp}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ ../../include/openeat_hip.h -nt "$o" ]; then
    echo "g++ $f"
    g++ -O2 -fPIC -std=c++17 -Wall -pthread -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT/libopeneat_hip.so" obj/*.o
echo "built $OUT/libopeneat_hip.so"
# OE_DIAG=1: also build the stamped diagnostic variant (tools/gemm_stamps.py); never loaded by the product path
if [ "${OE_DIAG:-0}" = "1" ]; then
  mkdir -p obj_diag
  for f in *.hip; do $HIPCC $FLAGS -DOE_GEMM_STAMPS -c "$f" -o "obj_diag/${f%.hip}.o" & done
  wait
  $HIPCC --offload-arch=gfx950 -shared -fPIC -pthread -o "$OUT/libopeneat_hip_diag.so" obj_diag/*.o obj/beam_host.o
  echo "built $OUT/libopeneat_hip_diag.so"
fi
