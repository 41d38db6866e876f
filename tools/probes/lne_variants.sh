# The packed-fp32 hazard in the LayerNorm-backward epilogue of oe_rowgemm6 (ffn6.hip): production library against a build with
# -DOE_LNE_REPRO (hipcc's v_pk_add_f32 / v_pk_mul_f32 form of (x - mean) rstd, no barrier in front of the epilogue).
#   cd openeat_amd/csrc && hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=fast -DOE_LNE_REPRO -c ffn6.hip -o /tmp/ffn6_repro.o \
#     && hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libopeneat_hip_lne_repro.so $(ls obj/*.o | grep -v obj/ffn6.o) /tmp/ffn6_repro.o
for lib in libopeneat_hip.so libopeneat_hip_lne_repro.so; do
  for a in 0 2; do
    echo "== $lib, activation $a"
    LNE_ACT=$a OE_HIP_LIB=openeat_amd/lib/$lib timeout -k 10 200 python tools/probes/dbg_lne.py 40 2>&1 | grep -v amdgpu.ids | awk '/^trial/{n++} /worst/{print} END{print n+0, "of 40 launches wrong"}'
  done
done
