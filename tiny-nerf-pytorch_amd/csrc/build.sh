#!/bin/bash
# Build libtnerf_hip.so for gfx950 in-tree (tiny-nerf-pytorch_amd/tnerf/libtnerf_hip.so).
# hipcc cross-compiles without a GPU.  -ffp-contract=off: the sample-bin arithmetic must round op by
# op like the reference's separate ATen kernels (explicit fmaf where a fused op is intended).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../tnerf/libtnerf_hip.so
OBJ=build
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result"
pids=()
for f in stage_kernels mlp_fwd mlp_bwd mlp16_fwd mlp16_bwd mlpx3 wgrad train_api step_api mlp_generic; do
  if [ ! -f "$OBJ/$f.o" ] || [ "$f.hip" -nt "$OBJ/$f.o" ] || [ -n "$(find . -maxdepth 1 \( -name '*.hpp' -o -name '*.h' \) -newer "$OBJ/$f.o" 2>/dev/null)" ] || [ ../../include/tnerf.h -nt "$OBJ/$f.o" ]; then
    hipcc $FLAGS -c "$f.hip" -o "$OBJ/$f.o" &
    pids+=($!)
  fi
done
if [ ! -f "$OBJ/host_plan.o" ] || [ host_plan.cpp -nt "$OBJ/host_plan.o" ] || [ tnerf_internal.h -nt "$OBJ/host_plan.o" ] || [ ../../include/tnerf.h -nt "$OBJ/host_plan.o" ]; then
  g++ -O2 -std=c++17 -fPIC -ffp-contract=off -c host_plan.cpp -o "$OBJ/host_plan.o" &
  pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/stage_kernels.o "$OBJ"/mlp_fwd.o "$OBJ"/mlp_bwd.o "$OBJ"/mlp16_fwd.o "$OBJ"/mlp16_bwd.o "$OBJ"/mlpx3.o "$OBJ"/wgrad.o "$OBJ"/train_api.o "$OBJ"/step_api.o "$OBJ"/mlp_generic.o "$OBJ"/host_plan.o -ldl
echo "built $OUT"
