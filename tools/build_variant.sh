#!/bin/bash
# Diagnostic builds: tools/build_variant.sh <name> <file> [-DFLAG ...]  ->  tiny-nerf-pytorch_amd/tnerf/libtnerf_variant_<name>.so
# (the product library with ONE object recompiled with -DTN_DIAG and the extra defines; loaded with TNERF_LIB=...; written next to the
#  product library so that it travels with gpurun; *.so is git-ignored)
set -euo pipefail
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../tiny-nerf-pytorch_amd/csrc"
bash build.sh >/dev/null
OBJ=build; V=build/variant_$NAME; mkdir -p "$V"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -DTN_DIAG "$@" -c "$SRC.hip" -o "$V/$SRC.o"
objs=""
for f in stage_kernels mlp_fwd mlp_bwd mlp16_fwd mlp16_bwd mlpx3 wgrad train_api step_api mlp_generic host_plan; do
  if [ "$f" = "$SRC" ]; then objs="$objs $V/$f.o"; else objs="$objs $OBJ/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tnerf/libtnerf_variant_$NAME.so $objs -ldl
echo "built tnerf/libtnerf_variant_$NAME.so"
