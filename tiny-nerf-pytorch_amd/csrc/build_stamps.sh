#!/bin/bash
# Diagnostic build: libtnerf_hip_stamps.so = the library with s_memtime stamps in the fused forward kernel.
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p build/stamps
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DTN_STAMPS"
hipcc $FLAGS -c mlp_fwd.hip -o build/stamps/mlp_fwd.o &
hipcc $FLAGS -c wgrad.hip -o build/stamps/wgrad.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../tnerf/libtnerf_hip_stamps.so build/stage_kernels.o build/stamps/mlp_fwd.o build/mlp_bwd.o build/mlp_pair.o build/mlp16_fwd.o build/mlp16_bwd.o build/stamps/wgrad.o build/train_api.o build/host_plan.o -ldl
echo built stamps
