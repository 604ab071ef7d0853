#!/bin/bash
# Register / scratch / occupancy figures of the main kernels: tools/kernel_resources.sh [csrc dir]
# (hipcc -Rpass-analysis=kernel-resource-usage; scratch > 0 or spills in a chain kernel mean the register-resident design broke.)
DIR=${1:-$(dirname "$0")/../tiny-nerf-pytorch_amd/csrc}
cd "$DIR" || exit 1
for f in mlp_fwd mlp_bwd wgrad mlp16_fwd mlp16_bwd mlpx3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -c $f.hip -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /SGPRs Spill/ {ss=$5} /VGPRs Spill/ {vs=$5} /LDS Size/ {printf "%-58s vgpr %-4s agpr %-4s scratch %-5s spill s%-4s v%-4s occ %-2s lds %s\n", name, v, a, s, ss, vs, o, $5}'
done
