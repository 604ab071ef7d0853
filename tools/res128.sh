#!/bin/bash
# Register / spill figures of the 128-wide x3 kernels only (fast: the 256-wide instantiations are skipped): tools/res128.sh [-DFLAG ...]
cd "$(dirname "$0")/../tiny-nerf-pytorch_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -DTN_DIAG -DTX_SKIP256 "$@" -c mlpx3.hip -o /dev/null \
      -Rpass-analysis=kernel-resource-usage 2>&1 | grep "remark:" |
awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /SGPRs Spill/ {ss=$5} /VGPRs Spill/ {vs=$5} /LDS Size/ {printf "%-58s vgpr %-4s agpr %-4s scratch %-5s spill s%-4s v%-4s occ %-2s\n", name, v, a, s, ss, vs, o}' | grep -v "x3stats\|packx3\|x3domain\|zero_words"
