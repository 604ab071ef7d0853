#!/bin/bash
# Round-2 profile capture, run ON the GPU box from the repo root (gpurun -- 'bash tools/capture_profiles.sh'):
#   1. PMC passes (FETCH_SIZE, WRITE_SIZE separately) + tools/traffic_capture -> gpurun_out/cap/r02_traffic*.json (and profiles/ on the box,
#      so that the bench runs below report the traffic of THESE kernel sources)
#   2. plain bench (the numbers quoted in DESIGN.md)                         -> gpurun_out/cap/bench.json
#   3. the same command under rocprofv3 --kernel-trace --stats                -> gpurun_out/cap/kernel_stats.csv, bench_profiled.json, timeline.txt
# Copy gpurun_out/cap/* into profiles/r02_* afterwards (gpurun_out/ is scratch).
set -eo pipefail
R=$PWD; O=$R/gpurun_out/cap; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$R"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra > "$O/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra > "$O/pmc_write.log" 2>&1
python3 tools/traffic_capture.py "$O/pmc_fetch" "$O/pmc_write" > "$O/traffic.log" 2>&1
cp profiles/r02_traffic.json profiles/r02_traffic_bf16.json "$O/"
timeout -k 10 500 python3 bench.py > "$O/bench.log" 2>&1
grep '^{' "$O/bench.log" | tail -1 > "$O/bench.json"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py > "$O/bench_profiled.log" 2>&1
grep '^{' "$O/bench_profiled.log" | tail -1 > "$O/bench_profiled.json"
cp "$(ls "$O"/stats/*/*kernel_stats.csv | head -1)" "$O/kernel_stats.csv"
python3 tools/timeline.py "$O/stats" > "$O/timeline.txt"
rm -rf "$O/stats" "$O/pmc_fetch" "$O/pmc_write"
echo capture done; ls -la "$O"
