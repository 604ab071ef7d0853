#!/bin/bash
# Round-4 profile capture, run ON the GPU box from the repo root (gpurun -- 'bash tools/capture_profiles.sh'); build the stamps
# variant first in the build container (tools/build_variant.sh x3stamps mlpx3 -DTN_STAMPS):
#   1. PMC passes (FETCH_SIZE, WRITE_SIZE separately) + tools/traffic_capture -> r04_traffic*.json (also into profiles/ on the box, so
#      that the bench runs below report the traffic of THESE kernel sources)
#   2. PMC: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE over a short bench       -> r04_pmc_mfma_busy.csv
#   3. PMC: instruction mix / LDS waits of the x3 chain kernels alone            -> r04_pmc_x3_insts.txt
#   4. plain bench (the numbers quoted in DESIGN.md)                             -> r04_bench.json
#   5. the same command under rocprofv3 --kernel-trace --stats                    -> r04_bench_kernel_stats.csv, r04_bench_profiled.json, r04_timeline.txt
#   6. per-pass cycle stamps of the forward kernel (8x256 and the reference default 4x128) -> r04_x3_stamps.txt
# Copy gpurun_out/cap/r04_* into profiles/ afterwards (gpurun_out/ is scratch).
set -eo pipefail
R=$PWD; O=$R/gpurun_out/cap; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$R"
SHORT="bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 $SHORT > "$O/pmc_fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 $SHORT > "$O/pmc_write.log" 2>&1
python3 tools/traffic_capture.py "$O/pmc_fetch" "$O/pmc_write" > "$O/traffic.log" 2>&1
cp profiles/r04_traffic.json profiles/r04_traffic_bf16.json "$O/"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_mfma" -- python3 $SHORT > "$O/pmc_mfma.log" 2>&1
python3 tools/pmc_mfma_summary.py "$O/pmc_mfma" > "$O/r04_pmc_mfma_busy.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_i1" -- python3 tools/x3_pmc_run.py > "$O/pmc_i1.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$O/pmc_i2" -- python3 tools/x3_pmc_run.py > "$O/pmc_i2.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d "$O/pmc_i3" -- python3 tools/x3_pmc_run.py > "$O/pmc_i3.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/x3_pmc_run.py  (three passes; SQ_* in units of 4 cycles except *_BUSY_CYCLES; means per dispatch, summed over XCDs)"; python3 tools/pmc_table.py "$O/pmc_i1" "$O/pmc_i2" "$O/pmc_i3"; } > "$O/r04_pmc_x3_insts.txt"
timeout -k 10 600 python3 bench.py > "$O/bench.log" 2>&1
grep '^{' "$O/bench.log" | tail -1 > "$O/r04_bench.json"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 bench.py > "$O/bench_profiled.log" 2>&1
grep '^{' "$O/bench_profiled.log" | tail -1 > "$O/r04_bench_profiled.json"
cp "$(ls "$O"/stats/*/*kernel_stats.csv | head -1)" "$O/r04_bench_kernel_stats.csv"
python3 tools/timeline.py "$O/stats" > "$O/r04_timeline.txt"
if [ -f tiny-nerf-pytorch_amd/tnerf/libtnerf_variant_x3stamps.so ]; then
  timeout -k 10 200 python3 tools/x3_stamp_probe.py x3stamps > "$O/r04_x3_stamps.txt" 2>&1 || true
fi
rm -rf "$O/stats" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_mfma" "$O/pmc_i1" "$O/pmc_i2" "$O/pmc_i3"
echo capture done; ls -la "$O"
