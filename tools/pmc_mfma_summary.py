#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from a rocprofv3 PMC pass (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over the SIMDs = 32 x N_mfma for v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE is summed over the 8 XCDs):
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra
    python3 tools/pmc_mfma_summary.py gpurun_out/pmc_mfma > profiles/r02_pmc_mfma_busy.csv"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))      # kernel -> dispatch -> counter -> value
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
dur = defaultdict(dict)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
N_SIMD = 256 * 4
print("# " + __doc__.strip().splitlines()[2].strip())
print("# mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / duration (profiled runs clock lower than plain ones)")
print("kernel,dispatches,mean_duration_us,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,mfma_busy_frac,clock_ghz,mfma_busy_time_frac_at_2.4GHz")
for k in sorted(cnt, key=lambda k: -sum(dur[k].values()) if dur[k] else 0):
    ds = [d for d in cnt[k] if d in dur[k]]
    if not ds or "SQ_VALU_MFMA_BUSY_CYCLES" not in cnt[k][ds[0]]:
        continue
    n = len(ds)
    gui = sum(cnt[k][d]["GRBM_GUI_ACTIVE"] for d in ds) / n
    busy = sum(cnt[k][d]["SQ_VALU_MFMA_BUSY_CYCLES"] for d in ds) / n
    t = sum(dur[k][d] for d in ds) / n
    if busy == 0 or t < 20000:
        continue
    print(f"\"{k.split('(')[0]}\",{n},{t / 1e3:.1f},{gui:.0f},{busy:.0f},{busy / (gui / 8 * N_SIMD):.3f},{gui / 8 / t:.3f},{busy / N_SIMD / 2.4 / t:.3f}")
