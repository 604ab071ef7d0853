#!/usr/bin/env python3
"""HBM bytes per launch of the step's kernels from rocprofv3 PMC passes (MI355X_MICROARCH.md, "HBM": FETCH_SIZE and
WRITE_SIZE in SEPARATE passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports half the bytes of a
16-B-per-lane coalesced read stream: doubled here for the kernels whose loads are such streams).

On the GPU box (the profiler's program goes straight after `--`):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 5 --warmup 3 --psnr-steps 0 --no-cpu-baseline --no-extra
    python3 tools/traffic_capture.py gpurun_out/pmc_fetch gpurun_out/pmc_write
writes profiles/r04_traffic.json and profiles/r04_traffic_bf16.json tagged with the sha of csrc/ (bench.py reports them only
while the kernel sources are the ones that were profiled)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha, algorithmic sizes)

# kernel-name prefix -> (json file, key, FETCH_SIZE correction factor: 2 = the loads are 16-B-per-lane coalesced streams)
KERNELS = [
    ("void k_renderx3<256, true>", "fp32", "train_fwd", 1),
    ("void k_dgradx3<256>", "fp32", "dgrad", 1),
    ("void k_renderx3<256, false>", "fp32", "render_fwd", 1),
    ("void k_render_fused<256, 20, true>", "fp32", "train_fwd_fp32_mfma", 1),
    ("void k_train_bwd<256>", "fp32", "dgrad_fp32_mfma", 1),
    ("void k_wgrad<1>", "fp32", "wgrad", 2),
    ("void k_wgrad<0>", "fp32", "wgrad_fp32_mfma", 2),
    ("void k_render_fused<256, 20, false>", "fp32", "render_fwd_fp32_mfma", 1),
    ("void k_finish<true, true>", "fp32", "finish", 1),
    ("void k_render16<256, true>", "bf16", "train_fwd", 1),
    ("void k_dgrad16<256>", "bf16", "dgrad", 1),
    ("k_wgrad16", "bf16", "wgrad", 2),
    ("void k_render16<256, false>", "bf16", "render_fwd", 1),
]


def read_counter(root, counter):
    """mean counter value per dispatch by kernel name"""
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {root}")
    per = defaultdict(lambda: defaultdict(float))          # kernel -> dispatch -> value (summed over XCC instances)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def main():
    fetch = read_counter(sys.argv[1], "FETCH_SIZE")        # KB
    write = read_counter(sys.argv[2], "WRITE_SIZE")
    sha = bench.kernel_source_sha()
    M = bench.RAYS * bench.SAMPLES
    NT = bench.HIDDEN // 32
    rows32 = 2 * 20 + 2 * (bench.DEPTH * bench.HIDDEN + 4)          # stash rows of the fp32 path (NE = 20 input steps)
    algo = {"fp32": {"train_fwd": M * (2 * 20 + bench.DEPTH * bench.HIDDEN + 4) * 4 + bench.DEPTH * M * bench.HIDDEN // 8,
                     "dgrad": M * (bench.DEPTH * bench.HIDDEN + 4) * 4 + bench.DEPTH * M * bench.HIDDEN // 8,
                     "wgrad": M * 4 * ((bench.HIDDEN + 64) * 2 + (bench.DEPTH - 1) * 2 * bench.HIDDEN + 32 + bench.HIDDEN),
                     "finish": 64 * 2 ** 20 + 8 * 4 * 481796, "render_fwd": bench.RAYS * 36},
            "bf16": {}}
    for k_ in ("wgrad", "train_fwd", "dgrad", "render_fwd"):
        algo["fp32"][k_ + "_fp32_mfma"] = algo["fp32"][k_]
    out = {"fp32": {}, "bf16": {}}
    for prefix, fam, key, corr in KERNELS:
        f = next((v for k, v in fetch.items() if k.startswith(prefix)), None)
        w = next((v for k, v in write.items() if k.startswith(prefix)), None)
        if f is None and w is None:
            continue
        fkb, wkb = (f[0] if f else 0.0), (w[0] if w else 0.0)
        out[fam][key] = {"kernel": prefix, "fetch_kb_raw": round(fkb, 1), "write_kb_raw": round(wkb, 1), "fetch_correction": corr,
                         "hbm_bytes": int((fkb * corr + wkb) * 1024), "dispatches": (f or w)[1],
                         "algorithmic_bytes": algo[fam].get(key)}
    note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 5 --warmup 3 "
            "--psnr-steps 0 --no-cpu-baseline --no-extra; mean per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE "
            "counts half the bytes of a 16-B-per-lane coalesced read stream -> x2 where fetch_correction = 2 (the weight-gradient kernels' "
            "operand streams); the 4-B-per-lane stores/loads of the chain kernels are uncalibrated (raw values kept).")
    for fam, name in (("fp32", "r04_traffic.json"), ("bf16", "r04_traffic_bf16.json")):
        d = {"_note": note, "kernel_source_sha": sha}
        d.update(out[fam])
        with open(os.path.join(ROOT, "profiles", name), "w") as fh:
            json.dump(d, fh, indent=1)
        print(name, json.dumps(out[fam])[:1500])


if __name__ == "__main__":
    main()
