#!/usr/bin/env python3
"""Per-kernel durations and the idle gap in front of each kernel, from a rocprofv3 --kernel-trace CSV:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py ...
    python tools/timeline.py gpurun_out/tl [--last 400]
Answers "where does step - sum(big kernels) go": launch gaps vs small kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if last:
        rows = rows[-last:]
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    prev_end = None
    for s, e, n in rows:
        n = n.split("(")[0][:70]
        dur[n] += e - s; cnt[n] += 1
        if prev_end is not None:
            gap[n] += max(0, s - prev_end)
        prev_end = max(prev_end or 0, e)
    span = rows[-1][1] - rows[0][0]
    busy = sum(dur.values())
    print(f"{len(rows)} kernels over {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({100.0 * busy / span:.1f} %)")
    print(f"{'kernel':72s} {'n':>6s} {'avg us':>10s} {'gap before us':>14s}")
    for n in sorted(dur, key=lambda k: -dur[k]):
        print(f"{n:72s} {cnt[n]:6d} {dur[n] / cnt[n] / 1e3:10.2f} {gap[n] / cnt[n] / 1e3:14.2f}")


if __name__ == "__main__":
    main()
