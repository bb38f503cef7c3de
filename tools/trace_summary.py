#!/usr/bin/env python3
"""Per (kernel, grid size) launch statistics from a rocprofv3 --kernel-trace CSV.

    python tools/trace_summary.py DIR_OR_CSV OUT.csv [--prefix vq]

rocprofv3's own --stats table averages all launches of a kernel; bench.py launches the search kernel on
several batch sizes (the timed 4 Mi-row steps, the 16 Ki-row parity sample, the 64 Ki-row side timing, the
empty launches in the filtered search's fall-back role), so the table here keeps them apart.
"""
import collections
import csv
import glob
import os
import re
import statistics
import sys

csv.field_size_limit(sys.maxsize)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:100]


def main():
    src, out = sys.argv[1], sys.argv[2]
    prefix = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--prefix" else "vq"
    paths = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for p in paths:
        with open(p, newline="") as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if not k.startswith(prefix):
                    continue
                grid = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y", "1"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
                acc[(k, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_x_threads", "grid_y", "workgroup", "launches", "avg_us", "median_us", "min_us", "max_us", "total_ms"])
        for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, g[0], g[1], g[2], len(v), f"{sum(v) / len(v) / 1e3:.1f}", f"{statistics.median(v) / 1e3:.1f}",
                        f"{min(v) / 1e3:.1f}", f"{max(v) / 1e3:.1f}", f"{sum(v) / 1e6:.2f}"])
    print(out, len(acc), "rows")


if __name__ == "__main__":
    main()
