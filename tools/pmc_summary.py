#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc passes of bench.py.

    python tools/pmc_summary.py --fetch DIR --write DIR [--l2 DIR] --out profiles/rN_pmc_traffic.json

Each DIR is the output directory of one pass (`rocprofv3 --pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,
`--pmc TCC_HIT_sum TCC_MISS_sum`, with --output-format csv).  Per kernel: the counter averaged over
its launches.  gfx950 correction: FETCH_SIZE tallies 128-byte requests at 64 bytes, so read bytes =
2 * FETCH_SIZE * 1024; WRITE_SIZE (KiB) is exact.  The output records the SHA-256 of the kernel
source so that bench.py only quotes the figure for the build it was measured on.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(sys.maxsize)


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name[:120]


def averages(d: str) -> dict:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": max(len(v) for v in cs.values())}
            for k, cs in acc.items()}


def source_hash() -> str:
    with open(os.path.join(ROOT, "vq_vae_amd", "csrc", "vq_search.hip"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--l2")
    ap.add_argument("--out", required=True)
    ap.add_argument("--rows", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--command", default="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline")
    ap.add_argument("--prefix", default="vq", help="keep kernels whose name starts with this")
    a = ap.parse_args()
    fe, wr = averages(a.fetch), averages(a.write)
    l2 = averages(a.l2) if a.l2 else {}
    kernels = []
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith(a.prefix):
            continue
        f = fe.get(k, {}).get("FETCH_SIZE", 0.0)
        w = wr.get(k, {}).get("WRITE_SIZE", 0.0)
        e = {"kernel": k, "launches": fe.get(k, wr.get(k))["launches"], "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
             "hbm_bytes_corrected": (2 * f + w) * 1024}
        if k in l2:
            e["TCC_HIT_sum"] = l2[k].get("TCC_HIT_sum")
            e["TCC_MISS_sum"] = l2[k].get("TCC_MISS_sum")
        kernels.append(e)
    out = {"note": "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum TCC_MISS_sum: one run each) of `"
                   + a.command + "`; per-launch averages.  gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B, "
                   "so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.",
           "source_sha256": {"vq_vae_amd/csrc/vq_search.hip": source_hash()},
           "workload": {"K": 8192, "D": 256, "rows": a.rows}, "kernels": kernels}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{a.out}: {len(kernels)} kernels")


if __name__ == "__main__":
    main()
