#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of scripts/collect_profiles.sh (FETCH_SIZE, WRITE_SIZE; one TCC pass
each, as MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE is in KiB and, on gfx950, reports exactly half of the bytes of wide coalesced reads (checked here:
the LayerNorm kernel reads what it writes and its FETCH_SIZE is half its WRITE_SIZE), hence the factor 2.

    python scripts/summarize_traffic.py gpurun_out/profiles_r01 profiles/r01_hbm_traffic.json
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            agg[m.group(1)][0] += 1
            agg[m.group(1)][1] += float(r["Counter_Value"])
    return agg


def main(src, dst):
    fetch, write = per_kernel(f"{src}/pmc_FETCH_SIZE.csv"), per_kernel(f"{src}/pmc_WRITE_SIZE.csv")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "bench.py --steps 1 --warmup 1; FETCH_SIZE doubled (gfx950), KiB -> bytes",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch[k][0], write[k][0], 1)
        hbm = (2.0 * fetch[k][1] + write[k][1]) * 1024.0
        out["kernels"][k] = {"launches": n, "fetch_kib_raw": fetch[k][1], "write_kib": write[k][1],
                             "hbm_bytes_per_launch": hbm / n}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{k:28s} launches {v['launches']:5d}  HBM/launch {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
