#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of scripts/collect_profiles.sh (FETCH_SIZE, WRITE_SIZE; one TCC pass
each, as MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE is in KiB and, on gfx950, reports exactly half of the bytes of wide coalesced reads (16 bytes per lane,
global_load and LDS-DMA alike; checked here: the LayerNorm kernel reads what it writes and its FETCH_SIZE is half its
WRITE_SIZE), hence the factor 2.  Kernels whose reads are NOT wide (NARROW below: dword-per-lane or scalar reads)
are outside that calibration: their figure uses the raw FETCH_SIZE and is marked "uncalibrated".
The summary is stamped with the hash of the kernel sources it was collected on (bench.kernel_source_sha) and the git
commit, so bench.py can refuse a summary that describes other kernels.

    python scripts/summarize_traffic.py gpurun_out/profiles_r02 profiles/r02_hbm_traffic.json
"""
import collections
import csv
import json
import re
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
# kernels that read a dword (or less) per lane: the gfx950 "FETCH_SIZE counts half" rule is calibrated for 16-byte reads only
NARROW = {"mask_head_kernel", "postprocess_kernel", "bct_to_btc_kernel", "row_blocks_kernel", "attn_small_kernel"}


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


def main(src, dst, suffix=""):
    """suffix: "_<mode>" of the passes (collect_profiles.sh names; "" = the bf16x3 passes of rounds 1-3)."""
    fetch, write = per_kernel(f"{src}/pmc_FETCH_SIZE{suffix}.csv"), per_kernel(f"{src}/pmc_WRITE_SIZE{suffix}.csv")
    import bench
    sha_file = os.path.join(src, "kernel_src_sha.txt")        # written on the GPU box by collect_profiles.sh
    src_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else bench.kernel_source_sha()
    try:
        git_sha = subprocess.run(["git", "-C", REPO, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", REPO, "status", "--porcelain", "--", "vrdone_amd", "include"],
                                    capture_output=True, text=True).stdout.strip())
    except OSError:
        git_sha, dirty = "", False
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     f"bench.py --steps 1 --warmup 1 --precision {suffix[1:] if suffix else 'bf16x3'}; FETCH_SIZE doubled (gfx950; wide reads only), KiB -> bytes",
           "kernel_src_sha": src_sha, "git_sha": git_sha + ("+uncommitted" if dirty else ""),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        n = max(fetch[k][0], write[k][0], 1)
        wide = k not in NARROW
        hbm = ((2.0 if wide else 1.0) * fetch[k][1] + write[k][1]) * 1024.0
        out["kernels"][k] = {"launches": n, "fetch_kib_raw": fetch[k][1], "write_kib": write[k][1],
                             "hbm_bytes_per_launch": hbm / n, "fetch_doubled": wide}
        if not wide:
            out["kernels"][k]["note"] = "narrow reads: FETCH_SIZE uncalibrated for this access width (MI355X_MICROARCH.md, HBM)"
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]):
        print(f"{k:28s} launches {v['launches']:5d}  HBM/launch {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
