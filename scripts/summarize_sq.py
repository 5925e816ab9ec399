#!/usr/bin/env python3
"""Summarise the SQ pass of scripts/collect_profiles.sh (one rocprofv3 --pmc pass, counters in their own run with
--kernel-trace only) per kernel:

    python scripts/summarize_sq.py gpurun_out/profiles_r02 profiles/r02_sq_counters.json

Per kernel (sums over its dispatches in one bench step):
  mfma_busy_frac      SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs-worth of busy CU cycles): SQ_VALU_MFMA_BUSY_CYCLES counts cycles
                      with an MFMA executing, summed over SIMDs; SQ_BUSY_CU_CYCLES counts quad-cycles... (units per
                      MI355X_MICROARCH.md 'rocprofv3 PMC slots' and the cycle-constants table: BUSY_CYCLES in cycles,
                      SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CU_CYCLES in quad-cycles)
  mfma_util_vs_time   SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 256 CUs x 4 SIMDs), clock from GRBM_GUI_ACTIVE / 8 /
                      duration (the guide's effective-clock recipe; reads high on dispatches < 0.3 ms)
  wait fractions      SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_WAIT_INST_LDS, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  mfma_mops           SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F16 / _F32 by mode (units of 512 FLOP-ish "MOPS" as the counter defines them; reported raw)
"""
import collections
import csv
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
N_CU, N_SIMD = 256, 4


def main(src, dst, suffix=""):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    with open(f"{src}/pmc_SQ{suffix}.csv") as f:
        for r in csv.DictReader(f):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen[k]:
                seen[k].add(r["Dispatch_Id"])
                agg[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                agg[k]["_vgpr"] = max(agg[k]["_vgpr"], float(r["VGPR_Count"]) + float(r["Accum_VGPR_Count"]))
    import bench
    sha_file = os.path.join(src, "kernel_src_sha.txt")
    out = {"source": "rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES "
                     "SQ_INSTS_VALU_MFMA_MOPS_<BF16|F16|F32 by mode> SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE "
                     "-- python3 bench.py --steps 1 --warmup 1 --precision " + (suffix[1:] if suffix else "bf16x3") + " (all dispatches of the process: warm-up + 1 step)",
           "kernel_src_sha": open(sha_file).read().strip() if os.path.exists(sha_file) else bench.kernel_source_sha(),
           "units": "SQ_VALU_MFMA_BUSY_CYCLES in cycles (summed over SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in "
                    "quad-cycles summed over waves; clock = GRBM_GUI_ACTIVE / 8 / duration",
           "kernels": {}}
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["_ns"]):
        ns = c["_ns"]
        if ns <= 0 or not c.get("SQ_WAVE_CYCLES"):
            continue
        clock_ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / ns
        simd_cycles = ns * clock_ghz * N_CU * N_SIMD
        wave = c["SQ_WAVE_CYCLES"]
        out["kernels"][k] = {
            "dispatches": len(seen[k]), "total_ms": ns / 1e6, "vgprs": int(c["_vgpr"]), "clock_ghz": round(clock_ghz, 3),
            "mfma_util_vs_time": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
            "mfma_mops": sum(v for n, v in c.items() if n.startswith("SQ_INSTS_VALU_MFMA_MOPS_")),
            "busy_cu_frac": round(4.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0) / (ns * clock_ghz * N_CU), 4),
            "wait_any_frac": round(c["SQ_WAIT_ANY"] / wave, 4), "wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / wave, 4),
            "wait_inst_lds_frac": round(c["SQ_WAIT_INST_LDS"] / wave, 4), "active_inst_frac": round(c["SQ_ACTIVE_INST_ANY"] / wave, 4)}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in list(out["kernels"].items())[:10]:
        print(f"{k:28s} {v['total_ms']:8.2f} ms  clock {v['clock_ghz']:.2f} GHz  MFMA util {v['mfma_util_vs_time']:.3f}  "
              f"wait_any {v['wait_any_frac']:.2f} wait_inst {v['wait_inst_any_frac']:.2f} (lds {v['wait_inst_lds_frac']:.2f}) active {v['active_inst_frac']:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
