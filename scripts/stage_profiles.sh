#!/bin/bash
# After `gpurun -- bash scripts/collect_profiles.sh <tag> both sq`: summarise gpurun_out/profiles_<tag>/ into profiles/<tag>_*
# (the files the judge reads) and check that the traffic stamp matches the tree.
set -e
TAG=${1:-r02}
S=gpurun_out/profiles_$TAG
python scripts/summarize_traffic.py $S profiles/${TAG}_hbm_traffic.json > /dev/null
python scripts/summarize_traffic.py $S profiles/${TAG}_hbm_traffic_f32.json _f32 > /dev/null
python scripts/summarize_sq.py $S profiles/${TAG}_sq_counters.json | head -3
[ -f $S/pmc_SQ_f32.csv ] && python scripts/summarize_sq.py $S profiles/${TAG}_sq_counters_f32.json _f32 | head -2
cp $S/kernel_stats.csv profiles/${TAG}_kernel_stats_bench_n1.csv
cp $S/kernel_stats_f32.csv profiles/${TAG}_kernel_stats_bench_n1_f32.csv
cp $S/bench.json profiles/${TAG}_bench_n1.json
cp $S/bench_under_rocprof.json profiles/${TAG}_bench_n1_under_rocprof.json
cp $S/bench_under_rocprof_f32.json profiles/${TAG}_bench_n1_under_rocprof_f32.json
python - <<PY
import json, bench
d = json.load(open("profiles/${TAG}_hbm_traffic.json"))
print("stamp", d["kernel_src_sha"], "tree", bench.kernel_source_sha(), "OK" if d["kernel_src_sha"] == bench.kernel_source_sha() else "STALE")
b = json.load(open("profiles/${TAG}_bench_n1.json"))
print("bench", round(b["value"]), "pairs/s", round(b["ms_per_step"], 1), "ms  frac", round(b["roofline"]["frac"], 3))
PY
