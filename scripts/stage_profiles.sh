#!/bin/bash
# After `gpurun -- bash scripts/collect_profiles.sh <tag> "<modes>" sq`: summarise gpurun_out/profiles_<tag>/ into profiles/<tag>_*
# (the files the judge reads) and check that the traffic stamp matches the tree.
set -e
TAG=${1:-r04}
MODES=${2:-"f16x3 bf16x3 f32"}
S=gpurun_out/profiles_$TAG
for MODE in $MODES; do
  [ -f $S/pmc_FETCH_SIZE_$MODE.csv ] || continue
  python scripts/summarize_traffic.py $S profiles/${TAG}_hbm_traffic_$MODE.json _$MODE > /dev/null
  [ -f $S/pmc_SQ_$MODE.csv ] && python scripts/summarize_sq.py $S profiles/${TAG}_sq_counters_$MODE.json _$MODE | head -3
  cp $S/kernel_stats_$MODE.csv profiles/${TAG}_kernel_stats_bench_n1_$MODE.csv
  cp $S/bench_under_rocprof_$MODE.json profiles/${TAG}_bench_n1_under_rocprof_$MODE.json
done
cp $S/bench.json profiles/${TAG}_bench_n1.json
python - <<PY
import json, bench
b = json.load(open("profiles/${TAG}_bench_n1.json"))
mode = b["config"]["gemm_precision"]
d = json.load(open("profiles/${TAG}_hbm_traffic_%s.json" % mode))
print("stamp", d["kernel_src_sha"], "tree", bench.kernel_source_sha(), "OK" if d["kernel_src_sha"] == bench.kernel_source_sha() else "STALE")
print("bench", mode, round(b["value"]), "pairs/s", round(b["ms_per_step"], 1), "ms  frac", round(b["roofline"]["frac"], 3), "traffic", b["roofline"]["traffic"])
PY
