#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash scripts/collect_profiles.sh r01
# Produces, under gpurun_out/profiles_<tag>/ (copy the summaries you want judged into profiles/):
#   bench.json                 the bench line (events on, CPU baseline on)
#   kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch.csv / pmc_write.csv   FETCH_SIZE and WRITE_SIZE per dispatch, separate passes (TCC slots)
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-ragged --no-forward-test > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err || exit 1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --no-ragged --no-forward-test --no-prof > /dev/null 2> $OUT/pmc_$c.err || exit 1
  cp $OUT/pmc_$c/*/*counter_collection.csv $OUT/pmc_$c.csv
done
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
