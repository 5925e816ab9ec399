#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  bash scripts/collect_profiles.sh r04 "f16x3 bf16x3 f32" [sq]
# Produces, under gpurun_out/profiles_<tag>/ (scripts/stage_profiles.sh turns them into profiles/<tag>_*):
#   bench.json                        the bench line (events on, CPU baseline on), default mode, alt modes included; run LAST,
#                                     after the traffic summaries of this run were written into profiles/ on the box
#   kernel_stats_<mode>.csv           rocprofv3 --kernel-trace --stats of the same command in one precision mode
#   pmc_FETCH_SIZE_<mode>.csv / pmc_WRITE_SIZE_<mode>.csv   per dispatch, separate passes (TCC slots)
#   pmc_SQ_<mode>.csv                 (with `sq`) one SQ pass per mode: MFMA busy / wave cycles / waits / MFMA ops, every dispatch
#   kernel_src_sha.txt                hash of the kernel sources these were collected on (bench.kernel_source_sha)
# rocprofv3 is always followed directly by `python3 bench.py ...` (no wrapper after `--`), counters in their own passes.
set -o pipefail
TAG=${1:-r04}
MODES=${2:-"f16x3 bf16x3 f32"}
SQ=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.kernel_source_sha())" > $OUT/kernel_src_sha.txt || exit 1
LEAN="--no-cpu-baseline --no-alt --no-ragged --no-forward-test --no-train-step --no-shard-projection"
for MODE in $MODES; do
  SUF="_$MODE"
  echo "[collect] $MODE kernel trace"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$SUF -- python3 $R/bench.py --steps 3 --warmup 1 --precision $MODE $LEAN > $OUT/bench_under_rocprof$SUF.json 2> $OUT/rocprof$SUF.err || exit 1
  cp $OUT/trace$SUF/*/*kernel_stats.csv $OUT/kernel_stats$SUF.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "[collect] $MODE $c"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c$SUF -- python3 $R/bench.py --steps 1 --warmup 1 --precision $MODE $LEAN --no-prof > /dev/null 2> $OUT/pmc_$c$SUF.err || exit 1
    cp $OUT/pmc_$c$SUF/*/*counter_collection.csv $OUT/pmc_$c$SUF.csv
  done
  rm -rf $OUT/trace$SUF $OUT/pmc_FETCH_SIZE$SUF $OUT/pmc_WRITE_SIZE$SUF
  if [ "$SQ" = sq ]; then
    MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16; [ $MODE = f16x3 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_F16; [ $MODE = f32 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32
    echo "[collect] SQ pass $MODE"
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES $MOPS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
      --output-format csv -d $OUT/pmc_SQ$SUF -- python3 $R/bench.py --steps 1 --warmup 1 --precision $MODE $LEAN --no-prof > /dev/null 2> $OUT/pmc_SQ$SUF.err || exit 1
    cp $OUT/pmc_SQ$SUF/*/*counter_collection.csv $OUT/pmc_SQ$SUF.csv
    rm -rf $OUT/pmc_SQ$SUF
  fi
  # this run's traffic summary in place on the box: bench.py reads roofline.traffic from the newest
  # profiles/*_hbm_traffic_<mode>.json whose source stamp matches the tree (stage_profiles.sh writes the same files at home)
  (cd $R && python3 scripts/summarize_traffic.py $OUT profiles/${TAG}_hbm_traffic$SUF.json $SUF > /dev/null)
done
echo "[collect] bench line" && timeout -k 10 600 python3 $R/bench.py --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || exit 1
ls -la $OUT
