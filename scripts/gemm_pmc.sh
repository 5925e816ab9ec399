#!/bin/bash
# Stall breakdown of the GEMM kernels on the path's shapes (run through gpurun from the repo root):
#   bash scripts/gemm_pmc.sh [extra gemm_bench.py args]
# Writes gpurun_out/gemm_pmc/{timing.txt,counters.csv}.  One SQ pass (8 slots) + GRBM.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/scripts/gemm_bench.py --pair "$@" > $OUT/timing.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc -- python3 $R/scripts/gemm_bench.py --pair --iters 2 "$@" > /dev/null 2> $OUT/pmc.err || exit 1
cp $OUT/pmc/*/*counter_collection.csv $OUT/counters.csv
rm -rf $OUT/pmc
