#!/bin/bash
# rocprofv3 kernel stats of the ragged batch alone (row-space form, default settings): 3 warm-up + 5 timed steps of ragged_sweep.py
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r05
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rag -- python3 $R/scripts/dev/ragged_sweep.py rows:4096:2 > /tmp/rag.log 2>&1; echo "rc $?"
cp $(find /tmp/rag -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r05/kernel_stats_ragged.csv
grep "row space" /tmp/rag.log | cut -c1-100
