#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_ft; mkdir -p $O
timeout -k 10 300 python scripts/dev/eval_graph_probe.py > $O/probe.txt 2>&1; cat $O/probe.txt | tail -8
