#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g17; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $O/gpu_tests.txt
