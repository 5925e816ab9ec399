#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_ft; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_model.py -x -q -k "forward_test or row_space or golden or sharded" > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/tests.txt
grep -q passed $O/tests.txt || { tail -40 $O/tests.txt; exit 1; }
timeout -k 10 300 python scripts/dev/ft_small.py > $O/ft_small.txt 2>&1; cat $O/ft_small.txt
