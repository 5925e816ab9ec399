#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g28; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu --durations=12 > $O/gpu_tests.txt 2>&1; echo "pytest rc $?"; tail -20 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
