#!/bin/bash
# the GPU suite and smoke on the final tree
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_full; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=5 > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -9 $O/tests.txt
grep -q " passed" $O/tests.txt || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
