#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_attn; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -k "flash or attention or oracle" > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/tests.txt
grep -q passed $O/tests.txt || { tail -40 $O/tests.txt; exit 1; }
timeout -k 10 120 python scripts/flash_bench.py --pair > $O/flash.txt 2>&1; grep "w64" $O/flash.txt
VRDONE_HIP_LIB=$PWD/scripts/lab/libs/libvrdone_stamp.so timeout -k 10 120 python scripts/dev/flash_stamps.py > $O/stamps.txt 2>&1
grep -v "it[1-6] " $O/stamps.txt
