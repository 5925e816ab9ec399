#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_attn; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -k "flash or attention or oracle" > $O/tests.txt 2>&1; echo "pytest rc $?"; tail -3 $O/tests.txt
grep -q passed $O/tests.txt || { tail -40 $O/tests.txt; exit 1; }
rm -f $O/sweep.txt
for T in 64 96 128; do echo "hd128 T=$T" >> $O/sweep.txt; timeout -k 10 120 python scripts/flash_bench.py --pair --heads 4 --hd 128 --T $T --valid $((T-6)) --B 2048 2>&1 | grep "w32 again" >> $O/sweep.txt; done
for T in 128 288 512; do echo "hd64 T=$T" >> $O/sweep.txt; timeout -k 10 120 python scripts/flash_bench.py --pair --heads 8 --hd 64 --T $T --valid $((T-6)) --B 1024 2>&1 | grep "w32 again" >> $O/sweep.txt; done
cat $O/sweep.txt
