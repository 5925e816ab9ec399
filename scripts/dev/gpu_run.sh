#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_g24; mkdir -p $O
export GEMM_LAB_F16=1 GEMM_LAB_SHAPES=2 GEMM_LAB_SUSTAIN_MS=1200
for rep in 1 2; do for b in tree gm4 gm8 gm16; do echo "== $b" | tee -a $O/lab.txt; timeout -k 10 120 scripts/lab/r06/bin/gemm6_$b 0 2>&1 | grep -v "^lab mode\|consumer\|producer\|tile start" | tee -a $O/lab.txt; done; done
