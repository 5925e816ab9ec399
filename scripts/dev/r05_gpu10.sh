#!/bin/bash
mkdir -p gpurun_out/r05
cd /root/repo
for v in NO_MAGIC NO_ZCHAIN; do
echo "== lab $v persist=1"
GEMM_LAB_F16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_$v 0 > gpurun_out/r05/lab_$v.txt 2>&1; echo "rc $?"; grep -c "var 11" gpurun_out/r05/lab_$v.txt; grep "fault" gpurun_out/r05/lab_$v.txt | head -2
done
exit 0
