#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_attn; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2; do for lib in prev base; do
  if [ $lib = base ]; then unset VRDONE_HIP_LIB; else export VRDONE_HIP_LIB=$PWD/scripts/lab/libs/libvrdone_prev.so; fi
  for prec in bf16x3 f16x3; do
  echo "== $lib $prec hd64 T=512 / hd64 T=288 / hd128 T=128" >> $O/ab.txt
  FB_PREC=$prec timeout -k 10 120 python scripts/flash_bench.py --pair --heads 8 --hd 64 --T 512 --valid 506 --B 1024 2>&1 | grep "w32 again" >> $O/ab.txt
  FB_PREC=$prec timeout -k 10 120 python scripts/flash_bench.py --pair --heads 8 --hd 64 --T 288 --valid 282 --B 1024 2>&1 | grep "w32 again" >> $O/ab.txt
  FB_PREC=$prec timeout -k 10 120 python scripts/flash_bench.py --pair --heads 4 --hd 128 --T 128 --valid 122 --B 2048 2>&1 | grep "w32 again" >> $O/ab.txt
  done
done; done
cat $O/ab.txt
