#!/bin/bash
# round 6 collection: scripts/collect_profiles.sh (kernel stats, HBM traffic, SQ counters per mode, the bench line) + BASELINE config 5 at
# its size + forward_test on real-sized videos + the GEMM lab harness on round 5's kernel and this tree's + the vidor-size training step
# + the flash kernel's item stamps
set -o pipefail
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06_collect; mkdir -p $O
bash scripts/collect_profiles.sh r06 "f16x3 bf16x3 f32" sq > $O/collect.log 2>&1 || { tail -5 $O/collect.log; exit 1; }
tail -3 $O/collect.log
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python bench.py --config vidor_x --pairs 4096 --frames 512 --steps 3 --warmup 1 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection > $O/bench_vidor_x_4096x512.json 2> $O/bench_vidor_x.err || exit 1
timeout -k 10 300 python scripts/dev/ft_small.py > $O/ft_small.txt 2>&1
rm -f $O/lab_tile_stamps.txt
export GEMM_LAB_F16=1 GEMM_LAB_SHAPES=31 GEMM_LAB_SUSTAIN_MS=1200
for b in r05 tree r05 tree; do echo "== gemm6_$b" >> $O/lab_tile_stamps.txt; timeout -k 10 120 scripts/lab/r06/bin/gemm6_$b 0 2>&1 | grep -v "^lab mode\|consumer\|producer" >> $O/lab_tile_stamps.txt; done
timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 6 > $O/train_vidor48.txt 2>&1
tail -2 $O/train_vidor48.txt | cut -c1-200
VRDONE_HIP_LIB=$PWD/scripts/lab/libs/libvrdone_stamp.so timeout -k 10 120 python scripts/dev/flash_stamps.py > $O/flash_stamps.txt 2>&1
timeout -k 10 120 python scripts/flash_bench.py --pair > $O/flash_bench.txt 2>&1
grep w64 $O/flash_bench.txt
