#!/bin/bash
# BASELINE config 5 at its size + the vidor-size training step's kernel stats (profiles/r05_bench_vidor_x_4096x512.json, r05_train_step_vidor48_kernel_stats.csv)
mkdir -p gpurun_out/r05
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python bench.py --config vidor_x --pairs 4096 --frames 512 --steps 3 --warmup 1 --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline > gpurun_out/r05/bench_vidor_x_4096x512.json 2> gpurun_out/r05/bench_vidor_x.err; echo "cfg5 rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r05/bench_vidor_x_4096x512.json')); print(d['metric'], round(d['value']), round(d['ms_per_step'],1), d['roofline']['frac'], d['profile']['kernel_ms_per_step'] if 'profile' in d else '', 'ragged', round(d['ragged_variant']['ms_per_step'],1) if d.get('ragged_variant') else None)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trs
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trs -- python3 $R/scripts/train_step.py --config vidor --pairs 48 --steps 4 > /tmp/trs.log 2>&1; echo "train stats rc $?"
cp $(find /tmp/trs -name '*kernel_stats.csv' | head -1) $R/gpurun_out/r05/train_step_vidor48_kernel_stats.csv
grep "^step" /tmp/trs.log
