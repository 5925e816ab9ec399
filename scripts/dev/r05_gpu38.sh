#!/bin/bash
# the N = 2 path of bench.py with two ranks sharing the one GPU over gloo (plumbing rehearsal, not a scaling number)
mkdir -p gpurun_out/r05
cd /root/repo
BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-alt > gpurun_out/r05/bench_n2_rehearsal_gloo.json 2> gpurun_out/r05/bench_n2.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r05/bench_n2_rehearsal_gloo.json')); print(d['n_gpus'], d['scaling'], round(d['value']), round(d['ms_per_step'],1), d['roofline']['frac'], d['config'])"
