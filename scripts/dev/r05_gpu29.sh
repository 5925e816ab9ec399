#!/bin/bash
# BASELINE config 5 (vidor_x.yaml, 4096 pairs x 512 frames): round 4's tree beside this one, one box; and the persistent form off
cd /root/repo
L="--config vidor_x --pairs 4096 --frames 512 --steps 3 --warmup 1 --no-forward-test --no-train-step --no-shard-projection --no-cpu-baseline --no-alt --no-ragged"
for rep in 1 2; do
(cd _r04 && timeout -k 10 400 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r04            ', round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['kernel_ms_per_step']['gemm_x3_big'])")
timeout -k 10 400 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05            ', round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['kernel_ms_per_step']['gemm_x3_big'])"
VRD_BIG_PERSIST=0 timeout -k 10 400 python bench.py $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('r05 persist off', round(d['ms_per_step'],1), round(d['roofline']['frac'],4), d['kernel_ms_per_step']['gemm_x3_big'])"
done
