#!/bin/bash
# round-5 GPU call 9: absmax fix (training step timing), tile set-up by multiplication (lab + step), tests
mkdir -p gpurun_out/r05
cd /root/repo
echo "== tests: gemm ops, backward"
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/r05/t9.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t9.txt
echo "== vidor-size training step (48 pairs x 512 frames)"
for fb in 1 0 1 0; do
  echo "-- VRDONE_F16_BACKWARD=$fb"; VRDONE_F16_BACKWARD=$fb timeout -k 10 300 python scripts/train_step.py --config vidor --pairs 48 --steps 6 2>&1 | grep "^step [345]"
done
echo "== lab"
GEMM_LAB_F16=1 timeout -k 10 120 scripts/lab/r05/gemm5_lab_dma1 0 | grep -v "consumer 0\|producer 0" | grep -A3 "chunk1024\|mlp up"
echo "== step"
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-alt --no-ragged --no-cpu-baseline --no-forward-test --no-train-step --no-shard-projection 2>/dev/null > gpurun_out/r05/b9.json
python -c "import json,sys; d=json.load(open('gpurun_out/r05/b9.json')); k=d['kernel_ms_per_step']; print('step', round(d['ms_per_step'],2), 'ms  big', round(k.get('gemm_x3_big',0),2), 'frac', round(d['roofline']['frac'],4))"
done
echo "== model tests"
timeout -k 10 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r05/t9m.txt 2>&1; echo "rc $?"; tail -3 gpurun_out/r05/t9m.txt
