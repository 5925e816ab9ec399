# phase stagger of the 256 x 256 GEMM (VRD_BIG_STAGGER = units of 1,024 cycles per CU slot, 32 slots per XCD)
for st in 0 1 2 4 0 8; do
VRD_BIG_STAGGER=$st python bench.py --steps 8 --warmup 2 --no-alt --no-ragged --no-forward-test --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_st.json 2> gpurun_out/b_st.err
python -c "
import json; d=json.load(open('gpurun_out/b_st.json')); k=d['kernel_ms_per_step']
print('stagger $st: step', round(d['ms_per_step'],2), 'gemm_big', k['gemm_x3_big'], 'frac', round(d['roofline']['frac'],4))"
done
