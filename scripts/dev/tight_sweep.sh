for cfg in "32 131072" "32 65536" "8 131072" "32 262144" "16 196608"; do
set -- $cfg
VRDONE_TIGHT_UNIT=$1 VRDONE_TIGHT_MIN_ROWS=$2 python bench.py --steps 4 --warmup 1 --no-alt --no-forward-test --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_rag.json 2> gpurun_out/b_rag.err
python -c "
import json; d=json.load(open('gpurun_out/b_rag.json')); r=d['ragged_variant']; k=r['kernel_ms_per_step']
print('unit $1 min_rows $2: headline', round(d['ms_per_step'],1), 'ragged', round(r['ms_per_step'],1), {a:b for a,b in k.items() if b>3})"
done
VRDONE_TIGHT_PADDING=0 python bench.py --steps 4 --warmup 1 --no-alt --no-forward-test --no-train-step --no-cpu-baseline --no-shard-projection > gpurun_out/b_rag.json 2> gpurun_out/b_rag.err
python -c "
import json; d=json.load(open('gpurun_out/b_rag.json')); r=d['ragged_variant']; k=r['kernel_ms_per_step']
print('off: headline', round(d['ms_per_step'],1), 'ragged', round(r['ms_per_step'],1), {a:b for a,b in k.items() if b>3})"
