# forward_test / train-step legs of bench.py under the tight-padding switch and the precision modes
for cfg in "1 f16x3" "0 f16x3" "1 bf16x3" "0 bf16x3"; do
set -- $cfg
VRDONE_TIGHT_PADDING=$1 python bench.py --steps 2 --warmup 1 --precision $2 --no-alt --no-ragged --no-cpu-baseline --no-shard-projection > gpurun_out/b_ft.json 2> gpurun_out/b_ft.err
python -c "
import json; d=json.load(open('gpurun_out/b_ft.json')); f=d['forward_test']; t=d['train_step']
print('tight $1 mode $2: forward_test', round(f['ms'],1), 'from tracklets', round(f['from_tracklets']['forward_test_ms'],1), 'train', round(t['ms_forward_backward'],1), round(t['ms_forward_backward_hip_graphs'],1), 'vidor48', round(t['vidor_48x512']['ms_forward_backward'],1))"
done
