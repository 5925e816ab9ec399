"""Ragged leg of bench.py (2048 pairs, lengths U[2, 256], fresh mask per step) over bucket floor x side streams, one process."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth, ops
from vrdone_amd.models.maskvrd import MaskVRD
from bench import padded_len

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
pairs, frames = 2048, 256
t_pad = padded_len(cfg, frames, model.max_div_factor)
gen = torch.Generator().manual_seed(1235)
lens = torch.randint(2, frames + 1, (pairs,), generator=gen)
lens[0] = frames
x, m = synth.synth_pairs(pairs, configs.input_channels(cfg), t_pad, lens.tolist(), seed=1234, device=dev)
ref = None
# arguments: "min_rows:streams" = bucket by bucket, "rows:min_rows" = all buckets in one row space
combos = [c.split(":") for c in sys.argv[1:]] or [["262144", "1"]]
from vrdone_amd.models import ragged
with torch.no_grad():
    for a, b, *rest in combos:
        ragged.ATTN_LANES = int(rest[0]) if rest else ragged.ATTN_LANES          # "rows:min_rows:lanes"
        if a == "rows":
            model.row_space, model.ROWS_MIN_ROWS, min_rows, streams = True, int(b), int(b), 0
        else:
            model.row_space, model.TIGHT_MIN_ROWS, model.TIGHT_STREAMS, min_rows, streams = False, int(a), int(b), int(a), int(b)
        model.__dict__.pop("_tight_stream_pool", None)
        for _ in range(3):
            out = model._mask_vrd(x, m.clone(), with_aux=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 5
        for _ in range(K):
            out = model._mask_vrd(x, m.clone(), with_aux=False)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / K
        plan = model._tight_plan(m.clone(), m.reshape(pairs, -1))["buckets"]
        res = (out["pred_logits"].clone(), out["pred_masks"].clone())
        if ref is None:
            ref = res
        same = all(torch.equal(a, b) for a, b in zip(ref, res))
        dl = max(float((a - b).abs().max()) for a, b in zip(ref, res))
        print(f"{'row space' if streams == 0 else 'buckets  '} min_rows {min_rows:7d} streams {streams} attention lanes {ragged.ATTN_LANES}: {ms:6.1f} ms/step, buckets {[(t, n) for t, _, n, _ in plan]}, "
              f"{'bit-equal to the first setting' if same else 'max |diff| vs first %.3g' % dl}", flush=True)
