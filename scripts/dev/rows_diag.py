"""Row-space form vs the batch at its own padded length vs bucket by bucket: per-pair differences, per precision mode; and the
per-family kernel time of one ragged step in both forms."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth, ops, _hip
from vrdone_amd.models.maskvrd import MaskVRD

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
B, T = 300, 288
gen = torch.Generator().manual_seed(177)
lens = torch.randint(2, T - 30, (B,), generator=gen)
lens[:5] = torch.tensor([T - 1, T - 8, T - 9, 2, 33])
x, m = synth.synth_pairs(B, configs.input_channels(cfg), T, lens.tolist(), seed=178, device=dev)
model.ROWS_MIN_ROWS = 2048
model.TIGHT_MIN_ROWS = 2048
with torch.no_grad():
    for mode in ("f32", "f16x3", "bf16x3"):
        ops.set_precision(mode)
        model.tight_padding, model.row_space = True, True
        rows = model._mask_vrd(x, m.clone(), with_aux=False)
        plan = model._tight_plan(m.clone(), m.reshape(B, T))
        model.row_space = False
        buck = model._mask_vrd(x, m.clone(), with_aux=False)
        model.tight_padding = False
        full = model._mask_vrd(x, m.clone(), with_aux=False)
        for name, a, b in (("rows - full", rows, full), ("buckets - full", buck, full), ("rows - buckets", rows, buck)):
            dl = (a["pred_logits"] - b["pred_logits"]).abs().amax(dim=(1, 2))
            dm = (a["pred_masks"] - b["pred_masks"]).abs().amax(dim=(1, 2))
            worst = torch.topk(dl, 6)
            print(f"[{mode}] {name}: logits max {float(dl.max()):.3e} median {float(dl.median()):.3e}; masks max {float(dm.max()):.3e}; "
                  f"worst pairs {[(int(i), int(lens[i]), round(float(v), 6)) for v, i in zip(worst.values, worst.indices)]}")
        if mode == "bf16x3":
            first = {int(idx[0]) for _, idx, _, _ in plan["buckets"]}
            print("   first pair of each bucket:", sorted(first), "buckets", [(t, n) for t, _, n, _ in plan["buckets"]])
