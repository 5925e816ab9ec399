"""Where do the once-per-tracklet entity rows differ from the per-pair ones?  (dev aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from conftest import load_case
from golden_cases import PROPOSAL_CASES
from oracle import proposal as P, vrd_oracle as O
from vrdone_amd import ops
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.proposals import prepare_test_proposal

cfg, case, prec = sys.argv[1], sys.argv[2], sys.argv[3]
ops.set_precision(prec)
torch.set_grad_enabled(False)
DEV = "cuda:0"
mc, ic, keys = load_case(cfg)
model = MaskVRD(mc, device=DEV)
model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]))
model = model.to(DEV).eval()
model._config_eval(ic)
bb = model.backbone
vid_kw, dl_kw = PROPOSAL_CASES[case]
raw = P.synth_raw_video(**dict(vid_kw, n_visual=bb.n_visual, n_clip=bb.n_clip))
prop = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
src = prop["pair_source"]
ids = sorted(range(len(src)), key=lambda i: src.lens[i])
shared = model._entity_streams(src, ids)
print("L", shared[2], "reach", shared[3], "streams", tuple(shared[0].shape), "lens", min(src.lens), max(src.lens))
T = -(-max(src.lens) // 96) * 96
sel = torch.tensor(ids, device=DEV)
so, so_box, mask = model._shared_entity_rows(src, sel, shared, 0, T)
vis, clip, so_box2, ent, mask2 = ops.gather_pairs(src, sel, T, bb.n_bbox_so, bb.n_bbox_entity, ops.pair_mode())
want = bb.entity_stage(vis, clip, ent, torch.cat([mask2, mask2], dim=0))
d = (so - want).abs().amax(-1)       # (2B, T)
print("max diff", float(d.max()), "rows differing", int((d > 0).sum()), "of", int(torch.cat([mask, mask]).sum()))
B = len(ids)
for e in range(2 * B):
    n = src.lens[ids[e % B]]
    bad = torch.nonzero(d[e] > 0).flatten().tolist()
    if bad:
        print(f"entity {e} n={n} bad rows {bad[:12]}{'...' if len(bad) > 12 else ''} ({len(bad)}) max {float(d[e].max()):.3g}")
        if e > 6:
            break
