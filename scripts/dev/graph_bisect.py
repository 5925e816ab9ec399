"""Dev aid: which stage of _mask_vrd changes under graph capture (no_grad inference path)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth, ops
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
x, m = model._train_batch(data["so_features_list"])
m2 = m.reshape(m.shape[0], -1).contiguous()
def leaves(o):
    out = []
    def rec(v):
        if torch.is_tensor(v): out.append(v)
        elif hasattr(v, "float") and not isinstance(v, (int, float)): out.append(v.float())
        elif isinstance(v, dict): [rec(t) for t in v.values()]
        elif isinstance(v, (list, tuple)): [rec(t) for t in v]
    rec(o); return out
def check(name, fn):
    want = leaves(fn())
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = fn()
    g.replay(); torch.cuda.synchronize()
    print(f"{name:24s}", [round(float((a.float() - b.float()).abs().max()), 5) for a, b in zip(want, leaves(out))], flush=True)
    return out
with torch.no_grad():
    bb = model.backbone
    feats, masks = bb.cl(x, m2)
    check("backbone.cl", lambda: bb.cl(x, m2))
    check("neck.cl", lambda: model.neck.cl(feats, masks))
    fpn_feat, _ = model.neck.cl(feats, masks)
    check("predictor.cl", lambda: model.predictor.cl(feats[-1], fpn_feat, masks[-1], masks[0], with_aux=True))
    check("_mask_vrd", lambda: model._mask_vrd(x, m, with_aux=True))
