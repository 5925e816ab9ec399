"""Debug aid: gradients w.r.t. the backbone's pyramid features (HIP, pinned drop-path) vs the float64 oracle with the same
per-sample factors, sample by sample."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case
from golden_cases import train_batch, replay_matching
from oracle import vrd_oracle as O
from vrdone_amd import ops
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.models.blocks import AffineDropPath
ops.set_precision("f32")
mc, ic, keys = load_case("vidvrd")
sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
lens, x, m, data_cpu = train_batch(mc, 2069)
B = len(lens)
dt = torch.float64
# ---- oracle (CPU f64) with pinned factors
ref_model = MaskVRD(mc, device="cpu").to(dt).train()
replay_matching(ref_model, meta["cases"]["pinned"]["indices"])
class DropSD(dict):
    def __init__(self, base): super().__init__(base); self.calls = {}
    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if k.endswith(".scale") and "drop_path" in k:
            name = k[:-6]; c = self.calls.get(name, 0); self.calls[name] = c + 1
            keep = torch.tensor(meta["keep"][name], dtype=dt)[c * B:(c + 1) * B]
            return v * (keep / 0.9).view(B, 1, 1)
        return v
leaves = {k: (v.to(dt).requires_grad_(True) if v.is_floating_point() and k != "empty_weight" else v.to(dt)) for k, v in sd.items()}
dsd = DropSD(leaves)
xr = x.to(dt)
with torch.enable_grad():
    feats_r, masks_r = O.backbone(dsd, mc, xr.requires_grad_(False), m)
    fr = list(feats_r)
    for f in fr: f.retain_grad()
    fpn, _ = O.neck(dsd, mc, fr, masks_r)
    pred = O.predictor(dsd, mc, fr[-1], fpn, masks_r[-1], masks_r[0])
    data64 = {k: [t.to(dt) if t.is_floating_point() else t for t in v] for k, v in data_cpu.items()}
    loss_r = ref_model.criterion(pred, data64)
    loss_r["total_loss"].backward()
# ---- HIP
model = MaskVRD(mc, device="cuda"); model.load_state_dict(sd); model = model.cuda().train()
for name, mod in model.named_modules():
    if isinstance(mod, AffineDropPath): mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
replay_matching(model, meta["cases"]["pinned"]["indices"])
data = {k: [t.cuda() for t in v] for k, v in data_cpu.items()}
xd, md = model._train_batch(data["so_features_list"])
feats, masks = model.backbone.cl(xd, md.reshape(B, -1).contiguous())
fd = list(feats)
for f in fd: f.retain_grad()
out = model._heads(fd, masks, True)
loss = model.criterion(out, data)
loss["total_loss"].backward()
print("loss", float(loss["total_loss"]), float(loss_r["total_loss"]))
for l in range(4):
    a = fd[l].grad.double().cpu()                 # (B, T_l, C)
    b = fr[l].grad.transpose(1, 2)                # (B, C, T_l) -> (B, T_l, C)
    fa, fb = feats[l].detach().double().cpu(), feats_r[l].detach().transpose(1, 2)
    print(f"level {l}: feature rel err {float((fa-fb).norm()/fb.norm()):.2e}  grad rel err {float((a-b).norm()/b.norm()):.2e}")
    per = [(float((a[i]-b[i]).norm()/(b[i].norm()+1e-30)), i) for i in range(B)]
    per.sort(reverse=True)
    print("   worst samples:", [(i, f"{e:.1e}") for e, i in per[:5]])
    i = per[0][1]
    d = (a[i] - b[i]).abs().amax(dim=1)
    bad_rows = torch.nonzero(d > 1e-4 * float(b[i].abs().max())).flatten().tolist()
    print("   sample", i, "valid rows", int(masks[l][i].sum()), "rows off:", [(t, f"hip {float(a[i][t].abs().max()):.2e}", f"ref {float(b[i][t].abs().max()):.2e}") for t in bad_rows[:8]])
