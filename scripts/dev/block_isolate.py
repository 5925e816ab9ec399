"""Debug aid: one branch block of the pinned training step in isolation, real activations, HIP vs float64 oracle."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case
from golden_cases import train_batch
from oracle import vrd_oracle as O
from vrdone_amd import ops
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.models.blocks import AffineDropPath
ops.set_precision("f32")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mc, ic, keys = load_case("vidvrd")
sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
lens, x, m, data_cpu = train_batch(mc, 2069)
B = len(lens)
model = MaskVRD(mc, device="cuda"); model.load_state_dict(sd); model = model.cuda().train()
for name, mod in model.named_modules():
    if isinstance(mod, AffineDropPath): mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
data = {k: [t.cuda() for t in v] for k, v in data_cpu.items()}
xd, md = model._train_batch(data["so_features_list"])
with torch.no_grad():
    for mod in model.modules():
        if isinstance(mod, AffineDropPath): mod.drop_prob_saved, mod.drop_prob = mod.drop_prob, mod.drop_prob
from golden_cases import replay_matching
replay_matching(model, meta["cases"]["pinned"]["indices"])
with torch.enable_grad():
    feats, masks = model.backbone.cl(xd, md.reshape(B, -1).contiguous())
    for f in feats: f.retain_grad()
    out = model._heads(feats, masks, True)
    loss = model.criterion(out, data)
    loss["total_loss"].backward()
real_dE = feats[L + 1].grad.detach().clone()
real_dIn = feats[L].grad.detach().clone()
e_in = feats[L].detach().clone().requires_grad_(True)
g = torch.Generator(device="cuda").manual_seed(0)
blk = model.backbone.branch[L]
with torch.enable_grad():
    e_out, m_out = blk.cl(e_in, masks[L])
dE = real_dE if os.environ.get("REAL_DE") else torch.randn(e_out.shape, device="cuda", generator=g) * masks[L + 1][:, :, None]
e_out.backward(dE)
# heads-only part of the input gradient (features detached), to rebuild the total from its two parts
fdet = [f.detach().requires_grad_(True) for f in feats]
del model.bipartite_match
replay_matching(model, meta["cases"]["pinned"]["indices"])
with torch.enable_grad():
    out2 = model._heads(fdet, masks, True)
    loss2 = model.criterion(out2, data)
    loss2["total_loss"].backward()
rebuilt = fdet[L].grad + e_in.grad
perr = sorted([(float((real_dIn[i] - rebuilt[i]).norm() / (rebuilt[i].norm() + 1e-30)), i) for i in range(B)], reverse=True)
print("full-graph total vs (heads part + block part):", [(i, f"{e:.1e}") for e, i in perr[:5]])
# oracle, float64, same factors
dt = torch.float64
pre = f"backbone.branch.{L}"
sd64 = {k: v.to(dt) for k, v in sd.items() if k.startswith(pre)}
for k in list(sd64):
    if "drop_path" in k and k.endswith(".scale"):
        keep = torch.tensor(meta["keep"][k[:-6]], dtype=dt)[:B]
        sd64[k] = sd64[k] * (keep / 0.9).view(B, 1, 1)
xr = e_in.detach().double().cpu().transpose(1, 2).contiguous().requires_grad_(True)
mr = masks[L].cpu()[:, None]
yr, _ = O.transformer_block(sd64, pre, xr, mr, mc["n_head"], mc["n_mha_win_size"], 2)
yr.backward(dE.double().cpu().transpose(1, 2))
a, b = e_in.grad.double().cpu(), xr.grad.transpose(1, 2)
print("out rel err", float((e_out.detach().double().cpu() - yr.detach().transpose(1, 2)).norm() / yr.norm()))
per = sorted([(float((a[i] - b[i]).norm() / (b[i].norm() + 1e-30)), i) for i in range(B)], reverse=True)
print("input-grad worst samples:", [(i, f"{e:.1e}", lens[i]) for e, i in per[:6]])
# the same block inside the full graph: total gradient of its input minus what does not come through the block is not
# available, so compare the block-alone input gradient of the two HIP runs on the worst sample of the full run instead
print("dE stats: per-sample norm", [f"{float(real_dE[i].norm()):.2e}" for i in (5, 21)], "max |dE| at padded rows (sample 21):",
      float((real_dE[21] * (~masks[L + 1][21])[:, None]).abs().max()))
i = per[0][1]
d = (a[i] - b[i]).abs()
rows = d.amax(dim=1)
print("sample", i, "rows with error:", [(t, f"{float(rows[t]):.2e}", f"{float(b[i][t].abs().max()):.2e}") for t in torch.nonzero(rows > 1e-4 * float(b[i].abs().max())).flatten().tolist()][:12],
      "valid rows", int(masks[L][i].sum()))
