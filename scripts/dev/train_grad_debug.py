"""Debug aid: per-parameter gradient error of the HIP training step vs the reference golden."""
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

mode, case = sys.argv[1], sys.argv[2]
ops.set_precision(mode)
mc, ic, keys = load_case("vidvrd")
sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
model = MaskVRD(mc, device="cuda"); model.load_state_dict(sd); model = model.cuda().train()
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
g = np.load(os.path.join(GOLDEN, "train_step_vidvrd.npz"))
cin = 2 * mc["visual_dim"] + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]
lens, _, _, data = train_batch(mc, cin, device="cuda")
for name, mod in model.named_modules():
    if isinstance(mod, AffineDropPath):
        if case == "nodrop": mod.drop_prob = 0.0
        else: mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
agreed = replay_matching(model, meta["cases"][case]["indices"])
loss = model(data)
print("pairs where own matching differs:", agreed, [lens[n] for c in agreed for n in c])
loss["total_loss"].backward()
want = meta["cases"][case]["losses"]
for k, v in want.items():
    print(f"{k:16s} got {float(loss[k]):.6f} want {v:.6f} rel {abs(float(loss[k])-v)/max(1,abs(v)):.2e}")
stats = meta["cases"][case]["grad_stats"]
biggest = max(s[2] for s in stats.values())
rows = []
for name, p in model.named_parameters():
    w = g[f"{case}/{name}"]
    got = (p.grad if p.grad.numel() <= 2048 else p.grad.flatten()[::meta["sample_stride"]]).detach().cpu().numpy()
    err = np.linalg.norm(got.astype(np.float64) - w) / (np.linalg.norm(w) + 1e-4 * biggest)
    rows.append((err, name, float(np.linalg.norm(w))))
rows.sort(reverse=True)
for r in rows[:25]: print(f"{r[0]:.3e}  {r[1]:60s} |ref| {r[2]:.3e}")
with open(os.path.join(REPO, "gpurun_out", f"tgd_all_{mode}_{case}.txt"), "w") as f:
    for r in sorted(rows, key=lambda r: r[1]): f.write(f"{r[0]:.3e}  {r[1]}\n")
print("median", np.median([r[0] for r in rows]), "biggest norm", biggest)

# ---- where inside visual_embd.0.conv.weight does the error sit?
name = "backbone.visual_embd.0.conv.weight"
p = dict(model.named_parameters())[name]
w = g[f"{case}/{name}"]
got = p.grad.flatten()[::meta["sample_stride"]].detach().cpu().numpy().astype(np.float64)
idx = np.arange(got.size) * meta["sample_stride"]
n, ci, k = np.unravel_index(idx, tuple(p.shape))
d = got - w
for kk in range(3):
    s = k == kk
    print("tap", kk, "err", np.linalg.norm(d[s]) / np.linalg.norm(w[s]))
for lo in range(0, 1024, 128):
    s = (ci >= lo) & (ci < lo + 128)
    print("ci", lo, "err", np.linalg.norm(d[s]) / np.linalg.norm(w[s]))
for lo in range(0, 512, 64):
    s = (n >= lo) & (n < lo + 64)
    print("n", lo, "err", np.linalg.norm(d[s]) / np.linalg.norm(w[s]))
worst = np.argsort(-np.abs(d))[:8]
print("largest abs errors:", [(int(n[i]), int(ci[i]), int(k[i]), float(got[i]), float(w[i])) for i in worst])

for name in ("backbone.branch.0.attn.query_norm.weight", "backbone.stem.0.attn.key_norm.weight", "backbone.visual_embd_norm.0.bias"):
    p = dict(model.named_parameters())[name]
    w = g[f"{case}/{name}"].reshape(-1)
    got = p.grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
    d = np.abs(got - w)
    order = np.argsort(-d)[:6]
    print(name, "total rel", np.linalg.norm(got - w) / np.linalg.norm(w), "top:", [(int(i), float(got[i]), float(w[i])) for i in order],
          "share of top-6 in err^2:", float((d[order] ** 2).sum() / (d ** 2).sum()))
