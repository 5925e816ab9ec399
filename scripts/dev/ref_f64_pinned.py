"""Debug aid (build container only): the REFERENCE's pinned-drop-path training step in float64 vs its float32 golden --
how far the golden's f32 gradients are from the exact ones, per parameter."""
import os, sys
os.environ.setdefault("PYTORCH_JIT", "0")
import json
import re
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import make_golden_train as M
from make_golden import OUT, build, load_cfg
from models import blocks as ref_blocks
torch.set_grad_enabled(True)
cfg, mc = load_cfg("vidvrd.yaml")
model, _, _ = build(mc)
model = model.double()
for p in model.parameters(): p.requires_grad_(True)
lens, data = M.batch(mc)
data = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in data.items()}
meta = json.load(open(os.path.join(OUT, "train_step_vidvrd.json")))
g = np.load(os.path.join(OUT, "train_step_vidvrd.npz"))
case = "pinned"
subset = sys.argv[1] if len(sys.argv) > 1 else ""
if case == "nodrop":
    ref_blocks.drop_path = lambda x, drop_prob=0.0, training=False: x
else:
    keeps = {}
    for name, mod in model.named_modules():
        if isinstance(mod, ref_blocks.AffineDropPath) and mod.drop_prob > 0:
            keeps[name] = torch.tensor(meta["keep"][name], dtype=torch.float64)
            if subset and not re.match(subset, name):
                keeps[name] = torch.ones_like(keeps[name]) * (1.0 - mod.drop_prob)      # factor 1: this module drops nothing
            state = {"calls": 0}
            def fwd(x, mod=mod, name=name, state=state):
                n = x.shape[0]
                k = keeps[name][state["calls"] * n:(state["calls"] + 1) * n]
                state["calls"] += 1
                return (mod.scale * x).div(1.0 - mod.drop_prob) * k.view(n, *([1] * (x.dim() - 1)))
            mod.forward = fwd
# replay the golden's matching so that the loss function is the same
rec = meta["cases"][case]["indices"]
calls = {"n": 0}
real = model.bipartite_match
def match(*a, **kw):
    idx, lm = real(*a, **kw)
    want = rec[calls["n"]]; calls["n"] += 1
    return [(torch.tensor(i), torch.tensor(j)) for i, j in want], lm
model.bipartite_match = match
# gradients arriving at every block's output (float64 ground truth for scripts/dev/pinned_bisect4.py)
inter = {}
def hook_out(name):
    def fwd_hook(mod, inp, out):
        y = out[0] if isinstance(out, (tuple, list)) else out
        if torch.is_tensor(y) and y.requires_grad:
            n = sum(k.startswith(name + "#") for k in inter_keys)
            key = f"{name}#{n}"; inter_keys.append(key)
            y.register_hook(lambda g, key=key: inter.__setitem__(key, g.detach().clone()))
    return fwd_hook
inter_keys = []
for name, mod in model.named_modules():
    if re.fullmatch(r"backbone\.(stem|branch|s_attn|o_attn)\.\d+", name) or re.fullmatch(r"predictor\.transformer\.decoder\.layers\.\d+", name):
        mod.register_forward_hook(hook_out(name))
loss = M.run(model, data)
print("total_loss f64", float(loss["total_loss"]), "golden f32", meta["cases"][case]["losses"]["total_loss"])
stride = meta["sample_stride"]
errs = {}
for name, p in model.named_parameters():
    gg = p.grad.detach()
    got = (gg if gg.numel() <= 2048 else gg.flatten()[::stride]).numpy()
    want = g[f"{case}/{name}"].astype(np.float64)
    errs[name] = float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-4 * max(s[2] for s in meta["cases"][case]["grad_stats"].values())))
v = np.array(list(errs.values()))
print(f"reference f32 golden vs reference f64: percentiles 50/90/99/max {np.percentile(v,50):.2e} {np.percentile(v,90):.2e} {np.percentile(v,99):.2e} {v.max():.2e}; > 1e-3: {(v>1e-3).sum()}")
for n in ["backbone.branch.0.attn.query.weight", "backbone.branch.0.attn.query_norm.weight", "backbone.stem.0.attn.query.weight", "backbone.visual_embd.0.conv.weight", "backbone.branch.0.mlp.3.weight", "backbone.branch.0.attn.proj.weight"]:
    print(f"  {n}: {errs[n]:.2e}")
np.savez_compressed(os.path.join(os.path.dirname(OUT), "..", "scripts", "lab", "libs", "ref_f64_inter_%s.npz" % (re.sub(r"\W", "_", subset) or "all")),
                    **{k: v.numpy() for k, v in inter.items()})
print("intermediate gradients:", {k: tuple(v.shape) for k, v in list(inter.items())[:3]}, len(inter))
np.savez_compressed(os.path.join(os.path.dirname(OUT), "..", "scripts", "lab", "libs", "ref_f64_%s.npz" % (re.sub(r"\W", "_", subset) or "all")), **{n: (p.grad if p.grad.numel() <= 2048 else p.grad.flatten()[::stride]).numpy() for n, p in model.named_parameters()})
