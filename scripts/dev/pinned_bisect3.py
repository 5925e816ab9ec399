"""Debug aid: our gradients (both modes) vs float64 reference gradients for subsets of pinned drop-path modules."""
import json, os, re, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case
from golden_cases import replay_matching, train_batch
from oracle import vrd_oracle as O
from vrdone_amd import ops, configs
from vrdone_amd.models.blocks import AffineDropPath
from vrdone_amd.models.maskvrd import MaskVRD
DEV = "cuda"
mc, _, keys = load_case("vidvrd")
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
c_in = configs.input_channels(mc)
stride = meta["sample_stride"]
def grads(mode, subset):
    ops.set_precision(mode)
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]), strict=True)
    model = model.to(DEV).train()
    lens, _, _, data = train_batch(mc, c_in, device=DEV)
    for name, mod in model.named_modules():
        if isinstance(mod, AffineDropPath):
            if re.match(subset, name): mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
            else: mod.drop_prob = 0.0
    replay_matching(model, meta["cases"]["pinned"]["indices"])
    with torch.enable_grad():
        loss = model(data); loss["total_loss"].backward()
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}, float(loss["total_loss"].detach())
for subset in sys.argv[1:]:
    ref = np.load(os.path.join(REPO, "scripts", "lab", "libs", "ref_f64_%s.npz" % re.sub(r"\W", "_", subset)))
    big = max(float(np.linalg.norm(ref[n])) for n in ref.files)
    for mode in ("f32", "bf16x3"):
        gr, loss = grads(mode, subset)
        errs = {}
        for n, gg in gr.items():
            got = (gg if gg.numel() <= 2048 else gg.flatten()[::stride]).numpy()
            errs[n] = float(np.linalg.norm(got - ref[n]) / (np.linalg.norm(ref[n]) + 1e-4 * big))
        v = np.array(list(errs.values()))
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
        print(f"{subset:22s} {mode:7s} loss {loss:.5f}: 50/90/99/max {np.percentile(v,50):.1e} {np.percentile(v,90):.1e} {np.percentile(v,99):.1e} {v.max():.1e}  >1e-3: {(v>1e-3).sum():3d}  worst: " + ", ".join(f"{k.replace('backbone.','')}={e:.1e}" for k, e in worst), flush=True)
if os.environ.get("DETAIL"):
    subset = os.environ["DETAIL"]
    ref = np.load(os.path.join(REPO, "scripts", "lab", "libs", "ref_f64_%s.npz" % re.sub(r"\W", "_", subset)))
    for mode in ("f32", "bf16x3"):
        gr, _ = grads(mode, subset)
        for n in ("backbone.branch.0.attn.query_norm.weight", "backbone.branch.0.attn.query.bias", "backbone.branch.0.ln1.weight"):
            d = gr[n].numpy().ravel() - ref[n].ravel()
            r = ref[n].ravel()
            idx = np.argsort(-np.abs(d))[:8]
            print(mode, n, "norm ref", np.linalg.norm(r), "norm diff", np.linalg.norm(d), "top |diff| idx", idx.tolist(), "vals", [f"{d[i]:.2e}/{r[i]:.2e}" for i in idx], "corr(diff, ref) =", float(np.dot(d, r) / (np.linalg.norm(d) * np.linalg.norm(r) + 1e-30)))
