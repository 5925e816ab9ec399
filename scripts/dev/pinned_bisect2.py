"""Debug aid: which drop-path module makes the f32-mode gradients differ from the bf16x3-mode ones."""
import json, os, sys
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
def grads(mode, pinned):
    ops.set_precision(mode)
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]), strict=True)
    model = model.to(DEV).train()
    lens, _, _, data = train_batch(mc, c_in, device=DEV)
    for name, mod in model.named_modules():
        if isinstance(mod, AffineDropPath):
            if name in pinned: mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
            else: mod.drop_prob = 0.0
    replay_matching(model, meta["cases"]["pinned"]["indices"])
    with torch.enable_grad():
        loss = model(data); loss["total_loss"].backward()
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters()}, float(loss["total_loss"])
probe = ["backbone.branch.0.mlp.3.weight", "backbone.branch.0.attn.query.weight", "backbone.stem.0.attn.query.weight", "backbone.branch.1.mlp.3.weight", "neck.fpn_convs.1.conv.weight"]
names = [n for n in meta["keep"] if n.startswith("backbone.branch") or n.startswith("predictor")] + ["backbone.stem.0.drop_path_attn", "backbone.s_attn.0.drop_path_attn1"]
for nm in [None] + names:
    pinned = set() if nm is None else {nm}
    (a, la), (b, lb) = grads("f32", pinned), grads("bf16x3", pinned)
    d = {p: float((a[p] - b[p]).norm() / (b[p].norm() + 1e-12)) for p in probe}
    print(f"{str(nm):58s} loss f32 {la:.5f} x3 {lb:.5f} | " + "  ".join(f"{v:.1e}" for v in d.values()), flush=True)
