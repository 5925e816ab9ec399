"""Debug aid: the pinned-drop-path training step in f32 vs bf16x3 mode vs the reference golden, per parameter."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
os.environ.setdefault("VRDONE_TEST_DEVICE", "cuda")
from conftest import GOLDEN, load_case
from golden_cases import replay_matching, train_batch
from oracle import vrd_oracle as O
from vrdone_amd import ops
from vrdone_amd.models.blocks import AffineDropPath
from vrdone_amd.models.maskvrd import MaskVRD
DEV = "cuda"
mc, _, keys = load_case("vidvrd")
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
g = np.load(os.path.join(GOLDEN, "train_step_vidvrd.npz"))
from vrdone_amd import configs
c_in = configs.input_channels(mc)
only = set(sys.argv[1:])          # module names whose drop path stays pinned; others -> all kept
def grads(mode, case="pinned"):
    ops.set_precision(mode)
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]), strict=True)
    model = model.to(DEV).train()
    lens, _, _, data = train_batch(mc, c_in, device=DEV)
    for name, mod in model.named_modules():
        if isinstance(mod, AffineDropPath):
            if case == "nodrop": mod.drop_prob = 0.0
            else: mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
    replay_matching(model, meta["cases"][case]["indices"])
    with torch.enable_grad():
        loss = model(data); loss["total_loss"].backward()
    return {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
stride = meta["sample_stride"]
for case in ("pinned",):
    a, b = grads("f32", case), grads("bf16x3", case)
    print("== case", case)
    for n in a:
        want = g[f"{case}/{n}"]
        sa = (a[n] if a[n].numel() <= 2048 else a[n].flatten()[::stride]).numpy().astype(np.float64)
        sb = (b[n] if b[n].numel() <= 2048 else b[n].flatten()[::stride]).numpy().astype(np.float64)
        nw = np.linalg.norm(want) + 1e-12
        ea, eb, eab = np.linalg.norm(sa - want) / nw, np.linalg.norm(sb - want) / nw, np.linalg.norm(sa - sb) / nw
        if max(ea, eb) > 5e-4:
            print(f"{n:60s} f32-vs-ref {ea:.1e}  x3-vs-ref {eb:.1e}  f32-vs-x3 {eab:.1e}")
print("keep vectors with a dropped sample:", {k: v for k, v in meta["keep"].items() if min(v) == 0})
