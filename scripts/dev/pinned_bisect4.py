"""Debug aid: gradients arriving at every block's output, f32 mode vs bf16x3 mode, for a subset of pinned drop-path modules."""
import json, os, re, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case
from golden_cases import replay_matching, train_batch
from oracle import vrd_oracle as O
from vrdone_amd import ops, configs
from vrdone_amd.models import blocks, local_transformer
from vrdone_amd.models.blocks import AffineDropPath
from vrdone_amd.models.maskvrd import MaskVRD
DEV = "cuda"
mc, _, keys = load_case("vidvrd")
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
c_in = configs.input_channels(mc)
subset = sys.argv[1]
store = {}
def wrap(cls):
    orig = cls.cl
    def cl(self, *a, **kw):
        out = orig(self, *a, **kw)
        y = out[0]
        name = getattr(self, '_dbg_name', 'standalone')
        n = store["count"].get(name, 0); store["count"][name] = n + 1
        if y.requires_grad:
            y.register_hook(lambda g, key=f"{name}#{n}": store["g"].__setitem__(key, g.detach().double().cpu()))
        store["y"][f"{name}#{n}"] = y.detach().double().cpu()
        return out
    cls.cl = cl
wrap(blocks.TransformerBlock); wrap(local_transformer.MaskedConvTransformerDecoderLayer)
def run(mode):
    ops.set_precision(mode)
    store.update(g={}, y={}, count={})
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]), strict=True)
    model = model.to(DEV).train()
    for name, mod in model.named_modules():
        mod._dbg_name = name
        if isinstance(mod, AffineDropPath):
            if re.match(subset, name): mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
            else: mod.drop_prob = 0.0
    lens, _, _, data = train_batch(mc, c_in, device=DEV)
    replay_matching(model, meta["cases"]["pinned"]["indices"])
    with torch.enable_grad():
        loss = model(data); loss["total_loss"].backward()
    return dict(store["g"]), dict(store["y"])
ga, ya = run("f32"); gb, yb = run("bf16x3")
for k in ga:
    d = float((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-30))
    dy = float((ya[k] - yb[k]).norm() / (yb[k].norm() + 1e-30))
    print(f"{k:55s} grad f32-vs-x3 {d:.1e}   output f32-vs-x3 {dy:.1e}   |grad| {float(gb[k].norm()):.2e}")
ref = np.load(os.path.join(REPO, "scripts", "lab", "libs", "ref_f64_inter_%s.npz" % re.sub(r"\W", "_", subset)))
print("---- against the reference's float64 gradients at the same points")
for k in ga:
    name = k.split("#")[0]
    parts = [ref[f"{name}#{i}"] for i in range(4) if f"{name}#{i}" in ref.files]
    want = torch.from_numpy(np.concatenate(parts, axis=0)).transpose(1, 2)          # (B, C, T) -> (B, T, C)
    for tag, g in (("f32", ga[k]), ("x3 ", gb[k])):
        if g.shape != want.shape:
            print(k, "shape mismatch", tuple(g.shape), tuple(want.shape)); continue
        print(f"{k:50s} {tag} vs f64 reference: {float((g - want).norm() / want.norm()):.1e}", end="   ")
    print()
# ---- branch.1 alone on the REAL tensors of the f32 run: input = branch.0's output, upstream gradient = what arrived at branch.1's output
import torch.nn.functional as F
print("---- branch.1 standalone on real data")
ops.set_precision("f32")
model = MaskVRD(mc, device=DEV)
sd_all = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
model.load_state_dict(sd_all, strict=True)
model = model.to(DEV).train()
blk = model.backbone.branch[1]
ka = torch.tensor(meta["keep"]["backbone.branch.1.drop_path_attn"][:24], dtype=torch.float32)
km = torch.tensor(meta["keep"]["backbone.branch.1.drop_path_mlp"][:24], dtype=torch.float32)
blk.drop_path_attn.keep, blk.drop_path_mlp.keep = ka, km
lens = torch.tensor(meta["lengths"])
m1 = (torch.arange(48)[None] < ((lens + 1) // 2)[:, None])                       # mask at T/2 (nearest down-sampling of t < len)
x = ya["backbone.branch.0#0"].float()            # (B, T/2, C)
dy = ga["backbone.branch.1#0"].float()
xd = x.to(DEV).requires_grad_(True)
with torch.enable_grad():
    y, _ = blk.cl(xd, m1.to(DEV))
y.backward(dy.to(DEV))
sd64 = {k[len("backbone.branch.1."):]: v.double().requires_grad_(True) for k, v in sd_all.items() if k.startswith("backbone.branch.1.")}
sd64 = {"blk." + k: v for k, v in sd64.items()}
x64 = x.double().transpose(1, 2).contiguous().requires_grad_(True)
def block64(sd, pre, x, mask, n_head, win, stride, ka, km):
    h = O.channel_ln(x, sd[f"{pre}.ln1.weight"], sd[f"{pre}.ln1.bias"])
    a, m = O.local_mhca(sd, f"{pre}.attn", h, mask, n_head, win, stride)
    mf = m.to(x.dtype)
    skip = F.max_pool1d(x, stride + 1, stride, (stride + 1) // 2)
    y = skip * mf + sd[f"{pre}.drop_path_attn.scale"] * a * ka.view(-1, 1, 1)
    h = O.channel_ln(y, sd[f"{pre}.ln2.weight"], sd[f"{pre}.ln2.bias"])
    h = F.conv1d(h, sd[f"{pre}.mlp.0.weight"], sd[f"{pre}.mlp.0.bias"])
    h = F.conv1d(F.gelu(h), sd[f"{pre}.mlp.3.weight"], sd[f"{pre}.mlp.3.bias"])
    return y + sd[f"{pre}.drop_path_mlp.scale"] * (h * mf) * km.view(-1, 1, 1), m
yr, _ = block64(sd64, "blk", x64, m1[:, None], mc["n_head"], mc["n_mha_win_size"][1] if isinstance(mc.get("n_mha_win_size"), list) else 7, 2, ka.double() / 0.9, km.double() / 0.9)
yr.backward(dy.double().transpose(1, 2))
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().norm())
print("out", rel(y.detach(), yr.detach().transpose(1, 2)), " dx", rel(xd.grad, x64.grad.transpose(1, 2)), " |dx|", float(x64.grad.norm()), " |dy|", float(dy.norm()))
want_total = torch.from_numpy(ref["backbone.branch.0#0"]).transpose(1, 2)
neck_true = want_total - x64.grad.transpose(1, 2)
neck_ours = ga["backbone.branch.0#0"] - xd.grad.detach().double().cpu()
print("neck (+ other consumers) contribution to d e1: |true|", float(neck_true.norm()), " |ours|", float(neck_ours.norm()), " |ours - true|", float((neck_ours - neck_true).norm()))
d = (neck_ours - neck_true)
print("error by sequence (l2):", [round(float(d[b].norm()), 5) for b in range(24)])
print("valid lengths at T/2:", ((lens + 1) // 2).tolist())
print("error by frame, summed over sequences:", [round(float(d[:, t].norm()), 5) for t in range(48)])
