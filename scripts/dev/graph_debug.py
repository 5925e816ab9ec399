"""Dev aid: eager vs graph-replayed predictions / gradients of the training network on one batch."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth, train_graph
from vrdone_amd.models.blocks import AffineDropPath
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
for mod in model.modules():
    if isinstance(mod, AffineDropPath): mod.drop_prob = 0.0
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
x, m = model._train_batch(data["so_features_list"])
def flat(p): return [p["pred_logits"], p["pred_masks"]] + [a[k] for a in p["aux_outputs"] for k in ("pred_logits", "pred_masks")]
e1 = flat(model._mask_vrd(x, m, with_aux=True)); e2 = flat(model._mask_vrd(x, m, with_aux=True))
print("eager vs eager", [float((a - b).abs().max()) for a, b in zip(e1, e2)])
model.enable_training_graphs()
for it in range(3):
    g = flat(train_graph.mask_vrd(model, x, m))
    print("graph vs eager", it, [float((a - b).abs().max()) for a, b in zip(e1, g)])

# ---- bisect
from torch import nn
model.enable_training_graphs(False)
def diff(a, b): return [round(float((p - q).abs().max()), 6) for p, q in zip(a, b)]
# V1: inference kernels (no_grad) eager vs captured
with torch.no_grad():
    n1 = flat(model._mask_vrd(x, m, with_aux=True))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): model._mask_vrd(x, m, with_aux=True)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = model._mask_vrd(x, m, with_aux=True)
    g.replay(); torch.cuda.synchronize()
    print("V1 no_grad graph vs eager", diff(n1, flat(out)))
# V2: grad path on aliases, eager
params = list(model.parameters())
aliases = [nn.Parameter(p.detach()) for p in params]
by_id = {id(p): a for p, a in zip(params, aliases)}
slots = [(mod, name, p) for mod in model.modules() for name, p in mod._parameters.items() if id(p) in by_id]
for mod, name, p in slots: mod._parameters[name] = by_id[id(p)]
a1 = flat(model._mask_vrd(x, m, with_aux=True))
print("V2 aliases eager vs eager", diff([t.detach() for t in e1], [t.detach() for t in a1]))
# V3: grad path, forward-only capture on aliases
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): model._mask_vrd(x, m, with_aux=True)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    out3 = model._mask_vrd(x, m, with_aux=True)
g3.replay(); torch.cuda.synchronize()
print("V3 grad-path graph vs eager", diff([t.detach() for t in e1], [t.detach() for t in flat(out3)]))
for mod, name, p in slots: mod._parameters[name] = p
