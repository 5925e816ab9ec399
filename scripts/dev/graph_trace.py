"""Dev aid: first library op whose output differs between an eager run and a graph replay of backbone.cl."""
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
log = []
def leaves(o):
    out = []
    def rec(v):
        if torch.is_tensor(v): out.append(v)
        elif hasattr(v, "t") and hasattr(v, "width"): out.append(v.t)
        elif hasattr(v, "float") and not isinstance(v, (int, float)): out.append(v.float())
        elif isinstance(v, (list, tuple)): [rec(t) for t in v]
    rec(o); return out
def wrap(name):
    fn = getattr(ops, name)
    def inner(*a, **k):
        out = fn(*a, **k)
        log.append((name, k.keys() if name == "conv_gemm" else "", leaves(out)))
        return out
    setattr(ops, name, inner)
for n in ("conv_gemm", "conv_gemm_batch", "layernorm", "dwconv_ln", "local_attention", "attention", "maxpool_mask", "bct_to_btc", "mask_head"):
    wrap(n)
with torch.no_grad():
    model.backbone.cl(x, m2); eager = list(log); log.clear()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): model.backbone.cl(x, m2)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize(); log.clear()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): model.backbone.cl(x, m2)
    graph = list(log)
    g.replay(); torch.cuda.synchronize()
print(len(eager), len(graph))
shown = 0
for i, ((n1, k1, o1), (n2, k2, o2)) in enumerate(zip(eager, graph)):
    d = [float((a.float() - b.float()).abs().max()) if a.shape == b.shape else -1 for a, b in zip(o1, o2)]
    if max(d + [0]) != 0 or len(o1) != len(o2):
        print(i, n1, list(k1), [tuple(a.shape) for a in o1], d); shown += 1
        if shown >= 6: break
