"""Dev aid: which aten ops launch fill kernels during one training step (torch.profiler, grouped by op and python caller)."""
import os, sys, collections, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch, param_groups
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
from torch.profiler import profile, ProfilerActivity
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
opt = torch.optim.AdamW(param_groups(model, 0.05), lr=1e-5)
def step():
    loss = model(data)["total_loss"]; opt.zero_grad(set_to_none=True); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step()
step(); step()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ("aten::zeros", "aten::zero_", "aten::fill_", "aten::zeros_like", "aten::full", "aten::new_zeros", "aten::ones_like", "aten::full_like"):
        st = [s for s in (e.stack or []) if "site-packages" not in s and "dist-packages" not in s][:1]
        par = e.cpu_parent.name if e.cpu_parent is not None else "-"
        c[(e.name, par, st[0][-70:] if st else "")] += 1
print(sum(c.values()))
for k, v in c.most_common(30): print(v, k)
