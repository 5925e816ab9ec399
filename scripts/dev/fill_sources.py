"""Dev aid: who asks for zero-filled tensors during one training step (python-level callers of torch.zeros & co)."""
import os, sys, collections, traceback, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch, param_groups
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
opt = torch.optim.AdamW(param_groups(model, 0.05), lr=1e-5)
def step():
    loss = model(data)["total_loss"]; opt.zero_grad(set_to_none=True); loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step()
step(); step()
counts = collections.Counter()
def wrap(mod, name):
    fn = getattr(mod, name)
    def inner(*a, **k):
        fr = traceback.extract_stack(limit=3)[0]
        counts[(name, os.path.basename(fr.filename), fr.lineno)] += 1
        return fn(*a, **k)
    setattr(mod, name, inner)
for n in ("zeros", "zeros_like", "full", "full_like", "ones", "ones_like"):
    wrap(torch, n)
for n in ("new_zeros", "zero_", "fill_", "new_full"):
    wrap(torch.Tensor, n)
step()
torch.cuda.synchronize()
print(sum(counts.values()), "python-level fill requests")
for k, v in counts.most_common(25): print(v, k)
