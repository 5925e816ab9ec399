"""Dev: where do the device-to-device copies / fills / adds of a training step (forward + backward, vidor.yaml 48 x 512) come
from?  torch profiler with stacks, grouped by the innermost frame inside this repository."""
import collections, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
dev = torch.device("cuda")
cfg = configs.model_config("vidor")
m = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).train()
data = synthetic_batch(cfg, configs.input_channels(cfg), dev, n_pairs=48, seed=0)
for _ in range(2):
    m.zero_grad(set_to_none=True); m(data)["total_loss"].backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.zero_grad(set_to_none=True); m(data)["total_loss"].backward()
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::contiguous")
groups = collections.Counter()
for e in prof.events():
    if e.name in want:
        fr = [s for s in e.stack if "/vrdone_amd/" in s or "/scripts/" in s]
        key = (e.name, fr[0].split("/")[-1][:70] if fr else (e.stack[0][-60:] if e.stack else "?"))
        groups[key] += 1
for (name, where), n in groups.most_common(45):
    print(f"{n:5d}  {name:18s} {where}")
