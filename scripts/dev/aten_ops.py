"""Debug aid: which ATen GPU ops a training step still launches, by op and by the Python frame that issued them."""
import collections, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch, param_groups
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
from torch.profiler import profile, ProfilerActivity
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
opt = torch.optim.AdamW(param_groups(model, 0.05), lr=1e-4)
def step(with_opt):
    loss = model(data)
    opt.zero_grad(set_to_none=True)
    loss["total_loss"].backward()
    if with_opt:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
for _ in range(2): step(True)
torch.cuda.synchronize()
for with_opt in (False, True):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step(with_opt)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    print(f"== with optimizer: {with_opt}: {len(evs)} GPU kernels/copies")
    by = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.kernels:
            frame = next((s for s in (e.stack or []) if "vrdone_amd" in s or "train_step" in s or "aten_ops" in s), (e.stack or ["?"])[0] if e.stack else "?")
            by[(e.name, frame.split("/")[-1][:70])] += len(e.kernels)
    for (name, frame), n in by.most_common(40):
        print(f"{n:5d}  {name:28s} {frame}")
