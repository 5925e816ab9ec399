"""Dev: one eager vidor.yaml training step (48 pairs x 512 frames, forward + backward): wall time against the device's busy time
and the number of device events (torch.profiler), to see how much of the step is kernels and how much is the space between them."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
cfg = configs.model_config("vidor")
m = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).train()
data = synthetic_batch(cfg, configs.input_channels(cfg), dev, n_pairs=48, seed=0)
def step():
    m.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m(data)["total_loss"].backward()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)
for _ in range(3): step()
ts = [step() for _ in range(5)]
print("host done after / step done after (ms):", [(round(a, 1), round(b, 1)) for a, b in ts])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as tp:
    step()
ev = [e for e in tp.events() if e.device_type == torch.autograd.DeviceType.CUDA]
busy = sum(e.device_time for e in ev) / 1e3
ev.sort(key=lambda e: e.time_range.start)
span = (ev[-1].time_range.end - ev[0].time_range.start) / 1e3
gaps = sorted(((ev[i].time_range.start - ev[i - 1].time_range.end), ev[i - 1].name[:40], ev[i].name[:40]) for i in range(1, len(ev)))
print(f"device events {len(ev)}, busy {busy:.1f} ms, first-to-last {span:.1f} ms")
print("largest gaps (us):", [(round(g), a, b) for g, a, b in gaps[-8:]])
import collections
h = collections.Counter()
for g, _, _ in gaps:
    h["<3us" if g < 3 else "<6us" if g < 6 else "<15us" if g < 15 else "<100us" if g < 100 else ">=100us"] += max(g, 0) / 1e3
print("gap time by size (ms):", {k: round(v, 2) for k, v in h.items()})
