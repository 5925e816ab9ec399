"""Host enqueue time vs GPU time of one _mask_vrd step at a given pair count (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
x, m = synth.synth_pairs(pairs, configs.input_channels(cfg), 288, [256] * pairs, seed=1, device=dev)
for _ in range(3):
    model._mask_vrd(x, m, with_aux=False)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model._mask_vrd(x, m, with_aux=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    wall.append(t2 - t0)
print(f"pairs {pairs}: host enqueue {1e3 * sorted(host)[4]:.2f} ms, wall {1e3 * sorted(wall)[4]:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    model._mask_vrd(x, m, with_aux=False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
