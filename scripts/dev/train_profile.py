"""Debug aid: where a training step's wall time goes (kernel time per family vs wall)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import _hip, configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
for it in range(3):
    model.zero_grad(set_to_none=True)
    if it == 2:
        _hip.prof_enable(True); _hip.prof_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = model(data)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss["total_loss"].backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
prof = _hip.prof_read(); _hip.prof_enable(False)
print(f"forward {1e3*(t1-t0):.1f} ms  backward {1e3*(t2-t1):.1f} ms")
tot_ms = sum(v["ms"] for v in prof.values()); tot_n = sum(v["launches"] for v in prof.values())
print(f"kernel time {tot_ms:.1f} ms over {tot_n} launches of the library")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    if v["launches"]: print(f"  {k:20s} {v['ms']:7.2f} ms {v['launches']:5d} launches")
# host-side view of the same step
import cProfile, pstats
model.zero_grad(set_to_none=True)
pr = cProfile.Profile(); pr.enable()
loss = model(data); loss["total_loss"].backward()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
