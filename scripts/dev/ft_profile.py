"""Kernel-family time vs wall time of forward_test fed from per-tracklet features (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import _hip, configs, ops, synth
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.proposals import prepare_test_proposal

share = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = configs.model_config("vidvrd")
ic = configs.inference_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
model._config_eval(ic)
model.share_tracklets = bool(share)
raw = synth.synth_raw_video(46, cfg["visual_dim"], 200, 256, seed=7)
prop = prepare_test_proposal(raw, ic["feat_stride"], 0, 2, dev)
for _ in range(2):
    model(prop)
torch.cuda.synchronize()
_hip.prof_enable(True)
_hip.prof_reset()
t0 = time.perf_counter()
model(prop)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
_hip.prof_enable(False)
prof = _hip.prof_read()
tot = sum(v["ms"] for v in prof.values())
print(f"share={share} wall {1e3 * wall:.1f} ms (with event recording), library kernels {tot:.1f} ms, launches {sum(v['launches'] for v in prof.values())}")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    if v["launches"]:
        print(f"  {k:22s} {v['ms']:8.2f} ms  {v['launches']:5d} launches")
# host-only timing of the stages
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
model(prop)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
