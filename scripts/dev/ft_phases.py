"""forward_test on bench.py's synthetic video: wall time of the whole call, of pair_candidates (network + per-pair post-processing)
and per-family kernel time, row-space form on / off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth, ops, _hip
from vrdone_amd.models.maskvrd import MaskVRD

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
model._config_eval(configs.inference_config("vidvrd"))
video = synth.synth_video(46, configs.input_channels(cfg), 200, 256, seed=7, device=dev)
lens = [int(f.shape[1]) for f in video["so_features_list"]]
import collections
print("pairs", len(lens), "length histogram (by 32):", sorted(collections.Counter((l + 31) // 32 * 32 for l in lens).items()))
orig = MaskVRD.pair_candidates
spent = []
def timed(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *a, **k)
    torch.cuda.synchronize(); spent.append(time.perf_counter() - t0)
    return r
MaskVRD.pair_candidates = timed
with torch.no_grad():
    for rs in (True, False, True):
        model.row_space = rs
        order, t_pad = model.eval_plan(lens)
        print("row space", rs, "buckets", sorted(collections.Counter(t_pad).items()))
        for it in range(4):
            spent.clear()
            _hip.prof_enable(it == 3); _hip.prof_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = model(video)
            torch.cuda.synchronize(); wall = time.perf_counter() - t0
            _hip.prof_enable(False)
        prof = {k: round(v["ms"], 2) for k, v in _hip.prof_read().items() if v["ms"] > 0.3}
        print(f"   whole call {1e3 * wall:.1f} ms, pair_candidates {1e3 * sum(spent):.1f} ms; kernels {round(sum(prof.values()), 1)} ms: {prof}")
