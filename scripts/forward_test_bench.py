#!/usr/bin/env python3
"""Wall time of the full eval call MaskVRD.forward_test (batching + hot path + device post-processing) on a
synthetic video: N tracklets, all ordered pairs (N=46 -> 2070 pairs), features already on the device."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vrdone_amd import configs, synth  # noqa: E402
from vrdone_amd.models.maskvrd import MaskVRD  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracklets", type=int, default=46)
ap.add_argument("--min-len", type=int, default=200)
ap.add_argument("--max-len", type=int, default=256)
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()
torch.set_grad_enabled(False)
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().eval()
model._config_eval(configs.inference_config("vidvrd"))
dev = synth.synth_video(args.tracklets, configs.input_channels(cfg), args.min_len, args.max_len, seed=7, device="cuda")
P = len(dev["sids"])
for it in range(args.iters + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = model(dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if it:
        print(f"forward_test: {P} pairs, {len(res['triplets'])} triplets, {dt * 1e3:.1f} ms = {P / dt:.0f} pairs/s", flush=True)

from vrdone_amd import _hip  # noqa: E402
_hip.prof_enable(True)
_hip.prof_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
res = model(dev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
prof = _hip.prof_read()
_hip.prof_enable(False)
print(f"profiled call: {dt * 1e3:.1f} ms wall; kernel ms by family:",
      {k: round(v["ms"], 2) for k, v in prof.items() if v["launches"]}, "sum", round(sum(v["ms"] for v in prof.values()), 1))

if os.environ.get("FT_CPROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    res = model(dev)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
