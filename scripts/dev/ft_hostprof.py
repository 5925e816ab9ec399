"""cProfile of forward_test on a small video (host-bound regime): where the host time of a call goes."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
model._config_eval(configs.inference_config("vidvrd"))
video = synth.synth_video(16, configs.input_channels(cfg), 20, 90, seed=3, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(video)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        model(video)
    torch.cuda.synchronize()
    pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue()[:6000])
