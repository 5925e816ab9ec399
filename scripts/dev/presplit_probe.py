import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts"); os.chdir("/root/repo")
import torch
from train_step import synthetic_batch
from vrdone_amd import configs, synth, ops, _hip
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
calls = {"one": 0, "batch": 0}
lib = _hip.lib
orig1, origb = lib.vrd_split_weight, lib.vrd_split_weights
class W:
    def __init__(self, f, k): self.f, self.k = f, k
    def __call__(self, *a): calls[self.k] += 1; return self.f(*a)
ops.lib.vrd_split_weight = W(orig1, "one"); ops.lib.vrd_split_weights = W(origb, "batch")
for step in range(3):
    calls["one"] = calls["batch"] = 0
    model.zero_grad(set_to_none=True)
    loss = model(data)["total_loss"]
    nf = calls["one"]
    loss.backward()
    print("step", step, "batch launches", calls["batch"], "single launches forward", nf, "backward", calls["one"] - nf)
    with torch.no_grad():
        for p in model.parameters(): p.mul_(1.0001)
print("--- training graphs: launches issued while recording (warm-up iterations + the two captures)")
model.enable_training_graphs(True)
for step in range(2):
    calls["one"] = calls["batch"] = 0
    model.zero_grad(set_to_none=True)
    loss = model(data)["total_loss"]
    loss.backward()
    print("graph step", step, "batch launches", calls["batch"], "single launches", calls["one"])
    with torch.no_grad():
        for p in model.parameters(): p.mul_(1.0001)
