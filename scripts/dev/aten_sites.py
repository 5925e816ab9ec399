"""Debug aid: which source lines of vrdone_amd issue the ATen ops of one training step (forward + backward), by op.
A TorchDispatchMode sees every aten call (autograd's backward included) and records the innermost vrdone_amd frame."""
import collections, os, sys, traceback, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch
from vrdone_amd import configs, synth
from vrdone_amd.models.maskvrd import MaskVRD
from torch.utils._python_dispatch import TorchDispatchMode

cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
for _ in range(2):
    model.zero_grad(set_to_none=True)
    model(data)["total_loss"].backward()
torch.cuda.synchronize()
WANT = ("copy_", "add_", "cat", "fill_", "add", "mul", "uniform_", "floor", "div", "zero_", "clone", "contiguous", "_to_copy", "index", "masked_fill_", "where", "sum", "stack")
sites = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WANT and any(isinstance(a, torch.Tensor) and a.is_cuda for a in args):
            fr = [f for f in traceback.extract_stack() if "vrdone_amd" in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "(autograd engine / outside)"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


model.zero_grad(set_to_none=True)
with Mode():
    model(data)["total_loss"].backward()
torch.cuda.synchronize()
for (name, where), n in sites.most_common(45):
    print(f"{n:5d}  {name:14s} {where}")
