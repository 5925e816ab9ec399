"""Debug aid: one TransformerBlock with pinned stochastic-depth decisions, forward + backward in both precision modes against
float64 autograd of the oracle's restatement with the keep factors applied."""
import os, sys, torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import vrd_oracle as O
from vrdone_amd import ops
from vrdone_amd.models.blocks import TransformerBlock
DEV = "cuda"
def rel(a, b, floor=0.0):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm()) / (float(b.norm()) + floor + 1e-12)
def block64(sd, pre, x, mask, n_head, win, stride, ka, km):
    h = O.channel_ln(x, sd[f"{pre}.ln1.weight"], sd[f"{pre}.ln1.bias"])
    a, m = O.local_mhca(sd, f"{pre}.attn", h, mask, n_head, win, stride)
    mf = m.to(x.dtype)
    skip = x if stride == 1 else F.max_pool1d(x, stride + 1, stride, (stride + 1) // 2)
    y = skip * mf + sd[f"{pre}.drop_path_attn.scale"] * a * ka.view(-1, 1, 1)
    h = O.channel_ln(y, sd[f"{pre}.ln2.weight"], sd[f"{pre}.ln2.bias"])
    h = F.conv1d(h, sd[f"{pre}.mlp.0.weight"], sd[f"{pre}.mlp.0.bias"])
    h = F.conv1d(F.gelu(h), sd[f"{pre}.mlp.3.weight"], sd[f"{pre}.mlp.3.bias"])
    return y + sd[f"{pre}.drop_path_mlp.scale"] * (h * mf) * km.view(-1, 1, 1), m
import json
meta = json.load(open(os.path.join(REPO, "tests", "golden", "train_step_vidvrd.json")))
for stride in (1, 2):
    torch.manual_seed(0)
    C, H, B, T, win = 512, 4, 24, 48, 7
    blk = TransformerBlock(C, H, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=win)
    keys = [(f"blk.{k}", list(v.shape)) for k, v in blk.state_dict().items()]
    sd = O.synth_state_dict(keys)
    blk.load_state_dict({k[4:]: v for k, v in sd.items()})
    blk = blk.to(DEV).train()
    keep_a = torch.tensor(meta["keep"]["backbone.branch.1.drop_path_attn"][:B], dtype=torch.float32)
    keep_m = torch.tensor(meta["keep"]["backbone.branch.1.drop_path_mlp"][:B], dtype=torch.float32)
    blk.drop_path_attn.keep, blk.drop_path_mlp.keep = keep_a, keep_m
    lens = (torch.tensor(meta["lengths"]) + 1) // 2
    m = (torch.arange(T)[None] < lens[:, None])[:, None]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, T, generator=g) * m
    dy = torch.randn(B, C, T // stride, generator=g)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    yr, _ = block64(sd64, "blk", x64, m, H, win, stride, keep_a.double() / 0.9, keep_m.double() / 0.9)
    yr.backward(dy.double())
    for mode in ("f32", "bf16x3"):
        ops.set_precision(mode)
        blk.zero_grad(set_to_none=True)
        xd = x.to(DEV).requires_grad_(True)
        with torch.enable_grad():
            y, _ = blk(xd, m.to(DEV))
        y.backward(dy.to(DEV))
        floor = 1e-3 * max(float(v.grad.norm()) for v in sd64.values())
        worst = max(((rel(p.grad, sd64["blk." + n].grad, floor), n) for n, p in blk.named_parameters()))
        print(f"stride {stride} {mode:7s}: out {rel(y, yr):.1e}  dx {rel(xd.grad, x64.grad):.1e}  worst param {worst[0]:.1e} ({worst[1]})")
        for n, p in blk.named_parameters():
            e = rel(p.grad, sd64["blk." + n].grad, floor)
            if e > 2e-4: print(f"      {n}: {e:.1e}")
