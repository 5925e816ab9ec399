import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vrdone_amd import ops
from scripts.flash_bench import to_pair
dev = torch.device("cuda", 0)
ops.set_precision("bf16x3")
torch.manual_seed(0)
B, H, hd, T, valid = 2, 4, 128, 288, 256
C = H * hd
def run(q, k, v, flag, mask):
    os.environ["VRD_FLASH_W64"] = flag
    return ops.attention(to_pair(q), to_pair(k), to_pair(v), mask, H, pair=False, q_mask=mask)
mask = (torch.arange(T, device=dev)[None] < valid).expand(B, T).contiguous()
def report(name, got, want):
    d = (got - want).abs()
    d = d * mask[..., None]
    print(name, "max err", d.max().item())
    # per 32-row block and per head/d-tile
    blk = d.view(B, T // 32, 32, H, hd // 32, 32).amax(dim=(0, 2, 5))     # (blocks, H, dtile)
    print(" per row-block (max over heads/dtiles):", [round(x, 4) for x in blk.amax(dim=(1, 2)).tolist()])
    print(" per head:", [round(x, 4) for x in blk.amax(dim=(0, 2)).tolist()], " per d-tile:", [round(x, 4) for x in blk.amax(dim=(0, 1)).tolist()])
q = torch.randn(B, T, C, device=dev); k = torch.randn(B, T, C, device=dev); v = torch.randn(B, T, C, device=dev)
for name, (qq, kk, vv) in {"random": (q, k, v), "K=0": (q, torch.zeros_like(k), v), "V=1": (q, k, torch.ones_like(v)),
                           "q small": (q * 0.01, k, v)}.items():
    a = run(qq, kk, vv, "0", mask); b = run(qq, kk, vv, "1", mask)
    report(name, b, a)
