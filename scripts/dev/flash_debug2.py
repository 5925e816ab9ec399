import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vrdone_amd import ops
from scripts.flash_bench import to_pair
dev = torch.device("cuda", 0)
ops.set_precision("bf16x3")
torch.manual_seed(0)
B, H, hd = 1, 1, 128
C = H * hd
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
for T, valid in ((256, 256), (256, 32), (256, 64), (256, 96)):
    mask = (torch.arange(T, device=dev)[None] < valid).expand(B, T).contiguous()
    qm = torch.ones(B, T, dtype=torch.bool, device=dev)
    q = torch.randn(B, T, C, device=dev); k = torch.randn(B, T, C, device=dev); v = torch.ones(B, T, C, device=dev)
    os.environ["VRD_FLASH_W64"] = "1"
    b = ops.attention(to_pair(q), to_pair(k), to_pair(v), mask, H, pair=False, q_mask=qm)
    print(f"T {T} valid keys {valid}: out[:, :, 0] rows 0..7:", [round(x, 4) for x in b[0, :8, 0].tolist()], " rows 32..35:", [round(x, 4) for x in b[0, 32:36, 0].tolist()])
    print("   min/max over all:", b.min().item(), b.max().item(), " col variation:", (b.amax(-1) - b.amin(-1)).max().item())
