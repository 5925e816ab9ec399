import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vrdone_amd import ops
from scripts.flash_bench import to_pair
dev = torch.device("cuda", 0)
ops.set_precision("bf16x3")
torch.manual_seed(0)
B, H, hd, T, valid = 1, 1, 128, 256, 32
C = H * hd
mask = (torch.arange(T, device=dev)[None] < valid).expand(B, T).contiguous()
qm = torch.ones(B, T, dtype=torch.bool, device=dev)
q = torch.randn(B, T, C, device=dev); k = torch.randn(B, T, C, device=dev); v = torch.ones(B, T, C, device=dev)
os.environ["VRD_FLASH_W64"] = "1"
b = ops.attention(to_pair(q), to_pair(k), to_pair(v), mask, H, pair=False, q_mask=qm)[0]
s = (q[0] @ k[0, :valid].T) / math.sqrt(hd) * 1.4426950408889634          # (T, valid), log2 domain
m = s.max(dim=1).values
l = torch.exp2(s - m[:, None]).sum(dim=1)
for r in (0, 1, 2, 3, 31, 32, 33, 64, 100, 255):
    print(f"row {r}: kernel l_tot {b[r,0]:.4f} m {b[r,1]:.4f} l_part {b[r,2]:.4f} sum {b[r,3]:.4f} | O(d=8) {b[r,8]:.4f} O(d=40) {b[r,40]:.4f} || ref l {l[r]:.4f} m {m[r]:.4f}")
# per-lane-half partial sums: lh=0 holds keys {0-3,8-11,16-19,24-27}, lh=1 the others
idx0 = torch.tensor([0,1,2,3,8,9,10,11,16,17,18,19,24,25,26,27], device=dev)
p = torch.exp2(s - m[:, None])
print("ref partial lh=0 rows 0..3:", p[:4][:, idx0].sum(1).tolist(), " (kernel l_part in col 2 is lane-half 0's; col 6 is half 1's):", b[:4, 6].tolist())
