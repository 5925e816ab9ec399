import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vrdone_amd import ops
from scripts.flash_bench import to_pair
dev = torch.device("cuda", 0)
ops.set_precision("bf16x3")
B, H, hd, T, valid = 2048, 4, 128, 288, 256
C = H * hd
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (to_pair(torch.randn(B, T, C, device=dev, generator=g)) for _ in range(3))
mask = (torch.arange(T, device=dev)[None] < valid).expand(B, T).contiguous()
os.environ["VRD_FLASH_W64"] = "1"
for _ in range(3):
    out = ops.attention(q, k, v, mask, H, pair=False, q_mask=mask)
torch.cuda.synchronize()
st = out[:, 0].reshape(B, H, hd)[:, :, :64].reshape(-1, 64)          # (B*H, 64) stamps
med = st.median(dim=0).values.tolist()
names = ["prologue done", "first S done"]
for it in range(8):
    names += [f"it{it} wait", f"it{it} barrier", f"it{it} S-phase", f"it{it} rescale", f"it{it} O-phase"]
names += ["epi barrier", "outputs in LDS", "stores issued"]
prev = 0
for i, n in enumerate(names):
    print(f"{i:2d} {n:16s} {med[i]:9.0f}  (+{med[i]-prev:7.0f})")
    prev = med[i]
