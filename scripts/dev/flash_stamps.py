"""Lab aid: where an item of the one-wave-per-SIMD flash kernel spends its cycles.  Needs a library built with -DVRD_ATTN_STAMP
(VRDONE_HIP_LIB=...): workgroup 0 sums, per stamp, the s_memtime distance from the start of each of its items.

    VRDONE_HIP_LIB=$PWD/scripts/lab/libs/libvrdone_stamp.so python scripts/dev/flash_stamps.py [f16]
"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vrdone_amd import ops, _hip
from scripts.flash_bench import to_pair
dev = torch.device("cuda", 0)
ops.set_precision("bf16x3")
B, H, hd, T, valid = int(os.environ.get('FS_B', 2048)), 4, 128, 288, 256
C = H * hd
g = torch.Generator(device=dev).manual_seed(1)
q, k, v = (to_pair(torch.randn(B, T, C, device=dev, generator=g)) for _ in range(3))
mask = (torch.arange(T, device=dev)[None] < valid).expand(B, T).contiguous()
os.environ["VRD_FLASH_W64"] = "1"
lib = _hip.lib
buf = (ctypes.c_ulonglong * 65)()
for pf in (os.environ.get("VRD_FLASH_PREFETCH", "1"),):
    for _ in range(3):
        out = ops.attention(q, k, v, mask, H, pair=True, q_mask=mask)
    torch.cuda.synchronize()
    lib.vrd_lab_attn_stamps(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = ops.attention(q, k, v, mask, H, pair=True, q_mask=mask)
    e1.record()
    torch.cuda.synchronize()
    lib.vrd_lab_attn_stamps(buf, 1)
    n = buf[64]
    names = {0: "item start", 1: "tables built", 2: "tiles requested", 3: "first S done", 40: "epi barrier", 41: "outputs in LDS", 42: "block 0 done",
             43: "block 1 done", 50: "item end"}
    for it in range(9):
        names[4 + 3 * it] = f"it{it} wait"
        names[5 + 3 * it] = f"it{it} barrier"
        names[6 + 3 * it + 3] = names.get(6 + 3 * it + 3, "")
    for it in range(1, 10):
        names[3 + 3 * it] = f"it{it-1} tile done"
    print(f"prefetch={pf}: {e0.elapsed_time(e1) / 10:.3f} ms per launch, {n} items stamped")
    names.update({51: "Q loads issued, accumulators zeroed", 52: "tile 0 requested", 53: "Q in the accumulator half", 54: "tile 0 landed"})
    prev = 0.0
    names.update({55: "block live", 56: "Q requested", 57: "accumulators zeroed", 51: "tile tables read", 52: "Q landed", 53: "Q in the accumulator half, tile 0 requested"})
    order = [0, 1, 55, 56, 57, 51, 52, 53, 2, 54] + list(range(3, 51))
    for i in order:
        if i not in names: continue
        t = buf[i] / max(n, 1)
        if (t == 0 and i) or t > 1e9: continue
        print(f"  {i:2d} {names[i]:16s} {t:9.0f}  (+{t - prev:7.0f})")
        prev = t
