#!/usr/bin/env python3
"""A/B of the two split-precision flash-attention kernels on the benchmark's SOS shape (one launch = 2048 sequences x 4 heads x
288 rows, 256 valid): per-launch time of each, max difference between them, and against the f32 VALU kernel on a sample.

    python scripts/flash_bench.py [--B 2048] [--T 288] [--valid 256] [--heads 4] [--hd 128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vrdone_amd import ops  # noqa: E402


def to_pair(t):
    """pair rows of the current precision mode: bf16 planes of x, or f16 planes of x * 2^F16_ACT_EXP"""
    from vrdone_amd import _hip
    f16 = ops.get_precision() == "f16x3"
    el = torch.float16 if f16 else torch.bfloat16
    y = t * 2.0 ** _hip.F16_ACT_EXP if f16 else t
    hi = y.to(el)
    lo = (y - hi.float()).to(el)
    C = t.shape[-1]
    raw = torch.stack([hi.reshape(*t.shape[:-1], C // 32, 32), lo.reshape(*t.shape[:-1], C // 32, 32)], dim=-2)
    return ops.Pair(raw.reshape(*t.shape[:-1], 2 * C).contiguous().view(torch.float32), C)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=2048)
    ap.add_argument("--T", type=int, default=288)
    ap.add_argument("--valid", type=int, default=256)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--hd", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--pair", action="store_true", help="pair-row output (what the model asks for)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ops.set_precision(os.environ.get("FB_PREC", "bf16x3"))          # FB_PREC=f16x3: the default mode's element format
    g = torch.Generator(device=dev).manual_seed(1)
    C = a.heads * a.hd
    q, k, v = (to_pair(torch.randn(a.B, a.T, C, device=dev, generator=g)) for _ in range(3))
    mask = (torch.arange(a.T, device=dev)[None] < a.valid).expand(a.B, a.T).contiguous()
    res = {}
    with torch.no_grad():
        for name, flag in (("w32 (2 waves/SIMD)", "0"), ("w64 (1 wave/SIMD)", "1"), ("w32 again", "0"), ("w64 again", "1")):
            os.environ["VRD_FLASH_W64"] = flag
            for _ in range(3):
                out = ops.attention(q, k, v, mask, a.heads, pair=a.pair, q_mask=mask)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                out = ops.attention(q, k, v, mask, a.heads, pair=a.pair, q_mask=mask)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            flops = 4.0 * a.B * a.heads * a.valid * a.valid * a.hd
            print(f"{name:22s} {ms:8.3f} ms / launch   {flops / ms / 1e9:7.1f} TFLOP/s executed", flush=True)
            res[flag] = out
    if a.pair:
        res = {k: v.float() for k, v in res.items()}
    d = (res["0"] - res["1"]).abs().max().item()
    print(f"max |w32 - w64| = {d:.3e}   (outputs are O(1))")
    # an f32 reference on a few sequences
    os.environ.pop("VRD_FLASH_W64")
    n = min(a.B, 8)
    qf, kf, vf = (t.float()[:n] for t in (q, k, v))
    ref = ops.attention(qf.contiguous(), kf.contiguous(), vf.contiguous(), mask[:n].contiguous(), a.heads, algo=1)
    live = mask[:n]
    for flag in ("0", "1"):
        print(f"kernel {flag}: max |x - f32 VALU kernel| on {n} sequences = {(res[flag][:n][live] - ref[live]).abs().max().item():.3e}")


if __name__ == "__main__":
    main()
