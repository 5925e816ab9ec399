#!/usr/bin/env python3
"""Micro-benchmark of vrd_gemm on the shapes of the path (per 256-pair chunk at T_pad 288).
    python scripts/gemm_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vrdone_amd import ops  # noqa: E402

SHAPES = [  # (M, N, Cin, taps, label)
    (147456, 512, 512, 1, "qkv/proj 2B"),
    (73728, 512, 512, 1, "qkv/proj B"),
    (147456, 2048, 512, 1, "mlp up"),
    (147456, 512, 2048, 1, "mlp down"),
    (147456, 512, 1024, 3, "visual_embd0 k3"),
    (147456, 512, 512, 3, "visual_embd1 k3"),
    (147456, 512, 1024, 1, "fuse l0"),
    (36864, 512, 512, 1, "branch T/2"),
    (73728, 256, 512, 1, "fpn lateral"),
    (2304, 256, 256, 1, "predictor"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--pair", action="store_true", help="feed pair-row activations (bf16x3 mode)")
    args = ap.parse_args()
    dev = "cuda"
    for i, (M, N, Cin, taps, label) in enumerate(SHAPES):
        if args.only >= 0 and i != args.only:
            continue
        T = 288
        x = torch.randn(M // T, T, Cin, device=dev)
        if args.pair and ops.pair_mode() and (Cin * taps) % 32 == 0 and Cin % 32 == 0:
            x = ops.Pair(x, Cin)      # timing only: the bit pattern is arbitrary bf16 data
        w = torch.randn(N, Cin, taps, device=dev) / (Cin * taps) ** 0.5
        b = torch.randn(N, device=dev)
        mask = torch.ones(M // T, T, dtype=torch.bool, device=dev)
        res = torch.randn(M // T, T, N, device=dev)
        out = torch.empty(M // T, T, N, device=dev)
        for epi, kw in (("plain", {}), ("gelu", dict(act=ops.ACT_GELU)),
                        ("mask+scale+res", dict(row_mask=mask, scale=b, res=res, res_masked=True))):
            for _ in range(3):
                ops.conv_gemm(x, w, b, out=out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                ops.conv_gemm(x, w, b, out=out, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            tf = 2.0 * M * N * Cin * taps / ms / 1e9
            print(f"{label:18s} M={M:7d} N={N:5d} K={Cin * taps:5d} {epi:15s} {ms:8.3f} ms  {tf:7.1f} TF/s  {tf / 157.3 * 100:5.1f}%", flush=True)


if __name__ == "__main__":
    main()
