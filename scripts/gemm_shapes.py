#!/usr/bin/env python3
"""Per-shape time of the conv GEMMs inside one _mask_vrd step (event pair around every call; run on the GPU box).
    python scripts/gemm_shapes.py [--pairs 2048]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vrdone_amd import configs, ops, synth  # noqa: E402
from vrdone_amd.models.maskvrd import MaskVRD  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=2048)
args = ap.parse_args()
torch.set_grad_enabled(False)
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().eval()
c_in = configs.input_channels(cfg)
x = torch.randn(args.pairs, c_in, 288, device="cuda")
mask = torch.ones(args.pairs, 1, 288, dtype=torch.bool, device="cuda")
mask[:, :, 256:] = False
model._mask_vrd(x, mask, with_aux=False)
torch.cuda.synchronize()

orig = ops.conv_gemm
rec = collections.OrderedDict()


def timed(xx, weight, bias=None, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(xx, weight, bias, **kw)
    e1.record()
    t = xx.t if isinstance(xx, ops.Pair) else xx
    rows = t.numel() // t.shape[-1]
    N, Cin, k = weight.shape
    key = (rows, N, Cin * k, k, "pair" if isinstance(xx, ops.Pair) else "f32", "pairout" if kw.get("out_pair") else "f32out",
           "gelu" if kw.get("act") == ops.ACT_GELU else ("rowin" if (kw.get("row_mask") is not None or kw.get("res") is not None) else "plain"))
    rec.setdefault(key, []).append((e0, e1))
    return out


ops.conv_gemm = timed
import vrdone_amd.models.blocks as blocks  # noqa: E402
model._mask_vrd(x, mask, with_aux=False)
torch.cuda.synchronize()
tot = 0.0
rows_out = []
for key, evs in rec.items():
    ms = sum(a.elapsed_time(b) for a, b in evs)
    tot += ms
    rows_out.append((ms, key, len(evs)))
rows_out.sort(reverse=True)
print(f"total GEMM time {tot:.1f} ms over {sum(r[2] for r in rows_out)} calls")
for ms, key, n in rows_out:
    M, N, K, k, ain, cout, epi = key
    tf = 2.0 * M * N * K * n / ms / 1e9
    print(f"{ms:8.2f} ms  {n:3d} x  M={M:8d} N={N:5d} K={K:5d} k={k} {ain:4s} {cout:7s} {epi:6s} {tf:7.1f} TF/s")
