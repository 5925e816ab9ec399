#!/usr/bin/env python3
"""Lab (round 4): can an HBM-bound row kernel run beside the MFMA-bound 256 x 256 GEMM?  Two kernels the step alternates
between, each alone, then together on two plain streams, then on two streams with disjoint CU masks
(hipExtStreamCreateWithCUMask).  Prints milliseconds per pair of launches."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from vrdone_amd import ops  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda"
B, T, D = 2048, 288, 512
x = torch.randn(2 * B, T, D, device=dev)
w = torch.randn(D, D, 1, device=dev) / D ** 0.5
bias = torch.randn(D, device=dev)
g1, b1 = torch.randn(1, D, 1, device=dev), torch.randn(1, D, 1, device=dev)
xp = ops.layernorm(x, g1, b1, pair=True)
y = torch.randn(B, T, D, device=dev)
out_g = torch.empty(2 * B, T, D, device=dev)
out_l = torch.empty(B, T, D, device=dev)


def gemm():
    ops.conv_gemm(xp, w, bias, out=out_g)


def ln():
    for _ in range(4):
        ops.layernorm(y, g1, b1, out=out_l)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def masked_stream(lo, hi):
    hip = C.CDLL("libamdhip64.so")
    mask = [0] * 8
    for cu in range(lo, hi):
        mask[cu // 32] |= 1 << (cu % 32)
    arr = (C.c_uint32 * 8)(*mask)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def both(sa, sb):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        gemm()
    with torch.cuda.stream(sb):
        ln()
    cur.wait_stream(sa)
    cur.wait_stream(sb)


print(f"gemm alone {timed(gemm):.3f} ms   4 x layernorm alone {timed(ln):.3f} ms")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two plain streams: {timed(lambda: both(s1, s2)):.3f} ms")
for n_row in (16, 32, 48, 64):
    # CU numbering of the mask: bit i = CU i in the driver's flat order (XCD-interleaved); take the row CUs evenly: every k-th
    ga, la = masked_stream(0, 256 - n_row), masked_stream(256 - n_row, 256)
    with torch.cuda.stream(ga):
        tg = timed(gemm)
    with torch.cuda.stream(la):
        tl = timed(ln)
    print(f"masks {256 - n_row} / {n_row} CUs: gemm alone {tg:.3f}, layernorm alone {tl:.3f}, together {timed(lambda: both(ga, la)):.3f} ms")
