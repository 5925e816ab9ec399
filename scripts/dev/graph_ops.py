"""Dev aid: every library op captured into a HIP graph and replayed on new input contents, against the eager result."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from vrdone_amd import ops
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s): return torch.randn(*s, device=dev, generator=g)
def check(name, fn, *inputs):
    """fn(*inputs) -> tensor or list; captured once, inputs overwritten in place, replayed."""
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(*inputs)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn(*inputs)
    for t in inputs:
        if t.dtype == torch.float32: t.copy_(rnd(*t.shape))
    want = fn(*inputs)
    gr.replay(); torch.cuda.synchronize()
    flat = lambda o: [o] if torch.is_tensor(o) else [p.float() if hasattr(p, "float") else p for p in o]
    d = [float((a.float() - b.float()).abs().max()) for a, b in zip(flat(out), flat(want))]
    print(f"{name:28s} max diff {d}", flush=True)

B, T, C = 24, 96, 512
mask = (torch.arange(T, device=dev)[None] < torch.randint(8, T + 1, (B, 1), device=dev)).contiguous()
for prec in ("bf16x3", "f32"):
    ops.set_precision(prec)
    with torch.no_grad():
        w1, b1 = rnd(512, 512, 1) * 0.04, rnd(512) * 0.02
        w3 = rnd(512, 512, 3) * 0.03
        wu = rnd(2048, 512, 1) * 0.04
        check(f"{prec} conv_gemm k1", lambda x: ops.conv_gemm(x, w1, b1, row_mask=mask), rnd(B, T, C))
        check(f"{prec} conv_gemm k3", lambda x: ops.conv_gemm(x, w3, b1, row_mask=mask), rnd(B, T, C))
        check(f"{prec} conv_gemm gelu N2048", lambda x: ops.conv_gemm(x, wu, None, act=ops.ACT_GELU), rnd(B, T, C))
        check(f"{prec} conv_gemm small M", lambda x: ops.conv_gemm(x, w1, b1), rnd(B, 9, C))
        gam, bet = rnd(1, C, 1) * 0.1 + 1, rnd(1, C, 1) * 0.1
        check(f"{prec} layernorm", lambda x: ops.layernorm(x, gam, bet), rnd(B, T, C))
        dw = rnd(C, 1, 3) * 0.5
        check(f"{prec} dwconv_ln", lambda x: ops.dwconv_ln(x, [dict(weight=dw, gamma=gam, beta=bet)], mask_out=mask), rnd(B, T, C))
        check(f"{prec} local_attention", lambda q, k, v: ops.local_attention(q, k, v, mask, 4, 3), rnd(B, T, C), rnd(B, T, C), rnd(B, T, C))
        check(f"{prec} attention", lambda q, k, v: ops.attention(q, k, v, mask, 4), rnd(B, T, C), rnd(B, T, C), rnd(B, T, C))
        check(f"{prec} attention 9q", lambda q, k, v: ops.attention(q, k, v, mask, 8), rnd(B, 9, 256), rnd(B, T, 256), rnd(B, T, 256))
