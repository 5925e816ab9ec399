import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from vrdone_amd import ops
from oracle import vrd_oracle as O
ops.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")
g = torch.Generator().manual_seed(0)
B, T, Cin, D = 48, 96, 1024, 512
lens = torch.randint(2, T + 1, (B,), generator=g)
m = (torch.arange(T)[None] < lens[:, None])
x = torch.randn(B, Cin, T, generator=g) * m[:, None]
w0 = torch.randn(D, Cin, 3, generator=g) / (3 * Cin) ** 0.5
w1 = torch.randn(D, D, 3, generator=g) / (3 * D) ** 0.5
g0, b0 = 1 + 0.1 * torch.randn(1, D, 1, generator=g), 0.1 * torch.randn(1, D, 1, generator=g)
R = torch.randn(B, D, T, generator=g)
def ref(dt):
    xs = [t.detach().clone().to(dt).requires_grad_(True) for t in (w0, g0, b0, w1)]
    h, _ = O.masked_conv1d(x.to(dt), m[:, None], xs[0])
    h = torch.relu(O.channel_ln(h, xs[1], xs[2]))
    y, _ = O.masked_conv1d(h, m[:, None], xs[3])
    (y * R.to(dt)).sum().backward()
    return [t.grad for t in xs]
r64, r32 = ref(torch.float64), ref(torch.float32)
dev = "cuda"
xs = [t.detach().clone().to(dev).requires_grad_(True) for t in (w0, g0, b0, w1)]
xc = x.transpose(1, 2).contiguous().to(dev)
md = m.to(dev)
with torch.enable_grad():
    h = ops.conv_gemm(xc, xs[0], None, row_mask=md)
    h = ops.layernorm(h, xs[1], xs[2], relu=True)
    y = ops.conv_gemm(h, xs[3], None, row_mask=md)
    (y * R.transpose(1, 2).contiguous().to(dev)).sum().backward()
print([t.grad is None for t in xs], [t is None for t in r64], y.requires_grad, h.requires_grad)
for name, a, b64, b32 in zip(("dW0", "dgamma0", "dbeta0", "dW1"), xs, r64, r32):
    e = float((a.grad.double().cpu() - b64).norm() / b64.norm())
    e32 = float((b32.double() - b64).norm() / b64.norm())
    print(f"{name:8s} hip vs f64 {e:.3e}   torch-f32 vs f64 {e32:.3e}")
