import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from vrdone_amd import ops
torch.set_grad_enabled(False)
B, T, D = 4096, 288, 512
x = torch.randn(B, T, D, device="cuda")
g1, b1 = torch.randn(1, D, 1, device="cuda"), torch.randn(1, D, 1, device="cuda")
xp = ops.layernorm(x, g1, b1, pair=True)
w = torch.randn(D, D, 1, device="cuda") / D ** 0.5
bias = torch.randn(D, device="cuda")
out = torch.empty(B, T, D, device="cuda")
def run():
    ops.conv_gemm(xp, w, bias, out=out)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
torch.cuda.synchronize()
print(os.environ.get("VRD_BIG_DIRECT_EPI"), "ms per gemm", 1e3 * (time.perf_counter() - t0) / 20, float(out.double().sum()))
