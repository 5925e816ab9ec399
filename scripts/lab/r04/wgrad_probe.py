"""Lab: vrd_gemm_wgrad_x3 at the vidor training step's sizes (24,576 rows and its strided levels), timed with events: partial tiles
added with atomics against partial tiles stored + summed by a second launch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from vrdone_amd import _hip
lib = _hip.lib
dev = torch.device("cuda")
torch.manual_seed(0)
def run(M, N, Cin, taps=1, T=512, reps=20):
    G = torch.randn(M, N, device=dev); X = torch.randn(M, Cin, device=dev)
    dW = torch.zeros(N, Cin * taps, device=dev); db = torch.zeros(N, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    part = torch.empty(24 << 20, device=dev) if USE_SCRATCH else None
    pp, pn = (part.data_ptr(), part.numel()) if USE_SCRATCH else (None, 0)
    for _ in range(3):
        _hip.check(lib.vrd_gemm_wgrad_x3(G.data_ptr(), N, X.data_ptr(), Cin, None, M, N, Cin, taps, T, dW.data_ptr(), db.data_ptr(), pp, pn, s), "wgrad")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _hip.check(lib.vrd_gemm_wgrad_x3(G.data_ptr(), N, X.data_ptr(), Cin, None, M, N, Cin, taps, T, dW.data_ptr(), db.data_ptr(), pp, pn, s), "wgrad")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us, 2.0 * M * N * Cin * taps / us / 1e6
for mode in ("atomics", "partials"):
    USE_SCRATCH = mode == "partials"
    per_cu = mode + " VRD_WGRAD_BIG=" + os.environ.get("VRD_WGRAD_BIG", "default")
    for (M, N, C) in [(24576, 512, 512), (24576, 2048, 512), (24576, 512, 2048), (12288, 512, 512), (12288, 2048, 512), (6144, 512, 512), (6144, 2048, 512), (3072, 512, 512), (3072, 2048, 512)]:
        us, tf = run(M, N, C)
        print(f"{per_cu}: M {M} N {N} K {C}: {us:8.1f} us  {tf:6.1f} TFLOP/s", flush=True)
