"""Debug aid: run the pinned training step and check every backward kernel call in situ against torch ops on the same inputs."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case
from golden_cases import train_batch, replay_matching
from oracle import vrd_oracle as O
from vrdone_amd import ops, autograd as A
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.models.blocks import AffineDropPath
ops.set_precision("f32")
mc, ic, keys = load_case("vidvrd")
sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
model = MaskVRD(mc, device="cuda"); model.load_state_dict(sd); model = model.cuda().train()
meta = json.load(open(os.path.join(GOLDEN, "train_step_vidvrd.json")))
lens, _, _, data = train_batch(mc, 2069, device="cuda")
for name, mod in model.named_modules():
    if isinstance(mod, AffineDropPath): mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
replay_matching(model, meta["cases"]["pinned"]["indices"])
bad = []
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
orig_lin = A.Linear.backward
def lin_bwd(ctx, dy):
    out = orig_lin(ctx, dy)
    x, w = ctx.saved_tensors
    m = ctx.row_mask
    g = dy.reshape(-1, dy.shape[-1]).double()
    if m is not None: g = g * m.reshape(-1, 1).double()
    N, Cin, k = w.shape
    if k == 1 and out[1] is not None:
        want = g.T @ x.reshape(-1, Cin).double()
        e = rel(out[1][:, :, 0], want)
        if e > 1e-5: bad.append(("wgrad", tuple(w.shape), tuple(x.shape), e))
    if out[2] is not None:
        e = rel(out[2], g.sum(0))
        if e > 1e-5: bad.append(("bias", tuple(w.shape), e))
    if k == 1 and out[0] is not None:
        want = (g @ w[:, :, 0].double()).reshape(out[0].shape)
        e = rel(out[0], want)
        if e > 1e-5: bad.append(("dgrad", tuple(w.shape), e))
    return out
A.Linear.backward = staticmethod(lin_bwd)
orig_sr = A.ScaleResidual.backward
def sr_bwd(ctx, dy):
    out = orig_sr(ctx, dy)
    v, scale = ctx.saved_tensors
    f = torch.ones(v.shape[:-1], device=v.device, dtype=torch.float64).reshape(-1)
    if ctx.row_scale is not None: f = f * ctx.row_scale.double()
    if ctx.row_mask is not None: f = f * ctx.row_mask.reshape(-1).double()
    g = dy.reshape(-1, dy.shape[-1]).double()
    if out[0] is not None:
        want = g * f[:, None] * (scale.double().reshape(1, -1) if ctx.has_scale else 1.0)
        e = rel(out[0].reshape(want.shape), want)
        if e > 1e-5: bad.append(("sr_dv", tuple(v.shape), e))
    if out[1] is not None:
        want = (g * f[:, None] * v.reshape(g.shape).double()).sum(0)
        e = rel(out[1].reshape(-1), want)
        if e > 1e-5: bad.append(("sr_dscale", tuple(v.shape), e))
    return out
A.ScaleResidual.backward = staticmethod(sr_bwd)
orig_mp = A.MaxPoolMask.backward
def mp_bwd(ctx, dy, dm):
    out = orig_mp(ctx, dy, dm)
    (x,) = ctx.saved_tensors
    xr = x.detach().double().transpose(1, 2).requires_grad_(True)
    with torch.enable_grad():
        yr = torch.nn.functional.max_pool1d(xr, 3, 2, 1) * ctx.mask_in[:, None, ::2].double()
        yr.backward(dy.double().transpose(1, 2))
    want = xr.grad.transpose(1, 2)
    per = [(float((out[0][b].double() - want[b]).norm() / (want[b].norm() + 1e-30)), b) for b in range(x.shape[0])]
    per.sort(reverse=True)
    bad.append(("maxpool", tuple(x.shape), per[:3]))
    if per[0][0] > 1e-5:
        b = per[0][1]
        d = (out[0][b].double() - want[b]).abs()
        idx = torch.nonzero(d > 1e-6 * float(want[b].abs().max()))
        print("maxpool mismatch sample", b, "n", idx.shape[0], "first (t, c):", idx[:6].tolist())
        for t, c in idx[:4].tolist():
            print("   x window", [float(x[b, tt, c]) if 0 <= tt < x.shape[1] else None for tt in range(t - 2, t + 3)], "mask", [bool(ctx.mask_in[b, tt]) if 0 <= tt < x.shape[1] else None for tt in range(t - 2, t + 3)],
                  "got", float(out[0][b, t, c]), "want", float(want[b, t, c]))
    return out
A.MaxPoolMask.backward = staticmethod(mp_bwd)
loss = model(data)
loss["total_loss"].backward()
torch.cuda.synchronize()
print("in-situ mismatches:", len(bad))
for b in bad[:40]: print(b)
