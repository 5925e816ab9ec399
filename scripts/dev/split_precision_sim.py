#!/usr/bin/env python3
"""CPU experiment (no GPU, no reference import): how far is each GEMM arithmetic from a float64 run of the oracle?

The oracle's dense convolutions and global-attention products are replaced by an emulation of the split-precision MFMA
arithmetic (operands split into two 16-bit planes, N cross products, f32 accumulation); everything else stays f32.
Prints max |out - out64| for: plain f32, bf16 x 3 products, fp16 (scaled) x 3 and x 4 products.

    python scripts/dev/split_precision_sim.py [vidvrd|vidor_x] [B] [T]
"""
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import vrd_oracle as O   # noqa: E402

torch.set_grad_enabled(False)


def split(x, kind, scale_exp):
    """x f32 -> (hi, lo) as f32 tensors holding 16-bit values, of x * 2^scale_exp."""
    dt = torch.float16 if kind == "f16" else torch.bfloat16
    xs = x * (2.0 ** scale_exp)
    hi = xs.to(dt)
    lo = (xs - hi.float()).to(dt)
    return hi.float(), lo.float()


class Emu:
    def __init__(self, kind, n_prod, act_exp=0, dyn_w=True):
        self.kind, self.n_prod, self.act_exp, self.dyn_w = kind, n_prod, act_exp, dyn_w

    def w_exp(self, w):
        if self.kind != "f16" or not self.dyn_w:
            return 0
        m = float(w.abs().max())
        return 14 - math.frexp(m)[1] if m > 0 else 0      # max |w| * 2^e in [2^13, 2^14)

    def mm(self, a, w, ea, ew):
        """a (M,K) f32, w (K,N) f32 -> a @ w through the split products, f32 accumulate per product."""
        ah, al = split(a, self.kind, ea)
        wh, wl = split(w, self.kind, ew)
        acc = ah @ wh + (ah @ wl + al @ wh)
        if self.n_prod == 4:
            acc = acc + al @ wl
        return acc * (2.0 ** -(ea + ew))

    def conv1d(self, x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 or x.dtype != torch.float32:
            return F.conv1d(x, weight, bias, stride, padding, dilation, groups)
        B, C, T = x.shape
        Co, Ci, k = weight.shape
        assert stride == 1 and dilation == 1
        xp = F.pad(x, (padding, padding))
        cols = xp.unfold(2, k, 1)                                   # (B, C, T', k)
        Tn = cols.shape[2]
        a = cols.permute(0, 2, 1, 3).reshape(B * Tn, C * k)
        w = weight.reshape(Co, Ci * k).t()
        y = self.mm(a, w, self.act_exp, self.w_exp(weight))
        y = y.reshape(B, Tn, Co).permute(0, 2, 1)
        if bias is not None:
            y = y + bias[None, :, None]
        return y

    def full_attention(self, q, k, v, kv_mask, n_head):
        q, k, v = O._split_heads(q, n_head), O._split_heads(k, n_head), O._split_heads(v, n_head)
        hd = q.shape[-1]
        if q.dtype != torch.float32:
            raise RuntimeError
        qs = q * (1.0 / math.sqrt(hd))
        qh, ql = split(qs, self.kind, self.act_exp)
        kh, kl = split(k, self.kind, self.act_exp)
        kt_h, kt_l = kh.transpose(-2, -1), kl.transpose(-2, -1)
        att = qh @ kt_h + (qh @ kt_l + ql @ kt_h)
        if self.n_prod == 4:
            att = att + ql @ kt_l
        att = att * (2.0 ** -(2 * self.act_exp))
        att = att.masked_fill(~kv_mask[:, :, None, :], float("-inf"))
        # un-normalised probabilities in (0, 1] are what the kernel splits; the sum divides at the end
        mx = att.max(dim=-1, keepdim=True).values
        p = torch.exp(att - mx)
        denom = p.sum(-1, keepdim=True)
        pe = 10 if self.kind == "f16" else 0
        ph, pl = split(p, self.kind, pe)
        vv = v * kv_mask[:, :, :, None].to(v.dtype)
        vh, vl = split(vv, self.kind, self.act_exp)
        o = ph @ vh + (ph @ vl + pl @ vh)
        if self.n_prod == 4:
            o = o + pl @ vl
        o = o * (2.0 ** -(pe + self.act_exp)) / denom
        return O._merge_heads(o)


class FProxy:
    def __init__(self, emu):
        self.emu = emu

    def __getattr__(self, name):
        if name == "conv1d":
            return self.emu.conv1d
        return getattr(F, name)


def run(sd, cfg, x, m, emu=None):
    if emu is None:
        return O.mask_vrd(sd, cfg, x, m)
    saveF, saveA = O.F, O.full_attention
    O.F, O.full_attention = FProxy(emu), emu.full_attention
    try:
        return O.mask_vrd(sd, cfg, x, m)
    finally:
        O.F, O.full_attention = saveF, saveA


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "vidvrd"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    in_scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    meta = json.load(open(os.path.join(REPO, "tests", "golden", f"state_keys_{name}.json")))
    cfg = meta["model_config"]
    sd = O.synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], eos_coef=cfg["loss_coeff_dict"]["eos_coef"])
    V, E, S = cfg["visual_dim"], cfg["bbox_entity_dim"], cfg["bbox_so_dim"]
    Cc = cfg["clip_dim"] if cfg.get("with_clip_feature", False) else 0
    c_in = 2 * V + 2 * Cc + S + 2 * E
    lens = [T, T - 1, max(2, T // 2 - 7), 2][:B] + [T] * max(0, B - 4)
    x, m = O.synth_pairs(B, c_in, T, lens, seed=99)
    x = x * in_scale
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    ref64 = run(sd64, cfg, x.double(), m)
    outs = {"f32": run(sd, cfg, x, m)}
    for tag, emu in [("bf16x3", Emu("bf16", 3)), ("f16x3 s0", Emu("f16", 3, 0)), ("f16x3 s4", Emu("f16", 3, 4)),
                     ("f16x4 s4", Emu("f16", 4, 4)), ("f16x3 s4 w-unscaled", Emu("f16", 3, 4, dyn_w=False))]:
        outs[tag] = run(sd, cfg, x, m, emu)
    for tag, o in outs.items():
        dl = (o["pred_logits"].double() - ref64["pred_logits"]).abs()
        dm = (o["pred_masks"].double() - ref64["pred_masks"]).abs()
        print(f"{tag:22s} logits max {dl.max():.3e} rms {dl.pow(2).mean().sqrt():.3e}   masks max {dm.max():.3e} "
              f"rms {dm.pow(2).mean().sqrt():.3e}")




def goldens():
    """python scripts/dev/split_precision_sim.py --goldens : the same comparison on the inputs of the committed goldens, as ratios
    to the reference's own |ref32 - ref64| (tests/golden/mask_vrd_f64.npz)."""
    import numpy as np
    G = os.path.join(REPO, "tests", "golden")
    f64 = np.load(os.path.join(G, "mask_vrd_f64.npz"))
    cases = {"vidvrd": [(4, 96, [96, 95, 41, 2]), (3, 144, [144, 97, 130]), (2, 288, [288, 201])],
             "vidor_x": [(2, 512, [512, 333])], "vidor_local": [(2, 512, [512, 77])], "vidor": [(2, 512, [512, 301])]}
    for name, shapes in cases.items():
        meta = json.load(open(os.path.join(G, f"state_keys_{name}.json")))
        cfg = meta["model_config"]
        sd = O.synth_state_dict([(k, tuple(s)) for k, s in meta["keys"]], eos_coef=cfg["loss_coeff_dict"]["eos_coef"])
        V, E, S = cfg["visual_dim"], cfg["bbox_entity_dim"], cfg["bbox_so_dim"]
        Cc = cfg["clip_dim"] if cfg.get("with_clip_feature", False) else 0
        g32 = np.load(os.path.join(G, f"mask_vrd_{name}.npz"))
        for B, T, lens in shapes:
            x, m = O.synth_pairs(B, 2 * V + 2 * Cc + S + 2 * E, T, lens, seed=1234 + T)
            row = [f"{name} T{T}"]
            for key in ("pred_logits", "pred_masks"):
                r64 = f64[f"{name}_T{T}_{key}"]
                e32 = np.abs(g32[f"T{T}_{key}"] - r64)
                row.append(f"{key[5:]} e32 max {e32.max():.2e} rms {np.sqrt((e32 ** 2).mean()):.2e}")
            print(" | ".join(row), flush=True)
            for tag, emu in [("oracle f32", None), ("f16x3 s4", Emu("f16", 3, 4)), ("f16x4 s4", Emu("f16", 4, 4))]:
                o = run(sd, cfg, x, m, emu)
                row = [f"    {tag:12s}"]
                for key in ("pred_logits", "pred_masks"):
                    r64 = f64[f"{name}_T{T}_{key}"]
                    e32 = np.abs(g32[f"T{T}_{key}"] - r64)
                    e = np.abs(o[key].double().numpy() - r64)
                    row.append(f"{key[5:]} max x{e.max() / e32.max():.2f} rms x{np.sqrt((e ** 2).mean() / (e32 ** 2).mean()):.2f}")
                print(" | ".join(row), flush=True)


if __name__ == "__main__":
    goldens() if "--goldens" in sys.argv else main()
