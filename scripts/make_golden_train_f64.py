#!/usr/bin/env python3
"""float64 gradients of the REAL reference's training step (build container only), next to the float32 ones of
scripts/make_golden_train.py: tests/golden/train_step_vidvrd_f64.npz.

Why: "our gradient error is of the size of the reference's own float32 error" was prose.  With the reference differentiated
in float64 on the same batch, the same pinned stochastic-depth decisions and the same (replayed) matching, every parameter has
e32 = |ref32 - ref64|, and tests/test_gpu_train.py bounds |ours - ref64| by a multiple of it.  Stored per case (nodrop,
pinned) and parameter, in the layout of train_step_vidvrd.npz (the full gradient up to 2048 elements, else the stride-499
sample): d = float32(ref64 - ref32) -- ref64 = ref32 + d to ~1e-13 relative, at half the bytes of a float64 array.

    python scripts/make_golden_train_f64.py
"""
import os
os.environ.setdefault("PYTORCH_JIT", "0")
import json
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden_train as M                                # noqa: E402
from make_golden import OUT, build, load_cfg                 # noqa: E402
from models import blocks as ref_blocks                      # noqa: E402  (reference)


def main():
    torch.set_grad_enabled(True)
    cfg, mc = load_cfg("vidvrd.yaml")
    meta = json.load(open(os.path.join(OUT, "train_step_vidvrd.json")))
    g32 = np.load(os.path.join(OUT, "train_step_vidvrd.npz"))
    arrs = {}
    orig_drop = ref_blocks.drop_path
    for case in ("nodrop", "pinned"):
        model, _, _ = build(mc)
        model = model.double()
        for p in model.parameters():
            p.requires_grad_(True)
        lens, data = M.batch(mc)
        assert lens == meta["lengths"]
        data = {k: [t.double() if t.is_floating_point() else t for t in v] for k, v in data.items()}
        if case == "nodrop":
            ref_blocks.drop_path = lambda x, drop_prob=0.0, training=False: x
        else:
            ref_blocks.drop_path = orig_drop
            for name, mod in model.named_modules():
                if isinstance(mod, ref_blocks.AffineDropPath) and mod.drop_prob > 0:
                    keep = torch.tensor(meta["keep"][name], dtype=torch.float64)
                    state = {"calls": 0}

                    def fwd(x, mod=mod, keep=keep, state=state):
                        n = x.shape[0]
                        k = keep[state["calls"] * n:(state["calls"] + 1) * n]
                        state["calls"] += 1
                        return (mod.scale * x).div(1.0 - mod.drop_prob) * k.view(n, *([1] * (x.dim() - 1)))
                    mod.forward = fwd
        # the float32 run's matching, replayed: both precisions differentiate the same loss function
        rec = meta["cases"][case]["indices"]
        calls = {"n": 0}
        real = model.bipartite_match

        def match(*a, **kw):
            idx, lm = real(*a, **kw)
            want = rec[calls["n"]]
            calls["n"] += 1
            return [(torch.tensor(i), torch.tensor(j)) for i, j in want], lm
        model.bipartite_match = match
        loss = M.run(model, data)
        stride = meta["sample_stride"]
        biggest = max(s[2] for s in meta["cases"][case]["grad_stats"].values())
        errs = []
        for name, p in model.named_parameters():
            gg = p.grad.detach()
            g64 = (gg if gg.numel() <= 2048 else gg.flatten()[::stride]).numpy().copy()
            want = g32[f"{case}/{name}"].astype(np.float64)
            arrs[f"{case}/{name}"] = (g64 - want).astype(np.float32)
            errs.append(float(np.linalg.norm(g64 - want) / (np.linalg.norm(g64) + 1e-4 * biggest)))
        v = np.array(errs)
        print(f"{case}: total_loss f64 {float(loss['total_loss']):.9f} (f32 golden {meta['cases'][case]['losses']['total_loss']:.9f}); "
              f"|ref32 - ref64| / |ref64| per parameter: median {np.median(v):.2e}, 99 % {np.percentile(v, 99):.2e}, max {v.max():.2e}")
        arrs[f"{case}/total_loss"] = np.float64(float(loss["total_loss"]))
    ref_blocks.drop_path = orig_drop
    np.savez_compressed(os.path.join(OUT, "train_step_vidvrd_f64.npz"), **arrs)
    print("wrote train_step_vidvrd_f64.npz", len(arrs), "arrays")


if __name__ == "__main__":
    main()
