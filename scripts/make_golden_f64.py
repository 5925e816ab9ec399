#!/usr/bin/env python3
"""float64 runs of the REAL reference on the inputs of the `_mask_vrd` goldens (build container only, like
scripts/make_golden.py): tests/golden/mask_vrd_f64.npz.

Why: the reference computes in float32 (models/maskvrd.py:161-167, no autocast anywhere); its own outputs therefore sit
some distance e32 = |ref32 - ref64| from the exact result of its arithmetic.  A GEMM mode of the HIP path counts as
"reference-grade" when its distance to ref64 is within 2 x e32 on the same inputs (tests/test_gpu_model.py
::test_reference_grade_against_float64).  Stored: pred_logits / pred_masks as float64 for every case of
mask_vrd_{vidvrd,vidor,vidor_x,vidor_local}.npz, the stored pairs of mask_vrd_vidvrd_b256.npz and of
mask_vrd_vidvrd_cfg2.npz (run as a batch of just those pairs: pairs are independent, and in float64 the batch composition
moves results by ~1e-14).

    python scripts/make_golden_f64.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import CASES, OUT, build, c_in, load_cfg          # noqa: E402  (puts the reference on sys.path)
from make_golden_r2 import B256, CFG2, b256_lengths, cfg2_lengths  # noqa: E402
from oracle import vrd_oracle as O                                  # noqa: E402

torch.set_grad_enabled(False)


def double_model(mc):
    model, _, _ = build(mc)
    return model.double()


def run64(model, x, m):
    out = model._mask_vrd(x.double(), m)
    assert out["pred_logits"].dtype == torch.float64 and out["pred_masks"].dtype == torch.float64
    return out


def main():
    arrs = {}
    cases = dict(CASES)
    cases["vidor"] = ("vidor.yaml", [(2, 512, [512, 301])])          # make_golden_r2.vidor_case
    for name, (fname, shapes) in cases.items():
        cfg, mc = load_cfg(fname)
        model = double_model(mc)
        for (B, T, lens) in shapes:
            t0 = time.time()
            x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=1234 + T)
            out = run64(model, x, m)
            arrs[f"{name}_T{T}_pred_logits"] = out["pred_logits"].numpy()
            arrs[f"{name}_T{T}_pred_masks"] = out["pred_masks"].numpy()
            g = np.load(os.path.join(OUT, f"mask_vrd_{name}.npz"))
            e_l = np.abs(g[f"T{T}_pred_logits"] - arrs[f"{name}_T{T}_pred_logits"]).max()
            e_m = np.abs(g[f"T{T}_pred_masks"] - arrs[f"{name}_T{T}_pred_masks"]).max()
            print(f"{name} T{T}: |ref32 - ref64| logits {e_l:.3e} masks {e_m:.3e}  ({time.time() - t0:.0f} s)")
        if name == "vidvrd":
            for tag, spec, lens_all in (("b256", B256, b256_lengths()), ("cfg2", CFG2, cfg2_lengths())):
                t0 = time.time()
                x, m = O.synth_pairs(spec["B"], c_in(mc), spec["T"], lens_all, seed=spec["seed"])
                e = spec["every"]
                out = run64(model, x[::e].contiguous(), m[::e].contiguous())
                arrs[f"{tag}_pred_logits"] = out["pred_logits"].numpy()
                arrs[f"{tag}_pred_masks"] = out["pred_masks"].numpy()
                g = np.load(os.path.join(OUT, f"mask_vrd_vidvrd_{tag}.npz"))
                e_l = np.abs(g["pred_logits"] - arrs[f"{tag}_pred_logits"]).max()
                e_m = np.abs(g["pred_masks"] - arrs[f"{tag}_pred_masks"]).max()
                print(f"{tag}: |ref32 - ref64| logits {e_l:.3e} masks {e_m:.3e}  ({time.time() - t0:.0f} s)")
    np.savez_compressed(os.path.join(OUT, "mask_vrd_f64.npz"), **arrs)
    print("wrote mask_vrd_f64.npz:", sorted(arrs))


if __name__ == "__main__":
    main()
