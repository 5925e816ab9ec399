#!/usr/bin/env python3
"""Max |difference| of the HIP path vs the reference's golden outputs, per precision mode (runs on the GPU box)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import GOLDEN, load_case  # noqa: E402
from oracle import vrd_oracle as O  # noqa: E402
from vrdone_amd import ops  # noqa: E402
from vrdone_amd.models.maskvrd import MaskVRD  # noqa: E402

torch.set_grad_enabled(False)
for name, Ts in (("vidvrd", (96, 144, 288)), ("vidor_x", (512,)), ("vidor_local", (512,))):
    mc, ic, keys = load_case(name)
    sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    model = MaskVRD(mc, device="cuda")
    model.load_state_dict(sd)
    model = model.cuda().eval()
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    cc = mc["clip_dim"] if mc.get("with_clip_feature", False) else 0
    cin = 2 * mc["visual_dim"] + 2 * cc + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]
    for T in Ts:
        lens = g[f"T{T}_lengths"].tolist()
        x, m = O.synth_pairs(len(lens), cin, T, lens, seed=1234 + T)
        for mode in ("f32", "bf16x3"):
            ops.set_precision(mode)
            out = model._mask_vrd(x.cuda(), m.cuda(), with_aux=False)
            dl = float(np.abs(out["pred_logits"].cpu().numpy() - g[f"T{T}_pred_logits"]).max())
            dm = float(np.abs(out["pred_masks"].cpu().numpy() - g[f"T{T}_pred_masks"]).max())
            print(f"{name:12s} T={T:3d} {mode:7s} max|dlogits| {dl:.2e}   max|dmasks| {dm:.2e}", flush=True)
ops.set_precision("f32")
