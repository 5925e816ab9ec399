#!/usr/bin/env python3
"""Training-step goldens from the REAL reference (build container only): one optimisation step's loss dict and
parameter gradients on BASELINE config 3's shape -- configs/vidvrd.yaml, 24 pairs (6 videos x 4 pairs), T = max_seq_len
= 96, ragged lengths, seeded ground truth -- by the reference's own forward_training + autograd
(models/maskvrd.py:168-198, train.py:182-186).

Two cases go to tests/golden/train_step_vidvrd.npz / .json:
  nodrop   stochastic depth switched off (models.blocks.drop_path patched to the identity)
  pinned   stochastic depth ON with pinned keep decisions: every AffineDropPath gets a seeded 0/1 vector (name-seeded,
           2B entries; a module called twice -- the stem blocks, shared by subject and object -- takes the first B for
           its first call and the next B for its second), which the test pins into vrdone_amd's modules too.
Stored: the loss dict, the matcher's assignments (final head + 3 auxiliary layers), for EVERY parameter (sum, abs-sum, l2) of its gradient, and the full
gradient of every parameter with at most 2048 elements plus a strided sample of the larger ones.

    PYTORCH_JIT=0 is set here: under this image's torch the reference's @torch.jit.script losses reject the Python float
    num_masks (an ordinary TypeError of the reference).
"""
import os
os.environ.setdefault("PYTORCH_JIT", "0")
import hashlib
import json
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, build, c_in, load_cfg          # noqa: E402  (puts the reference on sys.path)
from models import blocks as ref_blocks                     # noqa: E402  (reference)
from oracle import vrd_oracle as O                          # noqa: E402
from oracle.synth import synth_relations                    # noqa: E402

B, T, SEED_LEN, SEED_X, SEED_GT = 24, 96, 2024, 3, 2025
SAMPLE = 499         # stride of the stored sample of a large gradient
# second golden (round 3): configs/vidor.yaml -- 8 heads of 64 channels, window 9, 50 classes, T = max_seq_len = 512 -- on 6 pairs
# with ragged lengths, stochastic depth off: the backward kernels at the other shipped training shape (the attention
# backward as matrix-core products, the weight-gradient tiles with row chunks that end inside sequences)
VIDOR = dict(config="vidor", B=6, T=512, seed_len=3024, seed_x=5, seed_gt=3025)
VARIANTS = {"vidor_x": dict(B=4, T=512, seed_len=4024, seed_x=11, seed_gt=4025),
            "vidor_local": dict(B=4, T=512, seed_len=5024, seed_x=9, seed_gt=5025)}


def keep_vector(name, n, keep_prob):
    seed = int.from_bytes(hashlib.sha256(("keep:" + name).encode()).digest()[:4], "little")
    g = torch.Generator().manual_seed(seed)
    return torch.floor(keep_prob + torch.rand(n, generator=g))


def batch(mc, B=B, T=T, seed_len=SEED_LEN, seed_x=SEED_X, seed_gt=SEED_GT):
    lens = torch.randint(2, T + 1, (B,), generator=torch.Generator().manual_seed(seed_len)).tolist()
    x, _ = O.synth_pairs(B, c_in(mc), T, lens, seed=seed_x)
    gp, gm, gs = synth_relations(lens, T, mc["num_classes"], max_rel=4, seed=seed_gt)
    return lens, {"so_features_list": [x[i, :, :n].contiguous() for i, n in enumerate(lens)],
                  "preds_list": gp, "masks_list": gm, "segs_list": gs}


def run(model, data):
    model.train()
    model.zero_grad(set_to_none=True)
    with torch.enable_grad():
        loss = model(data)
        loss["total_loss"].backward()
    return loss


def main(config="vidvrd", cases=("nodrop", "pinned"), spec=None):
    cfg, mc = load_cfg(config + ".yaml")
    model, _, _ = build(mc)
    for p in model.parameters():
        p.requires_grad_(True)
    spec = spec or dict(B=B, T=T, seed_len=SEED_LEN, seed_x=SEED_X, seed_gt=SEED_GT)
    lens, data = batch(mc, **spec)
    arrs, meta = {}, {"B": spec["B"], "T": spec["T"], "lengths": lens, "sample_stride": SAMPLE, "cases": {}}

    orig_drop = ref_blocks.drop_path
    for case in cases:
        restore = []
        if case == "nodrop":
            ref_blocks.drop_path = lambda x, drop_prob=0.0, training=False: x
        else:
            ref_blocks.drop_path = orig_drop
            keeps = {}
            for name, mod in model.named_modules():
                if isinstance(mod, ref_blocks.AffineDropPath) and mod.drop_prob > 0:
                    keeps[name] = keep_vector(name, 2 * spec["B"], 1.0 - mod.drop_prob)
                    state = {"calls": 0}

                    def fwd(x, mod=mod, name=name, state=state):
                        n = x.shape[0]
                        k = keeps[name][state["calls"] * n:(state["calls"] + 1) * n]
                        assert k.numel() == n, (name, state["calls"])
                        state["calls"] += 1
                        return (mod.scale * x).div(1.0 - mod.drop_prob) * k.view(n, *([1] * (x.dim() - 1)))
                    restore.append((mod, mod.forward))
                    mod.forward = fwd
            meta["keep"] = {k: v.int().tolist() for k, v in keeps.items()}
        # record the matcher's assignments (main head, then one per auxiliary layer): tests replay them, so that a
        # near-tie in a cost matrix cannot make the two sides differentiate different loss functions
        recorded = []
        real_match = model.bipartite_match

        def match(*a, **kw):
            idx, lm = real_match(*a, **kw)
            recorded.append([[i.tolist(), j.tolist()] for i, j in idx])
            return idx, lm
        model.bipartite_match = match
        loss = run(model, data)
        del model.bipartite_match
        assert len(recorded) == 4
        for mod, f in restore:
            mod.forward = f
        stats = {}
        for name, p in model.named_parameters():
            assert p.grad is not None, name              # DDP's find_unused_parameters=False relies on this (train.py:107)
            g = p.grad.detach()
            stats[name] = [float(g.double().sum()), float(g.double().abs().sum()), float(g.double().norm())]
            arrs[f"{case}/{name}"] = (g if g.numel() <= 2048 else g.flatten()[::SAMPLE]).numpy().copy()
        meta["cases"][case] = {"losses": {k: float(v.detach()) for k, v in loss.items()}, "grad_stats": stats,
                               "indices": recorded}
        print(case, "total_loss", float(loss["total_loss"]), "params with grads", len(stats),
              "max |grad|", max(float(p.grad.abs().max()) for p in model.parameters()))
    ref_blocks.drop_path = orig_drop
    np.savez_compressed(os.path.join(OUT, f"train_step_{config}.npz"), **arrs)
    with open(os.path.join(OUT, f"train_step_{config}.json"), "w") as f:
        json.dump(meta, f)


if __name__ == "__main__":
    torch.set_grad_enabled(True)
    if "--vidor" in sys.argv:
        v = dict(VIDOR)
        main(v.pop("config"), cases=("nodrop",), spec=v)
    elif "--vidor-variants" in sys.argv:     # the CLIP backbone (vidor_x.yaml) and the banded SOS attention (vidor_local.yaml)
        for name, spec in VARIANTS.items():
            main(name, cases=("nodrop",), spec=dict(spec))
    else:
        main()
