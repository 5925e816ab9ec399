#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (runs only in the build container,
where /root/reference exists; the reference never travels to the GPU box).

    python scripts/make_golden.py

Weights are not stored: both sides regenerate them from the name-seeded recipe in
oracle/vrd_oracle.py (checksums of every tensor are stored to prove both sides hold
the same weights).  Inputs are regenerated from seeds as well; only reference OUTPUTS
(and small sub-sampled intermediates) are committed.
"""
import json
import os
import sys

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("VRD_REFERENCE", "/root/reference")
sys.path.insert(0, REF)     # `models.*` = the reference's package
sys.path.insert(1, REPO)
from models.maskvrd import MaskVRD                      # noqa: E402  (reference)
from models import blocks as ref_blocks                 # noqa: E402
from models import local_transformer as ref_lt          # noqa: E402
from oracle import vrd_oracle as O                      # noqa: E402
from oracle.synth import synth_proposal, synth_relations  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.manual_seed(0)

CASES = {   # name -> (config file, [(B, T_pad, lengths)])
    "vidvrd": ("vidvrd.yaml", [
        (4, 96, [96, 95, 41, 2]),
        (3, 144, [144, 97, 130]),
        (2, 288, [288, 201]),
    ]),
    "vidor_x": ("vidor_x.yaml", [
        (2, 512, [512, 333]),
    ]),
    "vidor_local": ("vidor_local.yaml", [
        (2, 512, [512, 77]),
    ]),
}


def load_cfg(fname):
    with open(os.path.join(REF, "configs", fname)) as f:
        cfg = yaml.safe_load(f)
    mc = cfg["model_config"]
    if "with_clip_feature" in cfg.get("dataset_config", {}):
        # eval.py:50-54 / train.py:47-49 copy this flag into the model config
        mc["with_clip_feature"] = cfg["dataset_config"]["with_clip_feature"]
    return cfg, mc


def c_in(mc):
    cc = mc["clip_dim"] if mc.get("with_clip_feature", False) else 0
    return 2 * mc["visual_dim"] + 2 * cc + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]


def build(mc):
    model = MaskVRD(mc, device="cpu").eval()
    keys = [(k, list(v.shape)) for k, v in model.state_dict().items()]
    sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    model.load_state_dict(sd, strict=True)
    return model, keys, sd


def sub(t, step=16):
    """channel-subsampled copy of a (B,C,T) tensor (keeps fixtures small)."""
    return t[:, ::step].contiguous().numpy()


def main():
    for name, (fname, shapes) in CASES.items():
        cfg, mc = load_cfg(fname)
        model, keys, sd = build(mc)
        with open(os.path.join(OUT, f"state_keys_{name}.json"), "w") as f:
            json.dump({"keys": keys, "n_params": int(sum(v.numel() for v in sd.values())),
                       "model_config": mc, "inference_config": cfg["inference_config"]}, f)
        chk = {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}
        with open(os.path.join(OUT, f"param_checksums_{name}.json"), "w") as f:
            json.dump(chk, f)
        arrs = {}
        crit = {}
        for (B, T, lens) in shapes:
            x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=1234 + T)
            feats, masks = model.backbone(x, m)
            fpn, _ = model.neck(feats, masks)
            out = model.predictor(feats[-1], fpn, masks[-1], output_mask=masks[0])
            tag = f"T{T}"
            arrs[f"{tag}_lengths"] = np.asarray(lens)
            arrs[f"{tag}_pred_logits"] = out["pred_logits"].numpy()
            arrs[f"{tag}_pred_masks"] = out["pred_masks"].numpy()
            for i, a in enumerate(out["aux_outputs"]):
                arrs[f"{tag}_aux{i}_pred_logits"] = a["pred_logits"].numpy()
                arrs[f"{tag}_aux{i}_pred_masks"] = a["pred_masks"].numpy()
            # training criterion (forward values) on these predictions: reference matcher + losses
            gp, gm, gs = synth_relations(lens, T, mc["num_classes"], seed=777 + T)
            segs = gs if mc.get("with_fuzzy", False) else None
            idx, lmask = model.bipartite_match(out["pred_logits"], gp, out["pred_masks"], gm, segs, _mask=masks[0])
            ld = model.loss(idx, out["pred_logits"], out["pred_masks"], gp, gm, segs, _mask=masks[0],
                            loss_mask=lmask, aux_outputs=out["aux_outputs"])
            ld["total_loss"] = torch.stack(list(ld.values())).sum()
            crit[tag] = {"seed": 777 + T, "indices": [[i.tolist(), j.tolist()] for i, j in idx],
                         "losses": {k: float(v) for k, v in ld.items()}}
            print(name, tag, "total_loss", crit[tag]["losses"]["total_loss"])
            if T <= 144 or name != "vidvrd":
                for l, ft in enumerate(feats):
                    arrs[f"{tag}_feat{l}"] = sub(ft)
                arrs[f"{tag}_fpn"] = sub(fpn, 8)
            print(name, tag, "logits std", float(out["pred_logits"].std()),
                  "masks std", float(out["pred_masks"].std()))
        if name == "vidvrd":
            # BASELINE config 3 shape: a 24-pair training batch (6 videos x 4 pairs) at T_pad 96, ragged lengths;
            # network in eval mode (no drop-path sampling) -> matcher -> losses, i.e. forward_training's values
            B, T = 24, 96
            lens = torch.randint(2, T + 1, (B,), generator=torch.Generator().manual_seed(2024)).tolist()
            x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=3)
            out = model._mask_vrd(x, m)
            gp, gm, gs = synth_relations(lens, T, mc["num_classes"], max_rel=4, seed=2025)
            idx, lmask = model.bipartite_match(out["pred_logits"], gp, out["pred_masks"], gm, gs, _mask=out["output_mask"])
            ld = model.loss(idx, out["pred_logits"], out["pred_masks"], gp, gm, gs, _mask=out["output_mask"],
                            loss_mask=lmask, aux_outputs=out["aux_outputs"])
            ld["total_loss"] = torch.stack(list(ld.values())).sum()
            crit["train24"] = {"seed": 2025, "lengths": lens, "indices": [[i.tolist(), j.tolist()] for i, j in idx],
                               "losses": {k: float(v) for k, v in ld.items()}}
            print(name, "train24 total_loss", crit["train24"]["losses"]["total_loss"])
        with open(os.path.join(OUT, f"criterion_{name}.json"), "w") as f:
            json.dump(crit, f)
        if "--criterion-only" in sys.argv:
            continue
        np.savez_compressed(os.path.join(OUT, f"mask_vrd_{name}.npz"), **arrs)

        if name == "vidvrd":
            # full forward_test on a synthetic proposal (short and long pairs mixed)
            model._config_eval(cfg["inference_config"])
            data = synth_proposal(6, c_in(mc), 20, 130, seed=4321)
            res = model(data)
            # full box lists are bulky: keep (length, sum) digests per kept triplet
            res["so_trajs_digest"] = [[len(t[0]), float(np.sum(np.asarray(t, dtype=np.float64)))]
                                      for t in res.pop("so_trajs")]
            res["n_pairs"] = len(data["sids"])
            res["pair_lengths"] = [int(f.shape[1]) for f in data["so_features_list"]]
            with open(os.path.join(OUT, "forward_test_vidvrd.json"), "w") as f:
                json.dump(res, f)
            print("forward_test: pairs", res["n_pairs"], "triplets", len(res["triplets"]))

    if "--criterion-only" in sys.argv:
        return
    # ---- per-operator fixtures (real channel widths, small B/T), weights name-seeded ----
    ops = {}
    g = torch.Generator().manual_seed(99)

    def seeded(module, prefix):
        keys = [(f"{prefix}.{k}", list(v.shape)) for k, v in module.state_dict().items()]
        sd = O.synth_state_dict(keys)
        module.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()}, strict=True)
        return module.eval()

    B, C, T = 3, 512, 48
    lens = torch.tensor([48, 31, 5])
    m = (torch.arange(T)[None] < lens[:, None])[:, None]
    x = torch.randn(B, C, T, generator=g) * m
    ops["x"] = x.numpy()
    ops["lens"] = lens.numpy()
    y = torch.randn(B, C, T, generator=g) * m
    ops["y"] = y.numpy()
    for stride in (1, 2):
        mod = seeded(ref_blocks.LocalMaskedMHCA(C, 4, window_size=7, n_qx_stride=stride, n_kv_stride=stride),
                     f"op.local_mhca_s{stride}")
        ops[f"local_mhca_s{stride}"] = mod(x, m)[0].numpy()
        mod = seeded(ref_blocks.TransformerBlock(C, 4, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=7),
                     f"op.block_s{stride}")
        ops[f"block_s{stride}"] = mod(x, m)[0].numpy()
    mod = seeded(ref_blocks.LocalMaskedMHCA(C, 8, window_size=9), "op.local_mhca_w9")
    ops["local_mhca_w9"] = mod(x, m)[0].numpy()
    mod = seeded(ref_lt.MaskedMHCA_QKV(C, 4, n_qx_stride=1, n_kv_stride=1), "op.mhca_qkv")
    ops["mhca_qkv"] = mod(x, y, y, m, m)[0].numpy()
    mod = seeded(ref_lt.MaskedConvTransformerDecoderLayer(C, 4, path_pdrop=0.1, n_qx_stride=1, n_kv_stride=1,
                                                          with_ffn=False, use_local=False), "op.sos")
    ops["sos"] = mod(x, y, m, m)[0].numpy()
    mod = seeded(ref_lt.MaskedConvTransformerDecoderLayer(C, 8, path_pdrop=0.1, n_qx_stride=1, n_kv_stride=1,
                                                          with_ffn=False, use_local=True, win_size=9), "op.sos_local")
    ops["sos_local"] = mod(x, y, m, m)[0].numpy()
    ln = seeded(ref_blocks.LayerNorm(C), "op.ln")
    ops["ln"] = ln(x).numpy()
    cv = seeded(ref_blocks.MaskedConv1D(C, C, 3, padding=1, bias=False), "op.conv3")
    ops["conv3"] = cv(x, m)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **ops)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
