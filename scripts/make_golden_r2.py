#!/usr/bin/env python3
"""Round-2 additions to tests/golden/ from the REAL reference (build container only, like scripts/make_golden.py,
whose helpers this reuses; nothing already committed is regenerated):

  forward_test_vidvrd_slices.json   forward_test on > 2 * max_so_pair pairs (3 slices of the reference's slice loop
                                    models/maskvrd.py:208-227, each padding its long pairs to its OWN longest)
  forward_test_vidor_x.json         forward_test under vidor_x.yaml (Q = 10, topk 6, feat_stride 4) with so_offset != 0
  mask_vrd_vidvrd_cfg2.npz          (round 3, --only-cfg2) _mask_vrd on 1024 pairs x 128 frames (T_pad 144): BASELINE config 2 at
                                    its size; logits / masks of every 64th pair
  mask_vrd_vidvrd_b256.npz          _mask_vrd on 256 pairs x T_pad 288 (the batch size that selects the 256 x 256 GEMM
                                    kernel and the padding maps); logits / masks of every 16th pair

    python scripts/make_golden_r2.py [--only-vidor-x]
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, build, c_in, load_cfg          # noqa: E402  (puts the reference on sys.path)
from oracle import vrd_oracle as O                          # noqa: E402
from oracle.synth import synth_proposal                     # noqa: E402

torch.set_grad_enabled(False)

# shared with the tests (tests/golden_cases.py imports nothing from here: the numbers are repeated there)
SLICES = dict(n_tracklets=23, min_len=30, max_len=250, seed=2718, sort_by_length=True)
VIDOR_X = dict(n_tracklets=5, min_len=150, max_len=800, seed=1618, feat_stride=4, random_offset=True)
FORWARD_TEST_VARIANTS = {"vidor": dict(n_tracklets=5, min_len=150, max_len=800, seed=2618, feat_stride=4, random_offset=True),
                         "vidor_local": dict(n_tracklets=5, min_len=150, max_len=700, seed=3618, feat_stride=4, random_offset=True)}
B256 = dict(B=256, T=288, seed=31415, every=16)
CFG2 = dict(B=1024, T=144, frames=128, seed=27182, every=64)       # BASELINE config 2 at its size: 1024 pairs x 128 frames -> T_pad 144


def digest(res, data):
    res["so_trajs_digest"] = [[len(t[0]), float(np.sum(np.asarray(t, dtype=np.float64)))] for t in res.pop("so_trajs")]
    res["n_pairs"] = len(data["sids"])
    res["pair_lengths"] = [int(f.shape[1]) for f in data["so_features_list"]]
    res["so_offset"] = data["so_offset"].tolist()
    return res


def b256_lengths():
    g = torch.Generator().manual_seed(B256["seed"])
    lens = torch.randint(2, B256["T"] + 1, (B256["B"],), generator=g)
    lens[::16] = torch.tensor([288, 287, 256, 255, 200, 129, 97, 96, 64, 33, 32, 31, 17, 3, 2, 288])
    return lens.tolist()


def cfg2_lengths():
    lens = [CFG2["frames"]] * CFG2["B"]
    for i, n in zip(range(0, CFG2["B"], CFG2["every"]), [128, 127, 144, 143, 97, 96, 65, 64, 33, 32, 2, 128, 100, 113, 129, 140]):
        lens[i] = n
    return lens


def cfg2_case():
    """BASELINE config 2 at its size (round 3): `_mask_vrd` of the reference on 1024 pairs x 128 frames, which its eval batching
    pads to T_pad 144 (not a multiple of 32); logits / masks of every 64th pair (those carry the special lengths)."""
    cfg, mc = load_cfg("vidvrd.yaml")
    model, _, _ = build(mc)
    t0 = time.time()
    x, m = O.synth_pairs(CFG2["B"], c_in(mc), CFG2["T"], cfg2_lengths(), seed=CFG2["seed"])
    out = model._mask_vrd(x, m)
    e = CFG2["every"]
    np.savez_compressed(os.path.join(OUT, "mask_vrd_vidvrd_cfg2.npz"), lengths=np.asarray(cfg2_lengths()),
                        pred_logits=out["pred_logits"][::e].numpy(), pred_masks=out["pred_masks"][::e].numpy())
    print("cfg2 case:", tuple(out["pred_logits"].shape), f"{time.time() - t0:.0f} s")


def global_block_case():
    """TransformerBlock with GLOBAL conv attention (n_mha_win_size <= 1, models/blocks.py:1029-1036), stride 1 and 2:
    inputs x / lens of tests/golden/ops.npz, every 4th output channel stored."""
    from models import blocks as ref_blocks
    ops = np.load(os.path.join(OUT, "ops.npz"))
    x, lens = torch.from_numpy(ops["x"]), torch.from_numpy(ops["lens"])
    m = (torch.arange(x.shape[-1])[None] < lens[:, None])[:, None]
    arrs = {}
    for stride in (1, 2):
        mod = ref_blocks.TransformerBlock(512, 4, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=-1)
        keys = [(f"op.block_global_s{stride}.{k}", list(v.shape)) for k, v in mod.state_dict().items()]
        sd = O.synth_state_dict(keys)
        mod.load_state_dict({k[len(f"op.block_global_s{stride}."):]: v for k, v in sd.items()}, strict=True)
        arrs[f"block_global_s{stride}"] = mod.eval()(x, m)[0][:, ::4].contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "ops_r2.npz"), **arrs)
    print("global-attention block case:", {k: v.shape for k, v in arrs.items()})


PROPOSAL_CASES = {   # name -> (synth_raw_video kwargs, dataloader settings)
    "vidvrd": (dict(n_tracklets=8, video_len=120, min_len=12, max_len=100, seed=5),
               dict(feat_stride=1, stride_offset=0, proposal_min_frames=2)),
    "strided": (dict(n_tracklets=6, video_len=400, min_len=30, max_len=380, seed=6),
                dict(feat_stride=4, stride_offset=2, proposal_min_frames=5)),
}


def proposal_case():
    """Eval-time pair construction: the reference dataloader's own `_test_getitem`
    (dataloaders/vidvrd.py:552-715: box clamp, vIoU de-dup, per-pair feature slicing + box features) on a synthetic
    `_prepare_test` output.  Stored per case: the surviving pairs, their lengths / offsets, the 21 box-feature channels of
    every pair in full, and (sum, abs-sum) of every pair's visual slab (a pure gather: checked bit for bit)."""
    from dataloaders.vidvrd import VidVRD
    from oracle.proposal import synth_raw_video
    arrs = {}
    for name, (vid_kw, dl_kw) in PROPOSAL_CASES.items():
        raw = synth_raw_video(**vid_kw)
        ds = object.__new__(VidVRD)                    # the method needs only these attributes, no files
        ds.feat_stride, ds.stride_offset, ds.proposal_min_frames, ds.random_stride = (
            dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], False)
        out = ds._test_getitem(raw)
        feats = out["so_features_list"]
        arrs[f"{name}/sids"], arrs[f"{name}/oids"] = out["sids"].numpy(), out["oids"].numpy()
        arrs[f"{name}/so_offset"] = out["so_offset"].numpy()
        arrs[f"{name}/lens"] = np.asarray([f.shape[1] for f in feats])
        arrs[f"{name}/box_feats"] = torch.cat([f[-21:].T for f in feats], dim=0).numpy()            # (sum L, 21)
        arrs[f"{name}/vis_sums"] = np.asarray([[float(f[:-21].double().sum()), float(f[:-21].double().abs().sum())] for f in feats])
        arrs[f"{name}/boxes_clamped"] = torch.cat(out["bboxes_list"], dim=0).numpy()
        print("proposal case", name, "pairs", len(feats), "of", len(raw["sids"]), "lens", arrs[f"{name}/lens"].tolist()[:12])
        assert len(feats) < len(raw["sids"])             # the de-dup (and the length filters) removed something
    # the on-disk layout: the reference's `_prepare_test` on two synthetic per-video pickles
    import tempfile
    from oracle.proposal import write_synth_pickles
    with tempfile.TemporaryDirectory() as tmp:
        info, feat = write_synth_pickles(tmp)
        ds = object.__new__(VidVRD)
        ds.split, ds.info_dir, ds.test_boxfeatures_dir = "test", os.path.dirname(info), os.path.dirname(feat)
        out = ds._prepare_test("vid0")
    arrs["pickles/sids"], arrs["pickles/oids"] = out["sids"].numpy(), out["oids"].numpy()
    arrs["pickles/traj_durations"] = out["traj_durations"].numpy()
    arrs["pickles/visual_features"] = torch.cat(out["visual_features_list"]).numpy()
    arrs["pickles/bboxes"] = torch.cat(out["bboxes_list"]).numpy()
    print("pickle case: tracklets", len(out["bboxes_list"]), "pairs", len(out["sids"]))
    np.savez_compressed(os.path.join(OUT, "proposal.npz"), **arrs)


def train_data_case():
    """Training-side data path: the reference dataloader's own `_prepare_train`, `_train_getitem` (seeded
    `random`) and `apply_policy` on a synthetic annotation + ground-truth feature file.  Stored: the cache entry (keys,
    intervals, merged relations, classes, concatenated features / boxes) and, per (feat_stride, max_seq_len) setting, the
    sample lists in full (C_in is small here)."""
    import json
    import random
    import tempfile
    from dataloaders.vidvrd import VidVRD
    from oracle.proposal import write_synth_train_files
    out = {}
    arrs = {}
    with tempfile.TemporaryDirectory() as tmp:
        anno_dir, feat_dir, ent, pred = write_synth_train_files(tmp)
        ds = object.__new__(VidVRD)
        ds.split, ds.video_ann_dir, ds.gt_boxfeatures_dir = "train", anno_dir, feat_dir
        ds.entity_cat_name_to_id, ds.pred_cat_name_to_id = ent, pred
        video = ds._prepare_train("vid0")
        out["relation_keys"] = video["relation_keys"]
        out["relation_merged"] = [[list(k), v] for k, v in video["relation_merged"].items()]
        out["traj_intervals"] = {str(k): v for k, v in video["traj_intervals"].items()}
        out["entity_classes"] = {str(k): v for k, v in video["entity_classes"].items()}
        out["video_hw"] = list(video["video_hw"])
        arrs["visual"] = torch.cat([t for k in sorted(video["visual_features"]) for t in video["visual_features"][k]]).numpy()
        arrs["boxes"] = torch.cat([t for k in sorted(video["entity_bboxes"]) for t in video["entity_bboxes"][k]]).numpy()
        out["samples"] = {}
        for name, (stride, max_len, cut, max_preds, dur, seed) in TRAIN_DATA_CASES.items():
            ds.feat_stride, ds.max_seq_len, ds.cut_max_preds, ds.proposal_max_preds = stride, max_len, cut, max_preds
            random.seed(seed)
            sample = ds._train_getitem(video, dur)
            out["samples"][name] = {"n": len(sample.get("so_features_list", [])),
                                    "lens": [int(f.shape[1]) for f in sample.get("so_features_list", [])],
                                    "preds": [p_.tolist() for p_ in sample.get("preds_list", [])],
                                    "segs": [s_.tolist() for s_ in sample.get("segs_list", [])]}
            for i, (f, m) in enumerate(zip(sample.get("so_features_list", []), sample.get("masks_list", []))):
                arrs[f"{name}/feat{i}"], arrs[f"{name}/mask{i}"] = f.numpy(), m.numpy()
            print("train data case", name, out["samples"][name]["lens"], out["samples"][name]["preds"])
        # the VidOR loader's sample construction with CLIP features (dataloaders/vidor.py:335-478) on the same cache entry
        from dataloaders.vidor import VidOR
        g = torch.Generator().manual_seed(77)
        video["clip_features"] = {k: [torch.randn(t.shape[0], 8, generator=g) for t in per] for k, per in video["visual_features"].items()}
        dv = object.__new__(VidOR)
        dv.feat_stride, dv.max_seq_len, dv.cut_max_preds, dv.proposal_max_preds, dv.with_clip_feature = 4, 96, False, 0, True
        random.seed(8)
        sample = dv._train_getitem(video, None)
        out["samples"]["vidor_clip"] = {"n": len(sample["so_features_list"]), "lens": [int(f.shape[1]) for f in sample["so_features_list"]],
                                        "preds": [p_.tolist() for p_ in sample["preds_list"]], "segs": None}
        for i, (f, m) in enumerate(zip(sample["so_features_list"], sample["masks_list"])):
            arrs[f"vidor_clip/feat{i}"], arrs[f"vidor_clip/mask{i}"] = f.numpy(), m.numpy()
        arrs["clip"] = torch.cat([t for k in sorted(video["clip_features"]) for t in video["clip_features"][k]]).numpy()
        print("train data case vidor_clip", out["samples"]["vidor_clip"]["lens"], "C_in", sample["so_features_list"][0].shape[0])
    ds.video_num_pairs = [["a", 3], ["b", 9], ["c", 1], ["d", 4], ["e", 13], ["f", 2]]
    ds.num_pairs = 6
    ds.apply_policy()
    out["policy"] = {"video_num_pairs": ds.video_num_pairs, "num_pairs": ds.num_pairs, "policy": ds.policy}
    with open(os.path.join(OUT, "train_data.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(OUT, "train_data.npz"), **arrs)


TRAIN_DATA_CASES = {   # name -> (feat_stride, max_seq_len, cut_max_preds, proposal_max_preds, pair_duration, random seed)
    "stride1": (1, 96, False, 0, None, 3),
    "stride1_crop": (1, 24, False, 0, None, 4),          # every pair longer than max_seq_len: random crops
    "stride4": (4, 96, False, 0, None, 5),               # random sub-sampling offsets
    "cut": (1, 96, True, 1, (1, 3), 6),                  # pairs with more than one relation dropped, keys 1..2 only
}


ABS_PE_CASES = {   # name -> (config file, [(B, T, lengths)])
    "vidvrd": ("vidvrd.yaml", [(3, 96, [96, 50, 7]), (3, 288, [288, 201, 30])]),          # T = max_len and T > max_len: interpolated table
    "vidor_x": ("vidor_x.yaml", [(2, 128, [128, 77])]),                                       # T < max_len, CLIP backbone
}


def abs_pe_case():
    """`use_abs_pe: True` (no shipped config sets it): the reference's `_mask_vrd` with the sinusoid position table added behind
    the visual embedding (backbones.py:180-196; CLIP variant :368-384) -- logits / masks of the last layer."""
    arrs = {}
    for name, (fname, shapes) in ABS_PE_CASES.items():
        cfg, mc = load_cfg(fname)
        mc = dict(mc, use_abs_pe=True)
        model, _, _ = build(mc)
        for (B, T, lens) in shapes:
            x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=4321 + T)
            out = model._mask_vrd(x, m)
            arrs[f"{name}/T{T}_pred_logits"] = out["pred_logits"].numpy()
            arrs[f"{name}/T{T}_pred_masks"] = out["pred_masks"].numpy()
            print("abs pe case", name, T, "logits std", float(out["pred_logits"].std()))
    np.savez_compressed(os.path.join(OUT, "abs_pe.npz"), **arrs)


REL_PE_CASES = {   # name -> (config file, [(B, T, lengths)])
    "vidvrd": ("vidvrd.yaml", [(3, 96, [96, 50, 7])]),               # stem / branch blocks (LocalMaskedMHCA)
    "vidor_local": ("vidor_local.yaml", [(2, 128, [128, 77])]),      # + the fusion layers' local q/k/v attention
}


def rel_pe_case():
    """`use_rel_pe: True` (no shipped config sets it): a learnable bias per (head, window slot) on the banded attention's
    scores (blocks.py:739-743,957-958; local_transformer.py:373-376,591-592).  The extra parameters' names and shapes go to
    rel_pe_keys.json; their values are the name-seeded synthetic ones like every other parameter."""
    arrs, extra = {}, {}
    for name, (fname, shapes) in REL_PE_CASES.items():
        cfg, mc = load_cfg(fname)
        mc = dict(mc, use_rel_pe=True)
        model, keys, _ = build(mc)
        extra[name] = [k for k in keys if k[0].endswith("rel_pe")]
        for (B, T, lens) in shapes:
            x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=8765 + T)
            out = model._mask_vrd(x, m)
            arrs[f"{name}/T{T}_pred_logits"] = out["pred_logits"].numpy()
            arrs[f"{name}/T{T}_pred_masks"] = out["pred_masks"].numpy()
            print("rel pe case", name, T, len(extra[name]), "tables; logits std", float(out["pred_logits"].std()))
    np.savez_compressed(os.path.join(OUT, "rel_pe.npz"), **arrs)
    with open(os.path.join(OUT, "rel_pe_keys.json"), "w") as f:
        json.dump(extra, f)


def vidor_case():
    """configs/vidor.yaml -- the fourth shipped config: the plain backbone at the VidOR settings (8 heads, window 9, T 512,
    9 queries, global SOS attention, no CLIP slabs) -- in the format of scripts/make_golden.py's cases: state keys + config,
    `_mask_vrd` outputs with the auxiliary layers, channel-subsampled backbone / pyramid features."""
    from make_golden import sub
    cfg, mc = load_cfg("vidor.yaml")
    model, keys, sd = build(mc)
    with open(os.path.join(OUT, "state_keys_vidor.json"), "w") as f:
        json.dump({"keys": keys, "n_params": int(sum(v.numel() for v in sd.values())),
                   "model_config": mc, "inference_config": cfg["inference_config"]}, f)
    B, T, lens = 2, 512, [512, 301]
    x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=1234 + T)
    feats, masks = model.backbone(x, m)
    fpn, _ = model.neck(feats, masks)
    out = model.predictor(feats[-1], fpn, masks[-1], output_mask=masks[0])
    tag = f"T{T}"
    arrs = {f"{tag}_lengths": np.asarray(lens), f"{tag}_pred_logits": out["pred_logits"].numpy(),
            f"{tag}_pred_masks": out["pred_masks"].numpy(), f"{tag}_fpn": sub(fpn, 8)}
    for i, a in enumerate(out["aux_outputs"]):
        arrs[f"{tag}_aux{i}_pred_logits"] = a["pred_logits"].numpy()
        arrs[f"{tag}_aux{i}_pred_masks"] = a["pred_masks"].numpy()
    for l, ft in enumerate(feats):
        arrs[f"{tag}_feat{l}"] = sub(ft)
    np.savez_compressed(os.path.join(OUT, "mask_vrd_vidor.npz"), **arrs)
    print("vidor case: logits std", float(out["pred_logits"].std()), "masks std", float(out["pred_masks"].std()))


def main():
    if "--only-cfg2" in sys.argv:
        return cfg2_case()
    if "--only-vidor" in sys.argv:
        return vidor_case()
    if "--only-rel-pe" in sys.argv:
        return rel_pe_case()
    if "--only-abs-pe" in sys.argv:
        return abs_pe_case()
    if "--only-train-data" in sys.argv:
        return train_data_case()
    if "--only-proposal" in sys.argv:
        return proposal_case()
    if "--only-global-block" in sys.argv:
        return global_block_case()
    if "--only-vidor-x" not in sys.argv:
        vidvrd_cases()
    vidor_x_case()
    global_block_case()
    proposal_case()


def vidvrd_cases():
    cfg, mc = load_cfg("vidvrd.yaml")
    model, _, _ = build(mc)
    model._config_eval(cfg["inference_config"])

    t0 = time.time()
    data = synth_proposal(c_in=c_in(mc), **SLICES)
    lens = [int(f.shape[1]) for f in data["so_features_list"]]
    P, S = len(lens), mc["max_so_pair"]
    d = model.max_div_factor
    t_long = [(max(lens[s:s + S] + [mc["max_seq_len"]]) + d - 1) // d * d for s in range(0, P, S)]
    print("slices case: pairs", P, "slice T_long", t_long, "long pairs", sum(n > mc["max_seq_len"] for n in lens))
    assert P > 2 * S and len(set(t_long)) >= 2, "need >= 3 slices with different padded lengths"
    res = digest(model(data), data)
    res["slice_t_long"] = t_long
    with open(os.path.join(OUT, "forward_test_vidvrd_slices.json"), "w") as f:
        json.dump(res, f)
    print("  triplets", len(res["triplets"]), f"{time.time() - t0:.0f} s")

    t0 = time.time()
    x, m = O.synth_pairs(B256["B"], c_in(mc), B256["T"], b256_lengths(), seed=B256["seed"])
    out = model._mask_vrd(x, m)
    e = B256["every"]
    np.savez_compressed(os.path.join(OUT, "mask_vrd_vidvrd_b256.npz"),
                        lengths=np.asarray(b256_lengths()), pred_logits=out["pred_logits"][::e].numpy(),
                        pred_masks=out["pred_masks"][::e].numpy())
    print("b256 case:", tuple(out["pred_logits"].shape), f"{time.time() - t0:.0f} s")


def vidor_x_case():
    t0 = time.time()
    cfg, mc = load_cfg("vidor_x.yaml")
    model, _, _ = build(mc)
    model._config_eval(cfg["inference_config"])
    data = synth_proposal(c_in=c_in(mc), **VIDOR_X)
    print("vidor_x case: pairs", len(data["sids"]), "lengths", sorted(int(f.shape[1]) for f in data["so_features_list"]),
          "offsets", sorted(set(data["so_offset"].tolist())))
    assert len(set(data["so_offset"].tolist())) >= 3
    res = digest(model(data), data)
    with open(os.path.join(OUT, "forward_test_vidor_x.json"), "w") as f:
        json.dump(res, f)
    print("  triplets", len(res["triplets"]), f"{time.time() - t0:.0f} s")


def forward_test_variants():
    """forward_test under vidor.yaml (plain backbone, 8 heads) and vidor_local.yaml (banded SOS attention): the same kind of
    synthetic video as the vidor_x case (feat_stride 4, so_offset in 0..3, pairs longer than max_seq_len), other seeds."""
    for name, spec in FORWARD_TEST_VARIANTS.items():
        t0 = time.time()
        cfg, mc = load_cfg(name + ".yaml")
        model, _, _ = build(mc)
        model._config_eval(cfg["inference_config"])
        data = synth_proposal(c_in=c_in(mc), **spec)
        print(name, "forward_test case: pairs", len(data["sids"]), "lengths", sorted(int(f.shape[1]) for f in data["so_features_list"]))
        res = digest(model(data), data)
        with open(os.path.join(OUT, f"forward_test_{name}.json"), "w") as f:
            json.dump(res, f)
        print("  triplets", len(res["triplets"]), f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    if "--only-forward-test-variants" in sys.argv:
        forward_test_variants()
    else:
        main()
