"""The oracle (oracle/vrd_oracle.py) against golden vectors emitted by the real reference
(scripts/make_golden.py).  CPU only.  Tolerances: the reference's own fp32-vs-fp64
discrepancy is 4e-6 (logits) / 3e-5 (masks)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from oracle import vrd_oracle as O
from oracle.synth import synth_proposal

torch.set_grad_enabled(False)
LOGIT_TOL, MASK_TOL = 2e-5, 2e-4


def c_in(mc):
    cc = mc["clip_dim"] if mc.get("with_clip_feature", False) else 0
    return 2 * mc["visual_dim"] + 2 * cc + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]


@pytest.fixture(scope="module")
def weights():
    cache = {}

    def get(name):
        if name not in cache:
            mc, ic, keys = load_case(name)
            cache[name] = (mc, ic, O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]))
        return cache[name]
    return get


@pytest.mark.parametrize("name", ["vidvrd", "vidor_x", "vidor_local"])
def test_param_checksums(name, weights):
    _, _, sd = weights(name)
    with open(os.path.join(GOLDEN, f"param_checksums_{name}.json")) as f:
        chk = json.load(f)
    assert list(chk.keys()) == list(sd.keys())
    for k, (s, a) in chk.items():
        assert abs(float(sd[k].double().sum()) - s) <= 1e-9 * max(1.0, a), k
        assert abs(float(sd[k].double().abs().sum()) - a) <= 1e-9 * max(1.0, a), k


@pytest.mark.parametrize("name,T", [("vidvrd", 96), ("vidvrd", 144), ("vidvrd", 288),
                                    ("vidor_x", 512), ("vidor_local", 512), ("vidor", 512)])
def test_mask_vrd_matches_reference(name, T, weights):
    mc, _, sd = weights(name)
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    lens = g[f"T{T}_lengths"].tolist()
    x, m = O.synth_pairs(len(lens), c_in(mc), T, lens, seed=1234 + T)
    feats, masks = O.backbone(sd, mc, x, m)
    fpn, _ = O.neck(sd, mc, feats, masks)
    out = O.predictor(sd, mc, feats[-1], fpn, masks[-1], masks[0])
    if f"T{T}_feat0" in g:
        for l, ft in enumerate(feats):
            np.testing.assert_allclose(ft[:, ::16].numpy(), g[f"T{T}_feat{l}"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(fpn[:, ::8].numpy(), g[f"T{T}_fpn"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g[f"T{T}_pred_logits"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(out["pred_masks"].numpy(), g[f"T{T}_pred_masks"], atol=MASK_TOL, rtol=0)
    for i, a in enumerate(out["aux_outputs"]):
        np.testing.assert_allclose(a["pred_logits"].numpy(), g[f"T{T}_aux{i}_pred_logits"], atol=LOGIT_TOL, rtol=0)
        np.testing.assert_allclose(a["pred_masks"].numpy(), g[f"T{T}_aux{i}_pred_masks"], atol=MASK_TOL, rtol=0)


def _op_sd(prefix, ref_keys):
    return O.synth_state_dict(ref_keys)


def test_operators_match_reference():
    g = np.load(os.path.join(GOLDEN, "ops.npz"))
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    lens = torch.from_numpy(g["lens"])
    B, C, T = x.shape
    m = (torch.arange(T)[None] < lens[:, None])[:, None]

    def attn_keys(p, ks=(3, 3, 3)):
        keys = []
        for n, k in zip(("query", "key", "value"), ks):
            keys += [(f"{p}.{n}_conv.conv.weight", (C, 1, k)), (f"{p}.{n}_norm.weight", (1, C, 1)),
                     (f"{p}.{n}_norm.bias", (1, C, 1))]
        for n in ("key", "query", "value", "proj"):
            keys += [(f"{p}.{n}.weight", (C, C, 1)), (f"{p}.{n}.bias", (C,))]
        return keys

    def ln_keys(p):
        return [(f"{p}.weight", (1, C, 1)), (f"{p}.bias", (1, C, 1))]

    for stride in (1, 2):
        p = f"op.local_mhca_s{stride}"
        sd = O.synth_state_dict(attn_keys(p))
        out, _ = O.local_mhca(sd, p, x, m, 4, 7, stride)
        np.testing.assert_allclose(out.numpy(), g[f"local_mhca_s{stride}"], atol=2e-5, rtol=0)
        p = f"op.block_s{stride}"
        keys = ln_keys(f"{p}.ln1") + ln_keys(f"{p}.ln2") + attn_keys(f"{p}.attn") + [
            (f"{p}.mlp.0.weight", (4 * C, C, 1)), (f"{p}.mlp.0.bias", (4 * C,)),
            (f"{p}.mlp.3.weight", (C, 4 * C, 1)), (f"{p}.mlp.3.bias", (C,)),
            (f"{p}.drop_path_attn.scale", (1, C, 1)), (f"{p}.drop_path_mlp.scale", (1, C, 1))]
        out, _ = O.transformer_block(O.synth_state_dict(keys), p, x, m, 4, 7, stride)
        np.testing.assert_allclose(out.numpy(), g[f"block_s{stride}"], atol=5e-5, rtol=0)
    g2 = np.load(os.path.join(GOLDEN, "ops_r2.npz"))
    for stride in (1, 2):            # global conv attention inside the block (n_mha_win_size <= 1, blocks.py:1029-1036)
        p = f"op.block_global_s{stride}"
        keys = ln_keys(f"{p}.ln1") + ln_keys(f"{p}.ln2") + attn_keys(f"{p}.attn") + [
            (f"{p}.mlp.0.weight", (4 * C, C, 1)), (f"{p}.mlp.0.bias", (4 * C,)),
            (f"{p}.mlp.3.weight", (C, 4 * C, 1)), (f"{p}.mlp.3.bias", (C,)),
            (f"{p}.drop_path_attn.scale", (1, C, 1)), (f"{p}.drop_path_mlp.scale", (1, C, 1))]
        out, _ = O.transformer_block(O.synth_state_dict(keys), p, x, m, 4, -1, stride)
        np.testing.assert_allclose(out[:, ::4].numpy(), g2[f"block_global_s{stride}"], atol=5e-5, rtol=0)
    p = "op.local_mhca_w9"
    out, _ = O.local_mhca(O.synth_state_dict(attn_keys(p)), p, x, m, 8, 9, 1)
    np.testing.assert_allclose(out.numpy(), g["local_mhca_w9"], atol=2e-5, rtol=0)
    p = "op.mhca_qkv"
    out, _ = O.mhca_qkv(O.synth_state_dict(attn_keys(p)), p, x, y, y, m, m, 4)
    np.testing.assert_allclose(out.numpy(), g["mhca_qkv"], atol=2e-5, rtol=0)
    for p, heads, hw in (("op.sos", 4, None), ("op.sos_local", 8, 4)):
        keys = ln_keys(f"{p}.ln1") + ln_keys(f"{p}.ln2") + attn_keys(f"{p}.self_attn") + \
            attn_keys(f"{p}.multihead_attn") + [(f"{p}.drop_path_attn1.scale", (1, C, 1)),
                                                 (f"{p}.drop_path_attn2.scale", (1, C, 1))]
        out, _ = O.decoder_layer(O.synth_state_dict(keys), p, x, y, m, m, heads, half_win=hw)
        np.testing.assert_allclose(out.numpy(), g[p[3:]], atol=5e-5, rtol=0)
    sd = O.synth_state_dict(ln_keys("op.ln"))
    np.testing.assert_allclose(O.channel_ln(x, sd["op.ln.weight"], sd["op.ln.bias"]).numpy(), g["ln"], atol=1e-6, rtol=0)
    sd = O.synth_state_dict([("op.conv3.conv.weight", (C, C, 3))])
    out, _ = O.masked_conv1d(x, m, sd["op.conv3.conv.weight"])
    np.testing.assert_allclose(out.numpy(), g["conv3"], atol=1e-5, rtol=0)


def test_forward_test_matches_reference(weights):
    mc, ic, sd = weights("vidvrd")
    with open(os.path.join(GOLDEN, "forward_test_vidvrd.json")) as f:
        ref = json.load(f)
    data = synth_proposal(6, c_in(mc), 20, 130, seed=4321)
    assert len(data["sids"]) == ref["n_pairs"]
    assert [int(f.shape[1]) for f in data["so_features_list"]] == ref["pair_lengths"]
    res = O.forward_test(sd, mc, ic, data)
    assert res["triplets"] == ref["triplets"]
    assert res["pred_durations"] == ref["pred_durations"]
    assert res["so_tids"] == ref["so_tids"]
    np.testing.assert_allclose(res["triple_scores"], ref["triple_scores"], atol=1e-5, rtol=0)
    np.testing.assert_allclose(res["triple_scores_avg"], ref["triple_scores_avg"], atol=1e-5, rtol=0)
    dig = [[len(t[0]), float(np.sum(np.asarray(t, dtype=np.float64)))] for t in res["so_trajs"]]
    np.testing.assert_allclose(dig, ref["so_trajs_digest"], rtol=1e-9)


def test_forward_test_many_slices_matches_reference(weights):
    """> 2 * max_so_pair pairs: three slices of the reference's slice loop (models/maskvrd.py:208-227), whose long pairs
    are padded to 192 / 240 / 240 frames; pins the per-slice padding rule the HIP path's bucketing must reproduce."""
    from golden_cases import SLICES, compare_forward_test
    mc, ic, sd = weights("vidvrd")
    with open(os.path.join(GOLDEN, "forward_test_vidvrd_slices.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **SLICES)
    assert [int(f.shape[1]) for f in data["so_features_list"]] == ref["pair_lengths"]
    assert len(data["sids"]) > 2 * mc["max_so_pair"] and len(set(ref["slice_t_long"])) >= 2
    res = O.forward_test(sd, mc, ic, data)
    np.testing.assert_allclose(res["triple_scores"], ref["triple_scores"], atol=1e-5, rtol=0)
    compare_forward_test(res, ref, ic["n_max_pair"], 1e-5, slack=0)


@pytest.mark.parametrize("name", ["vidor", "vidor_local"])
def test_forward_test_vidor_variants_match_reference(weights, name):
    """forward_test under vidor.yaml (plain backbone, 8 heads of 64 channels) and vidor_local.yaml (banded SOS attention;
    20 pairs x 9 queries = 180 candidates, fewer than n_max_pair): feat_stride 4, so_offset in 0..3."""
    from golden_cases import FORWARD_TEST_VARIANTS, compare_forward_test
    mc, ic, sd = weights(name)
    with open(os.path.join(GOLDEN, f"forward_test_{name}.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **FORWARD_TEST_VARIANTS[name])
    res = O.forward_test(sd, mc, ic, data)
    compare_forward_test(res, ref, ic["n_max_pair"], 1e-5, slack=0)


def test_forward_test_vidor_x_with_offsets_matches_reference(weights):
    """vidor_x.yaml (Q = 10, topk 6, feat_stride 4, pred_min_frames 5) with so_offset in {0..3}
    (models/maskvrd.py:283-299: start = first * feat_stride + offset)."""
    from golden_cases import VIDOR_X, compare_forward_test
    mc, ic, sd = weights("vidor_x")
    with open(os.path.join(GOLDEN, "forward_test_vidor_x.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **VIDOR_X)
    assert [int(f.shape[1]) for f in data["so_features_list"]] == ref["pair_lengths"]
    assert data["so_offset"].tolist() == ref["so_offset"] and len(set(ref["so_offset"])) >= 3
    res = O.forward_test(sd, mc, ic, data)
    compare_forward_test(res, ref, ic["n_max_pair"], 1e-5, slack=0)


def test_mask_vrd_b256_matches_reference(weights):
    """The reference ran the 256-pair batch; pairs are independent, so the oracle recomputes only the 16 stored ones."""
    from golden_cases import B256, b256_lengths
    mc, _, sd = weights("vidvrd")
    g = np.load(os.path.join(GOLDEN, "mask_vrd_vidvrd_b256.npz"))
    lens = b256_lengths()
    assert g["lengths"].tolist() == lens
    x, m = O.synth_pairs(B256["B"], c_in(mc), B256["T"], lens, seed=B256["seed"])
    e = B256["every"]
    out = O.mask_vrd(sd, mc, x[::e].contiguous(), m[::e].contiguous(), with_aux=False)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g["pred_logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pred_masks"].numpy(), g["pred_masks"], atol=2e-4, rtol=0)


def test_mask_vrd_cfg2_matches_reference(weights):
    """BASELINE config 2 at its size: the reference ran 1024 pairs x 128 frames (T_pad 144); the oracle recomputes the 16
    stored pairs (every 64th: the ones with the special lengths)."""
    from golden_cases import CFG2, cfg2_lengths
    mc, _, sd = weights("vidvrd")
    g = np.load(os.path.join(GOLDEN, "mask_vrd_vidvrd_cfg2.npz"))
    lens = cfg2_lengths()
    assert g["lengths"].tolist() == lens
    x, m = O.synth_pairs(CFG2["B"], c_in(mc), CFG2["T"], lens, seed=CFG2["seed"])
    e = CFG2["every"]
    out = O.mask_vrd(sd, mc, x[::e].contiguous(), m[::e].contiguous(), with_aux=False)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g["pred_logits"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(out["pred_masks"].numpy(), g["pred_masks"], atol=2e-4, rtol=0)


def test_training_step_gradients_match_reference(weights):
    """One training step's losses and parameter gradients (reference forward_training + autograd, stochastic depth off,
    tests/golden/train_step_vidvrd.*) against autograd through the oracle's network and the product's criterion
    (vrdone_amd.models.maskvrd.MaskVRD.criterion is plain tensor code and runs on CPU tensors): pins the golden, the
    oracle as a gradient reference for the per-op GPU tests, and the differentiability of the loss code."""
    from golden_cases import compare_grads, replay_matching, train_batch
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, sd = weights("vidvrd")
    with open(os.path.join(GOLDEN, "train_step_vidvrd.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "train_step_vidvrd.npz"))
    lens, x, m, data = train_batch(mc, c_in(mc))
    assert lens == meta["lengths"]
    model = MaskVRD(mc, device="cpu").train()
    differing = replay_matching(model, meta["cases"]["nodrop"]["indices"])
    names = [n for n, _ in model.named_parameters()]
    leaves = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    with torch.enable_grad():
        pred = O.mask_vrd(leaves, mc, x, m, with_aux=True)
        loss = model.criterion(pred, data)
        loss["total_loss"].backward()
    want = meta["cases"]["nodrop"]["losses"]
    assert set(loss) == set(want)
    for k, v in want.items():
        assert abs(float(loss[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    # the product's matcher finds the reference's assignments on these predictions, except near-ties of very short pairs
    assert len(differing) == 4 and all(lens[n] < 16 for call in differing for n in call)
    compare_grads([(n, leaves[n].grad) for n in names], g, meta, "nodrop", rtol=1e-3, median_tol=2e-5)


@pytest.mark.parametrize("name", ["vidor", "vidor_x", "vidor_local"])
def test_training_step_gradients_vidor_match_reference(weights, name):
    """The same for the T = 512 configs: vidor.yaml (8 heads of 64 channels, 6 ragged pairs), vidor_x.yaml (CLIP backbone),
    vidor_local.yaml (banded SOS attention); tests/golden/train_step_<name>.*."""
    from golden_cases import TRAIN_SPECS, compare_grads, replay_matching, train_batch
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, sd = weights(name)
    with open(os.path.join(GOLDEN, f"train_step_{name}.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, f"train_step_{name}.npz"))
    lens, x, m, data = train_batch(mc, c_in(mc), spec=TRAIN_SPECS[name])
    assert lens == meta["lengths"]
    model = MaskVRD(mc, device="cpu").train()
    differing = replay_matching(model, meta["cases"]["nodrop"]["indices"])
    names = [n for n, _ in model.named_parameters()]
    leaves = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    with torch.enable_grad():
        pred = O.mask_vrd(leaves, mc, x, m, with_aux=True)
        loss = model.criterion(pred, data)
        loss["total_loss"].backward()
    want = meta["cases"]["nodrop"]["losses"]
    assert set(loss) == set(want)
    for k, v in want.items():
        assert abs(float(loss[k]) - v) <= 1e-4 * max(1.0, abs(v)), k
    assert len(differing) == 4 and all(len(call) <= 2 for call in differing), differing
    # (atol_frac: a key projection's bias shifts every score of a query equally, its gradient is rounding noise on both sides)
    compare_grads([(n, leaves[n].grad) for n in names], g, meta, "nodrop", rtol=1e-3, atol_frac=1e-5, median_tol=2e-5)


def test_preprocess_eval_shapes():
    mc, _, _ = load_case("vidvrd")
    assert O.max_div_factor(mc) == 48
    feats = [torch.randn(7, L) for L in (10, 96, 97, 200)]
    short, long_ = O.preprocess_eval(mc, feats)
    assert short[0].shape == (2, 7, 96) and short[2] == [0, 1]
    assert long_[0].shape == (2, 7, 240) and long_[2] == [2, 3]
    assert short[1].sum().item() == 106 and long_[1].sum().item() == 297
    mc2, _, _ = load_case("vidor_x")
    assert O.max_div_factor(mc2) == 64


# ---------------------------------------------------------------------------------------------------
# training criterion (forward values): the reference's matcher + losses on ITS OWN stored predictions
# ---------------------------------------------------------------------------------------------------
CRIT_CASES = [("vidvrd", 96), ("vidvrd", 144), ("vidvrd", 288), ("vidor_x", 512), ("vidor_local", 512)]


def stored_predictions(name, T):
    """The reference's predictions of the golden case (all decoder layers) + the seeded ground truth."""
    from oracle.synth import synth_relations
    mc, _, _ = load_case(name)
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    with open(os.path.join(GOLDEN, f"criterion_{name}.json")) as f:
        want = json.load(f)[f"T{T}"]
    lens = g[f"T{T}_lengths"].tolist()
    t = lambda k: torch.from_numpy(g[f"T{T}_{k}"])        # noqa: E731
    pred = {"pred_logits": t("pred_logits"), "pred_masks": t("pred_masks"),
            "aux_outputs": [{"pred_logits": t(f"aux{i}_pred_logits"), "pred_masks": t(f"aux{i}_pred_masks")}
                            for i in range(3)],
            "output_mask": (torch.arange(T)[None, :] < torch.tensor(lens)[:, None])[:, None, :]}
    gp, gm, gs = synth_relations(lens, T, mc["num_classes"], seed=want["seed"])
    return mc, pred, (gp, gm, gs if mc.get("with_fuzzy", False) else None), want


@pytest.mark.parametrize("name,T", CRIT_CASES)
def test_criterion_oracle_matches_reference(name, T):
    mc, pred, (gp, gm, gs), want = stored_predictions(name, T)
    losses, idx = O.criterion(mc, pred, gp, gm, gs)
    assert [[list(map(int, r)), list(map(int, c))] for r, c in idx] == want["indices"]
    assert set(losses) == set(want["losses"])
    for k, v in want["losses"].items():
        assert abs(losses[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, losses[k], v)


@pytest.mark.parametrize("name,T", CRIT_CASES)
def test_criterion_host_code_matches_reference(name, T):
    """vrdone_amd's matcher + losses (device-agnostic tensor code, block-diagonal costs) on the same stored
    predictions, run on the CPU: same matches, same loss values, and each term equals the oracle's."""
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, pred, (gp, gm, gs), want = stored_predictions(name, T)
    model = MaskVRD(mc, device="cpu")
    data = {"preds_list": gp, "masks_list": gm}
    if gs is not None:
        data["segs_list"] = gs
    got = model.criterion(pred, data)
    idx, lmask = model.bipartite_match(pred["pred_logits"], gp, pred["pred_masks"], gm, gs, _mask=pred["output_mask"])
    assert [[i.tolist(), j.tolist()] for i, j in idx] == want["indices"]
    assert lmask.shape == (sum(len(p) for p in gp), T) and lmask.dtype == torch.bool
    assert list(got) == list(want["losses"])          # same keys in the same order, total_loss last
    for k, v in want["losses"].items():
        assert abs(float(got[k]) - v) <= 2e-5 * max(1.0, abs(v)), (k, float(got[k]), v)


def test_reference_named_loss_functions_agree_with_the_oracle():
    """models/losses.py's public names: the all-pairs cost matrices' diagonal blocks equal the oracle's
    per-(query, relation) costs, hard and fuzzy."""
    from oracle.synth import synth_relations
    from vrdone_amd.models import losses as L
    g = torch.Generator().manual_seed(5)
    B, Q, T = 3, 4, 40
    lens = [40, 17, 9]
    x = torch.randn(B, Q, T, generator=g) * 3
    logits = torch.randn(B, Q, 6, generator=g)
    valid = torch.arange(T)[None, :] < torch.tensor(lens)[:, None]
    gp, gm, gs = synth_relations(lens, T, 5, seed=11)
    owner = torch.repeat_interleave(torch.arange(B), torch.tensor([len(p) for p in gp]))
    flat, om = x.flatten(0, 1), valid[:, None, :].expand(B, Q, T).flatten(0, 1)
    tm, tgt, segs = valid[owner], torch.cat(gm), torch.cat(gs)
    for fuzzy in (False, True):
        cfg = {"with_fuzzy": fuzzy, "scale_range": 0.85, "cost_coeff_dict": {"cost_class": 0.0, "cost_mask": 1.0,
                                                                              "cost_dice": 0.0}}
        _, cm = O.bipartite_match(cfg, logits, x, valid[:, None, :], gp, gm, gs)
        cfg["cost_coeff_dict"] = {"cost_class": 0.0, "cost_mask": 0.0, "cost_dice": 1.0}
        _, cd = O.bipartite_match(cfg, logits, x, valid[:, None, :], gp, gm, gs)
        if fuzzy:
            fm = L.batch_masked_sigmoid_focal_fuzzy_loss(flat, tgt, om, tm, segs, scale_range=0.85)
            dm = L.batch_masked_dice_fuzzy_loss(flat, tgt, om, tm, segs, scale_range=0.85)
        else:
            fm = L.batch_masked_sigmoid_focal_loss(flat, tgt, om, tm)
            dm = L.batch_masked_dice_loss(flat, tgt, om, tm)
        assert fm.shape == dm.shape == (B * Q, len(tgt))
        g0 = 0
        for b in range(B):
            n = len(gp[b])
            np.testing.assert_allclose(fm[b * Q:(b + 1) * Q, g0:g0 + n].numpy(), cm[b], rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(dm[b * Q:(b + 1) * Q, g0:g0 + n].numpy(), cd[b], rtol=2e-5, atol=1e-6)
            g0 += n


def test_training_batch_loss_values_match_reference(weights):
    """BASELINE config 3 shape (24 pairs, T_pad 96, ragged): oracle network -> oracle matcher + losses against the
    reference's forward_training values (network in eval mode, PYTORCH_JIT=0)."""
    from oracle.synth import synth_relations
    mc, _, sd = weights("vidvrd")
    with open(os.path.join(GOLDEN, "criterion_vidvrd.json")) as f:
        want = json.load(f)["train24"]
    lens = want["lengths"]
    x, m = O.synth_pairs(len(lens), c_in(mc), 96, lens, seed=3)
    pred = O.mask_vrd(sd, mc, x, m, with_aux=True)
    gp, gm, gs = synth_relations(lens, 96, mc["num_classes"], max_rel=4, seed=want["seed"])
    losses, idx = O.criterion(mc, pred, gp, gm, gs)
    _, costs = O.bipartite_match(mc, pred["pred_logits"], pred["pred_masks"], pred["output_mask"], gp, gm, gs)
    for n, (r, c), w, C in zip(lens, idx, want["indices"], costs):
        # pairs of < 16 frames have one valid frame at the predictor's T/8 level: their queries cost the same to
        # ~1e-5 and the optimum is a near-tie; the reference's assignment must then be as cheap as ours
        assert C[w[0], w[1]].sum() - C[r, c].sum() <= 1e-4
        if n >= 16:
            assert [list(map(int, r)), list(map(int, c))] == w
    for k, v in want["losses"].items():
        assert abs(losses[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, losses[k], v)


def test_criterion_with_a_pair_without_relations():
    """A pair whose relations were all dropped (empty preds / masks / segs) takes part in the class loss only; the host
    code and the oracle agree (the reference's dataloader never emits one, its matcher handles it the same way)."""
    from oracle.synth import synth_relations
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, _ = load_case("vidvrd")
    model = MaskVRD(mc, device="cpu")
    model.deep_supervision = False
    g = torch.Generator().manual_seed(0)
    lens = [96, 50, 20, 7]
    pred = {"pred_logits": torch.randn(4, 9, 133, generator=g), "pred_masks": torch.randn(4, 9, 96, generator=g) * 3,
            "output_mask": (torch.arange(96)[None] < torch.tensor(lens)[:, None])[:, None]}
    gp, gm, gs = synth_relations(lens, 96, 132, seed=1)
    gp[2], gm[2], gs[2] = gp[2][:0], gm[2][:0], gs[2][:0]
    got = model.criterion(pred, {"preds_list": gp, "masks_list": gm, "segs_list": gs})
    want, idx = O.criterion(mc, pred, gp, gm, gs)
    assert len(idx[2][0]) == 0
    for k, v in want.items():
        assert abs(float(got[k]) - float(v)) <= 2e-5 * max(1.0, abs(float(v))), k


# ---------------------------------------------------------------------------------------------------
# eval-time pair construction: oracle/proposal.py vs the reference dataloader's _test_getitem
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["vidvrd", "strided"])
def test_pair_construction_matches_reference_dataloader(name):
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    g = np.load(os.path.join(GOLDEN, "proposal.npz"))
    vid_kw, dl_kw = PROPOSAL_CASES[name]
    out = P.test_getitem(P.synth_raw_video(**vid_kw), **dl_kw)
    assert out["sids"].tolist() == g[f"{name}/sids"].tolist() and out["oids"].tolist() == g[f"{name}/oids"].tolist()
    assert out["so_offset"].tolist() == g[f"{name}/so_offset"].tolist()
    feats = out["so_features_list"]
    assert [f.shape[1] for f in feats] == g[f"{name}/lens"].tolist()
    np.testing.assert_array_equal(torch.cat(out["bboxes_list"], dim=0).numpy(), g[f"{name}/boxes_clamped"])
    np.testing.assert_array_equal(torch.cat([f[-21:].T for f in feats], dim=0).numpy(), g[f"{name}/box_feats"])   # same torch ops: bit-equal
    sums = np.asarray([[float(f[:-21].double().sum()), float(f[:-21].double().abs().sum())] for f in feats])
    np.testing.assert_array_equal(sums, g[f"{name}/vis_sums"])


ABS_PE_CASES = [("vidvrd", 3, 96, [96, 50, 7]), ("vidvrd", 3, 288, [288, 201, 30]), ("vidor_x", 2, 128, [128, 77])]


@pytest.mark.parametrize("name,B,T,lens", ABS_PE_CASES)
def test_absolute_position_encoding_matches_reference(name, B, T, lens, weights):
    """`use_abs_pe: True` (scripts/make_golden_r2.py --only-abs-pe): the sinusoid table behind the visual embedding, as is
    below max_len, linearly re-interpolated from it on (vidvrd T = 96 and 288), CLIP variant behind the visual/CLIP fusion."""
    mc, _, sd = weights(name)
    mc = dict(mc, use_abs_pe=True)
    g = np.load(os.path.join(GOLDEN, "abs_pe.npz"))
    x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=4321 + T)
    out = O.mask_vrd(sd, mc, x, m, with_aux=False)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g[f"{name}/T{T}_pred_logits"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(out["pred_masks"].numpy(), g[f"{name}/T{T}_pred_masks"], atol=MASK_TOL, rtol=0)
    # and the encoding matters: without it the outputs are somewhere else
    off = O.mask_vrd(sd, dict(mc, use_abs_pe=False), x, m, with_aux=False)
    assert float((off["pred_logits"] - out["pred_logits"]).abs().max()) > 100 * LOGIT_TOL


@pytest.mark.parametrize("name,B,T,lens", [("vidvrd", 3, 96, [96, 50, 7]), ("vidor_local", 2, 128, [128, 77])])
def test_relative_position_encoding_matches_reference(name, B, T, lens, weights):
    """`use_rel_pe: True` (scripts/make_golden_r2.py --only-rel-pe): one bias per (head, window slot) on the scores of every
    stem / branch block's banded attention; the extra parameters are listed in rel_pe_keys.json."""
    mc, _, sd = weights(name)
    with open(os.path.join(GOLDEN, "rel_pe_keys.json")) as f:
        extra = json.load(f)[name]
    assert len(extra) == mc["backbone_arch"][1] + mc["backbone_arch"][2]
    sd = dict(sd, **O.synth_state_dict(extra))
    mc = dict(mc, use_rel_pe=True)
    g = np.load(os.path.join(GOLDEN, "rel_pe.npz"))
    x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=8765 + T)
    out = O.mask_vrd(sd, mc, x, m, with_aux=False)
    np.testing.assert_allclose(out["pred_logits"].numpy(), g[f"{name}/T{T}_pred_logits"], atol=LOGIT_TOL, rtol=0)
    np.testing.assert_allclose(out["pred_masks"].numpy(), g[f"{name}/T{T}_pred_masks"], atol=MASK_TOL, rtol=0)
    off = O.mask_vrd({k: v for k, v in sd.items() if not k.endswith("rel_pe")}, mc, x, m, with_aux=False)
    assert float((off["pred_logits"] - out["pred_logits"]).abs().max()) > 100 * LOGIT_TOL


def test_result_is_independent_of_the_padded_length_above_the_tight_one():
    """What MaskVRD's tight padding rests on (vrdone_amd/models/maskvrd.py): in exact arithmetic a pair's outputs depend on its
    padded length T only through "is there a padded frame behind the last valid one at every pyramid level" -- T / 8 >
    ceil(L / 8) for the shipped three-level pyramid.  The oracle in float64: any T at or above 8 * (ceil(L / 8) + 1) gives the
    outputs of the reference's own padded length to 1e-12; one step of 8 below it does not (the FPN's top level then ends on a
    valid frame: zero padding instead of LayerNorm(0) = beta behind it)."""
    mc, _, keys = load_case("vidvrd")
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]).items()}
    c_in = 2 * mc["visual_dim"] + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]
    for L, t_ref in ((41, 96), (100, 288), (7, 96)):
        feat = torch.randn(1, c_in, L, generator=torch.Generator().manual_seed(L), dtype=torch.float64)

        def run(T):
            x = torch.zeros(1, c_in, T, dtype=torch.float64)
            x[..., :L] = feat
            o = O.mask_vrd(sd, mc, x, (torch.arange(T) < L)[None, None], with_aux=False)
            return o["pred_logits"], o["pred_masks"][..., :L]
        want = run(t_ref)
        tight = 8 * (-(-L // 8) + 1)
        for T in (tight, tight + 8):
            got = run(T)
            assert float((got[0] - want[0]).abs().max()) < 1e-12 and float((got[1] - want[1]).abs().max()) < 1e-12, (L, T)
        if L > 8:
            below = run(tight - 8)
            assert float((below[1] - want[1]).abs().max()) > 1e-3, (L, tight - 8)
