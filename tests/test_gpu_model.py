"""End-to-end parity of the HIP path on a real MI355X: MaskVRD._mask_vrd and forward_test
against (a) outputs of the real reference stored in tests/golden/ and (b) the CPU oracle on
fresh inputs, plus size-independent properties at the benchmark shape.

Stated tolerance (BASELINE north star): predicate logits within 1e-3 of the reference.  Measured
max differences vs the reference: f32 mode ~6e-6 (logits) / 6e-5 (mask logits, range +-25); bf16x3 mode
(the default: split-bf16 MFMA products, f32 accumulate) ~7e-5 / 5e-4; f16x3 mode (scaled split-f16 products: the
reference-grade mode) like f32.  The asserts use 2e-4 / 2e-3 for all modes; the two reference-grade modes are in
addition held to 2 x the reference's own float32 error against a float64 run of the reference
(test_reference_grade_against_float64)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from oracle import vrd_oracle as O
from oracle.synth import synth_proposal

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
LOGIT_TOL, MASK_TOL = 2e-4, 2e-3


def c_in(mc):
    cc = mc["clip_dim"] if mc.get("with_clip_feature", False) else 0
    return 2 * mc["visual_dim"] + 2 * cc + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]


_models = {}


def get_model(name):
    if name not in _models:
        from vrdone_amd.models.maskvrd import MaskVRD
        mc, ic, keys = load_case(name)
        sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
        model = MaskVRD(mc, device=DEV)
        model.load_state_dict(sd, strict=True)
        model = model.to(DEV).eval()
        model._config_eval(ic)
        _models[name] = (model, mc, ic, sd)
    return _models[name]


def close(got, want, atol):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    assert got.shape == want.shape and np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=atol, rtol=0)


@pytest.fixture(params=["bf16x3", "f16x3", "f32"])
def precision(request):
    from vrdone_amd import ops
    old = ops.get_precision()
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(old)


@pytest.mark.parametrize("name,T", [("vidvrd", 96), ("vidvrd", 144), ("vidvrd", 288),
                                    ("vidor_x", 512), ("vidor_local", 512), ("vidor", 512)])
def test_mask_vrd_matches_reference_golden(name, T, precision):
    model, mc, _, _ = get_model(name)
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    lens = g[f"T{T}_lengths"].tolist()
    x, m = O.synth_pairs(len(lens), c_in(mc), T, lens, seed=1234 + T)
    xd, md = x.to(DEV), m.to(DEV)
    if f"T{T}_feat0" in g:          # intermediates through the reference-signature module calls
        feats, masks = model.backbone(xd, md)
        for l, ft in enumerate(feats):
            close(ft[:, ::16], g[f"T{T}_feat{l}"], 2e-4)
        fpn, _ = model.neck(feats, masks)
        close(fpn[:, ::8], g[f"T{T}_fpn"], 2e-4)
    out = model._mask_vrd(xd, md)
    close(out["pred_logits"], g[f"T{T}_pred_logits"], LOGIT_TOL)
    close(out["pred_masks"], g[f"T{T}_pred_masks"], MASK_TOL)
    assert torch.equal(out["output_mask"].cpu(), m)
    assert len(out["aux_outputs"]) == 3
    for i, a in enumerate(out["aux_outputs"]):
        close(a["pred_logits"], g[f"T{T}_aux{i}_pred_logits"], LOGIT_TOL)
        close(a["pred_masks"], g[f"T{T}_aux{i}_pred_masks"], MASK_TOL)
    # eval shortcut: last layer only, identical numbers
    fast = model._mask_vrd(xd, md, with_aux=False)
    assert "aux_outputs" not in fast
    assert torch.equal(fast["pred_logits"], out["pred_logits"]) and torch.equal(fast["pred_masks"], out["pred_masks"])


F64_CASES = [("vidvrd", 96), ("vidvrd", 144), ("vidvrd", 288), ("vidor_x", 512), ("vidor_local", 512), ("vidor", 512),
             ("b256", None), ("cfg2", None)]


def f64_case(name, T):
    """(model, x, m, ref32 outputs, ref64 outputs, row stride of the stored pairs) of a case of tests/golden/mask_vrd_f64.npz
    (scripts/make_golden_f64.py: the real reference run in float64 on the goldens' inputs)."""
    from golden_cases import B256, CFG2, b256_lengths, cfg2_lengths
    f64 = np.load(os.path.join(GOLDEN, "mask_vrd_f64.npz"))
    if T is None:
        spec, lens = (B256, b256_lengths()) if name == "b256" else (CFG2, cfg2_lengths())
        model, mc, _, _ = get_model("vidvrd")
        g = np.load(os.path.join(GOLDEN, f"mask_vrd_vidvrd_{name}.npz"))
        x, m = O.synth_pairs(spec["B"], c_in(mc), spec["T"], lens, seed=spec["seed"])
        return (model, x, m, {k: g[k] for k in ("pred_logits", "pred_masks")},
                {k: f64[f"{name}_{k}"] for k in ("pred_logits", "pred_masks")}, spec["every"])
    model, mc, _, _ = get_model(name)
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    lens = g[f"T{T}_lengths"].tolist()
    x, m = O.synth_pairs(len(lens), c_in(mc), T, lens, seed=1234 + T)
    return (model, x, m, {k: g[f"T{T}_{k}"] for k in ("pred_logits", "pred_masks")},
            {k: f64[f"{name}_T{T}_{k}"] for k in ("pred_logits", "pred_masks")}, 1)


def f64_distance(name, T):
    """{output: (max, rms) of |ours - ref64| / the same of |ref32 - ref64|} for the current precision mode"""
    model, x, m, ref32, ref64, every = f64_case(name, T)
    out = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    res = {}
    for k in ("pred_logits", "pred_masks"):
        ours = out[k][::every].double().cpu().numpy()
        assert np.isfinite(ours).all()
        e, e32 = np.abs(ours - ref64[k]), np.abs(ref32[k].astype(np.float64) - ref64[k])
        res[k] = (float(e.max() / e32.max()), float(np.sqrt((e ** 2).mean() / (e32 ** 2).mean())), float(e.max()), float(e32.max()))
    return res


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
@pytest.mark.parametrize("name,T", F64_CASES)
def test_reference_grade_against_float64(name, T, mode):
    """The two modes that may be quoted as the reference's arithmetic: their distance to a FLOAT64 run of the real reference
    (tests/golden/mask_vrd_f64.npz) is within 2 x the distance of the reference's own float32 run (the committed float32
    goldens) on the same inputs -- worst element and rms, logits and mask logits, every golden case of `_mask_vrd`
    (all four shipped configs, the 256-pair batch, BASELINE config 2 at its size).  Measured (profiles/r04_f64_distance.txt):
    f16x3 x0.56 - x1.36 worst element, x0.87 - x1.11 rms; f32 x0.92 - x2.88 / x1.09 - x1.79; the bf16x3 mode sits at ~x17 and
    is not held to this."""
    from vrdone_amd import ops
    old = ops.get_precision()
    ops.set_precision(mode)
    try:
        res = f64_distance(name, T)
    finally:
        ops.set_precision(old)
    # f32 (exact products, but ONE accumulation chain along K per output in the f32 MFMA: its rounding errors add up where
    # the 16-bit MFMA sums 16 products per instruction) measures x0.9 - x2.9 on the worst element: held to 4 x, rms to 2 x
    bound_max = 2.0 if mode == "f16x3" else 4.0
    for k, (r_max, r_rms, e_max, e32_max) in res.items():
        assert r_max <= bound_max and r_rms <= 2.0, \
            f"{name} T{T} {mode} {k}: max {e_max:.3e} vs ref32 {e32_max:.3e} (x{r_max:.2f}), rms x{r_rms:.2f}"


def test_tight_padding_gives_the_same_outputs(precision):
    """MaskVRD's tight padding (pairs computed at the shortest padded length that leaves a padded frame behind them at every
    pyramid level, bucket by bucket) against the same batch computed at the batch's own padded length: the same numbers up to
    the rounding of differently shaped launches, in the reference's layout (mask logits -10 behind every pair's frames)."""
    model, mc, _, _ = get_model("vidvrd")
    gen = torch.Generator().manual_seed(77)
    B, T = 300, 288
    lens = torch.randint(2, 257, (B,), generator=gen)
    lens[:6] = torch.tensor([288, 287, 281, 280, 256, 2])           # pairs that cannot shrink, and the shortest one
    x, m = O.synth_pairs(B, c_in(mc), T, lens.tolist(), seed=78)
    xd, md = x.to(DEV), m.to(DEV)
    assert model.tight_padding
    try:
        model.TIGHT_MIN_ROWS = 16384           # (buckets this small only here: the default merges a 300-pair batch into one)
        model.row_space = False                # (bucket by bucket here; the row-space form has its own test below)
        plan = model._tight_plan(md, md.reshape(B, T))
        assert plan is not None and not plan["rows"]
        plan = plan["buckets"]
        assert len(plan) >= 3 and sum(n for _, _, n, _ in plan) == B and max(t for t, _, _, _ in plan) == T
        for t2, idx, n, _ in plan:
            assert all(model.tight_len(int(lens[i]), T) <= t2 for i in idx.tolist())
        tight = model._mask_vrd(xd, md)
        model.tight_padding = False
        full = model._mask_vrd(xd, md.clone())
    finally:
        model.tight_padding = True
        del model.TIGHT_MIN_ROWS, model.row_space
    tol = 2e-4 if precision == "bf16x3" else 2e-5
    close(tight["pred_logits"], full["pred_logits"], tol)
    close(tight["pred_masks"], full["pred_masks"], 10 * tol)
    assert torch.equal(tight["output_mask"], full["output_mask"]) and len(tight["aux_outputs"]) == len(full["aux_outputs"]) == 3
    for a, b in zip(tight["aux_outputs"], full["aux_outputs"]):
        close(a["pred_logits"], b["pred_logits"], tol)
        close(a["pred_masks"], b["pred_masks"], 10 * tol)
    pad = ~m[:, 0][:, None, :].expand(-1, tight["pred_masks"].shape[1], -1)
    assert bool((tight["pred_masks"].cpu()[pad] == -10.0).all())


@pytest.mark.parametrize("name,B,T,chunk", [("vidvrd", 300, 288, None), ("vidvrd", 130, 96, 48), ("vidor", 96, 512, None),
                                            ("vidor_x", 64, 512, None), ("vidor_local", 64, 512, 40)])
def test_row_space_form_gives_the_same_outputs(name, B, T, chunk, precision):
    """A ragged batch with all its tight-padding buckets in ONE row space (models/ragged.py: LayerNorm and every dense conv GEMM
    once over all rows -- the k = 3 embedding convs with every sequence's last, padded frame zeroed in their input --, the
    kernels that need (sequences, frames) structure bucket by bucket) against the same batch computed at the batch's own padded
    length: the same numbers up to the rounding of differently shaped launches, in the reference's layout; all four shipped
    configs (banded and global SOS attention, CLIP slabs, 4 x 128 and 8 x 64 heads), also in waves (pair_chunk)."""
    model, mc, _, _ = get_model(name)
    gen = torch.Generator().manual_seed(177)
    lens = torch.randint(2, T - 30, (B,), generator=gen)
    lens[:7] = torch.tensor([T - 2, T - 8, T - 9, 2, 33, T, T - 1])    # (T, T - 1: no two padded frames behind them -- a bucket apart)
    x, m = O.synth_pairs(B, c_in(mc), T, lens.tolist(), seed=178)
    xd, md = x.to(DEV), m.to(DEV)
    old_chunk = model.pair_chunk
    try:
        model.ROWS_MIN_ROWS = 2048             # (buckets this small only here)
        if chunk:
            model.pair_chunk = chunk
        plan = model._tight_plan(md, md.reshape(B, T))
        assert plan is not None and plan["rows"] and len(plan["buckets"]) >= 3
        for t2, idx, n, flat in plan["buckets"]:
            assert all((int(lens[i]) <= t2 - 2) == flat for i in idx.tolist())
        assert [f for _, _, _, f in plan["buckets"]].count(False) == 1
        rows = model._mask_vrd(xd, md)
        model.tight_padding = False
        full = model._mask_vrd(xd, md.clone())
    finally:
        model.tight_padding = True
        model.pair_chunk = old_chunk
        del model.ROWS_MIN_ROWS
    tol = 2e-4 if precision == "bf16x3" else 2e-5
    close(rows["pred_logits"], full["pred_logits"], tol)
    close(rows["pred_masks"], full["pred_masks"], 10 * tol)
    assert torch.equal(rows["output_mask"], full["output_mask"]) and len(rows["aux_outputs"]) == len(full["aux_outputs"]) == 3
    for a, b in zip(rows["aux_outputs"], full["aux_outputs"]):
        close(a["pred_logits"], b["pred_logits"], tol)
        close(a["pred_masks"], b["pred_masks"], 10 * tol)
    pad = ~m[:, 0][:, None, :].expand(-1, rows["pred_masks"].shape[1], -1)
    assert bool((rows["pred_masks"].cpu()[pad] == -10.0).all())


def test_row_space_filler_sequences_change_nothing():
    """From ~65 k rows on, the row-space form appends a few all-padding sequences so that the row count is a multiple of 256
    (models/ragged.py filler_buckets: the 256 x 256 GEMM kernel then serves the first three pyramid levels).  A 700-pair ragged
    batch, large enough for the filler: same outputs as the batch at its own padded length, and as bucket by bucket."""
    from vrdone_amd.models import ragged
    model, mc, _, _ = get_model("vidvrd")
    gen = torch.Generator().manual_seed(277)
    B, T = 700, 288
    lens = torch.randint(2, T - 30, (B,), generator=gen)
    lens[:3] = torch.tensor([T, T - 1, T - 2])
    for k in range(9):                       # (a row count that is a multiple of 256 by itself needs none: move one pair up a bucket)
        rows = sum(model.tight_buckets(lens.tolist(), [T] * B, model.ROWS_MIN_ROWS))
        if rows % 256:
            break
        lens[10 + k] = 100 + 32 * (k % 4)
    x, m = O.synth_pairs(B, c_in(mc), T, lens.tolist(), seed=278)
    xd, md = x.to(DEV), m.to(DEV)
    plan = model._tight_plan(md, md.reshape(B, T))
    rows = sum(n * t for t, _, n, _ in plan["buckets"])
    assert plan["rows"] and rows >= 65536 and ragged.filler_buckets(rows, T), "the batch was meant to need filler sequences"
    try:
        got = model._mask_vrd(xd, md, with_aux=False)
        model.row_space = False
        buckets = model._mask_vrd(xd, md.clone(), with_aux=False)
        model.tight_padding = False
        full = model._mask_vrd(xd, md.clone(), with_aux=False)
    finally:
        model.tight_padding = True
        del model.row_space
    close(got["pred_logits"], full["pred_logits"], 2e-5)
    close(got["pred_masks"], full["pred_masks"], 2e-4)
    close(got["pred_logits"], buckets["pred_logits"], 2e-5)
    close(got["pred_masks"], buckets["pred_masks"], 2e-4)


def test_forward_test_graph_replay_equals_eager():
    """Small videos (the sizes real vidvrd data has) run bucket by bucket, each bucket's device side as a recorded HIP graph
    (vrdone_amd/eval_graph.py): the replayed call returns exactly what the eager call returns -- on the recording call, on
    replays, on another video that shares the recordings (other pair counts under the same padded sizes), and after the
    weights were changed in place (derived operands are rebuilt inside the graph)."""
    from vrdone_amd import eval_graph, synth
    model, mc, ic, _ = get_model("vidvrd")
    eval_graph.forget(model)
    videos = [synth.synth_video(n_trk, c_in(mc), lo, hi, seed=seed, device=DEV)
              for n_trk, lo, hi, seed in ((8, 10, 60, 3), (7, 12, 70, 5), (14, 20, 90, 7))]

    def run(data, graphs):
        old = eval_graph.ENABLED
        eval_graph.ENABLED = graphs
        try:
            return model(data)
        finally:
            eval_graph.ENABLED = old

    def same(a, b):
        assert len(a["triplets"]) == len(b["triplets"]) > 10
        for key in ("triplets", "pred_durations", "so_tids", "so_trajs", "triple_scores", "triple_scores_avg"):
            assert a[key] == b[key], key

    weight = model.backbone.stem[0].mlp[0].weight if hasattr(model.backbone.stem[0], "mlp") else next(model.parameters())
    try:
        for data in videos:
            want = run(data, False)
            same(run(data, True), want)            # records (first video) or replays
            same(run(data, True), want)            # replays
        n_rec = len(eval_graph.recordings(model))
        assert 0 < n_rec <= eval_graph.MAX_RECORDINGS
        with torch.no_grad():
            weight.mul_(1.03125)
        want = run(videos[0], False)
        same(run(videos[0], True), want)
        assert len(eval_graph.recordings(model)) == n_rec          # no new recording: the old ones saw the new weights
    finally:
        with torch.no_grad():
            weight.div_(1.03125)
        eval_graph.forget(model)


def test_forward_test_in_one_row_space_equals_bucket_by_bucket():
    """forward_test on a video large enough for waves and filler sequences (1,260 pairs of 60-250 frames, pair_chunk 512): all
    padded lengths of a wave in one row space against the bucket-by-bucket path -- the same triplets, tracks and durations, scores
    to 1e-5; from per-pair matrices and from per-tracklet features (window-edge pieces of all buckets in one entity-stage pass)."""
    from vrdone_amd import synth
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model("vidvrd")
    old_chunk = model.pair_chunk
    video = synth.synth_video(36, c_in(mc), 60, 250, seed=11, device=DEV)
    raw = synth.synth_raw_video(36, mc["visual_dim"], 60, 250, seed=11)
    prop = prepare_test_proposal(raw, ic["feat_stride"], 0, 2, torch.device(DEV))
    try:
        model.pair_chunk = 512
        for data in (video, prop):
            got = model(data)
            model.row_space = False
            want = model(data)
            del model.row_space
            assert len(got["triplets"]) == len(want["triplets"]) > 50
            for key in ("triplets", "pred_durations", "so_tids", "so_trajs"):
                assert got[key] == want[key], key
            for key in ("triple_scores", "triple_scores_avg"):
                np.testing.assert_allclose(np.array(got[key]), np.array(want[key]), atol=1e-5, rtol=0)
    finally:
        model.pair_chunk = old_chunk
        if "row_space" in model.__dict__:
            del model.row_space


@pytest.mark.parametrize("scale", [1.0 / 1024, 48.0])          # (each case is two CPU oracle runs, ~18-30 s: the two ends; scale 1 is every other test)
def test_f16x3_stays_reference_grade_across_input_magnitudes(scale):
    """The f16x3 mode's activation scale is fixed (2^4): inputs far below 1 push the lo halves of the FIRST layer's operands into
    f16 subnormals (2^-25 absolute instead of 2^-22 relative).  Against the oracle in float64 on inputs scaled by 1/1024 .. 48 the
    mode stays within 2 x the float32 oracle's own error (the first LayerNorm renormalises what follows); 48 x N(0, 1) reaches
    ~250, a sixteenth of the operand range."""
    from vrdone_amd import ops
    model, mc, _, sd = get_model("vidvrd")
    lens = [96, 95, 41, 2]
    x, m = O.synth_pairs(4, c_in(mc), 96, lens, seed=4321)
    x = x * scale
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    want = O.mask_vrd(sd64, mc, x.double(), m, with_aux=False)
    own32 = O.mask_vrd(sd, mc, x, m, with_aux=False)
    with ops.use_precision("f16x3"):
        got = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    for k in ("pred_logits", "pred_masks"):
        e = float((got[k].double().cpu() - want[k]).abs().max())
        e32 = float((own32[k].double() - want[k]).abs().max())
        print(f"scale {scale:g} {k}: |f16x3 - f64| {e:.2e}, |f32 oracle - f64| {e32:.2e} (x{e / e32:.2f})")
        assert e <= 2.0 * e32 + 1e-7, (scale, k, e, e32)


def test_f16x3_overflow_is_loud_and_forward_test_repeats_in_f32():
    """Inputs beyond the f16x3 mode's operand range (|x| >= 4094): the path returns NaN, never a wrong finite number, and
    forward_test repeats the video in the f32 mode -- whose result it then returns (the reference computes float32)."""
    import warnings
    from vrdone_amd import ops
    model, mc, _, _ = get_model("vidvrd")
    data = synth_proposal(4, c_in(mc), 20, 60, seed=99)
    big = dict(data, so_features_list=[f * 3.0e4 for f in data["so_features_list"]])
    dev_data = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in big.items()}
    inputs, masks, _ = model.preprocessing(dev_data["so_features_list"])
    with ops.use_precision("f16x3"):
        ops.f16_range_flag().zero_()
        out = model._mask_vrd(inputs[0], masks[0], with_aux=False)
        assert not bool(torch.isfinite(out["pred_logits"]).all())          # poisoned, not silently wrong
        assert ops.f16_range_exceeded() & 1                                 # ... and reported by the kernel that split the inputs
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = model(dev_data)
        assert any("f16x3" in str(x.message) for x in w)
    with ops.use_precision("f32"):
        want = model(dev_data)
    assert got is not None and got["triplets"] == want["triplets"] and got["triple_scores_avg"] == want["triple_scores_avg"]


def _model_with(sd_edit):
    """a vidvrd model whose synthetic state dict went through `sd_edit(sd)` first (+ that state dict)"""
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, ic, keys = load_case("vidvrd")
    sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    sd = {k: v.clone() for k, v in sd.items()}
    sd_edit(sd)
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    model._config_eval(ic)
    return model, mc, sd


def test_f16x3_heavy_tailed_weights_stay_in_range_and_reference_grade():
    """Trained-like weight statistics instead of the synthetic state dict's even ones: LayerNorm gains up to 10 on a few
    channels and a few rows x 30 in every block's MLP up-projection, so that INTERNAL tensors (MLP hiddens, residual streams,
    the SOS streams) run at hundreds instead of ones -- a sizeable part of the f16 planes' +-4094.  The f16x3 mode stays within
    2 x the float32 oracle's own distance to float64, and no producer reports its range exceeded."""
    from vrdone_amd import ops

    def edit(sd):
        gen = torch.Generator().manual_seed(11)
        for k, v in sd.items():
            if (k.endswith("ln1.weight") or k.endswith("ln2.weight") or k.endswith("_norm.weight")) and v.dim() == 3 and v.shape[1] >= 256:
                idx = torch.randperm(v.shape[1], generator=gen)[:max(2, v.shape[1] // 64)]
                v[:, idx] *= 10.0
            if k.endswith("mlp.0.weight"):
                v[torch.randperm(v.shape[0], generator=gen)[:4]] *= 30.0
    model, mc, sd = _model_with(edit)
    x, m = O.synth_pairs(4, c_in(mc), 96, [96, 95, 41, 2], seed=4322)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    want = O.mask_vrd(sd64, mc, x.double(), m, with_aux=False)
    own32 = O.mask_vrd(sd, mc, x, m, with_aux=False)
    with ops.use_precision("f16x3"):
        ops.f16_range_flag().zero_()
        got = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
        assert ops.f16_range_exceeded() == 0
    for k in ("pred_logits", "pred_masks"):
        e = float((got[k].double().cpu() - want[k]).abs().max())
        e32 = float((own32[k].double() - want[k]).abs().max())
        print(f"{k}: |f16x3 - f64| {e:.2e}, |f32 oracle - f64| {e32:.2e} (x{e / e32:.2f})")
        assert e <= 2.0 * e32 + 1e-7, (k, e, e32)


def test_f16x3_internal_overflow_is_reported_and_repeated_in_f32(precision):
    """An activation INSIDE the network beyond the f16 planes' range, with ordinary inputs: the first block's MLP
    up-projection scaled by 4000 (its down-projection by 1 / 4000, so the float32 network computes what it computed
    before).  The hidden tensor is a pair-row GEMM output: the kernel that writes it reports (flag bit 8), forward_test
    repeats the video and forward_training / forward_loss the step in the f32 mode and return ITS results; the other modes
    are untouched."""
    import warnings
    from vrdone_amd import ops
    from vrdone_amd.models.blocks import AffineDropPath

    def edit(sd):
        sd["backbone.stem.0.mlp.0.weight"] *= 4000.0
        sd["backbone.stem.0.mlp.0.bias"] *= 4000.0
        sd["backbone.stem.0.mlp.3.weight"] /= 4000.0
    model, mc, _ = _model_with(edit)
    data = synth_proposal(4, c_in(mc), 20, 60, seed=98)
    dev_data = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in data.items()}
    with ops.use_precision("f32"):
        want = model(dev_data)
    inputs, masks, _ = model.preprocessing(dev_data["so_features_list"])
    ops.f16_range_flag().zero_()
    model._mask_vrd(inputs[0], masks[0], with_aux=False)
    bits = ops.f16_range_exceeded()
    assert (bits & 8) != 0 if precision == "f16x3" else bits == 0, (precision, bits)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = model(dev_data)
    assert any("f16x3" in str(x.message) for x in w) == (precision == "f16x3")
    if precision == "f16x3":
        assert got is not None and got["triplets"] == want["triplets"] and got["triple_scores_avg"] == want["triple_scores_avg"]
    # a training step and a validation pass
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_step", os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "train_step.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    from vrdone_amd import configs
    cfg = configs.model_config("vidvrd")
    batch = ts.synthetic_batch(cfg, configs.input_channels(cfg), DEV, n_pairs=6, seed=5)
    model.train()
    try:
        for mod in model.modules():
            if isinstance(mod, AffineDropPath):
                mod.drop_prob = 0.0
        with torch.enable_grad():
            with ops.use_precision("f32"):
                model.zero_grad(set_to_none=True)
                want_loss = model(batch)["total_loss"]
                want_loss.backward()
                want_grad = model.backbone.stem[0].mlp[3].weight.grad.clone()
            model.zero_grad(set_to_none=True)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                loss = model(batch)["total_loss"]
            loss.backward()
            grad = model.backbone.stem[0].mlp[3].weight.grad
        assert bool(torch.isfinite(loss)) and bool(torch.isfinite(grad).all())
        if precision == "f16x3":
            assert any("f32 mode" in str(x.message) for x in w)
            assert float((loss - want_loss).abs()) <= 1e-5 * float(want_loss.abs())
            assert float((grad - want_grad).abs().max()) <= 1e-4 * float(want_grad.abs().max())
        with torch.no_grad():
            val = model(batch)["total_loss"]
        assert bool(torch.isfinite(val))
        if precision == "f16x3":
            assert float((val - want_loss).abs()) <= 1e-5 * float(want_loss.abs())
    finally:
        model.eval()
        model.zero_grad(set_to_none=True)


def test_mask_vrd_matches_oracle_on_a_ragged_batch(precision):
    """BASELINE config 1 shape (64 pairs x 64 frames -> T_pad 96), ragged lengths, vs the oracle."""
    model, mc, _, sd = get_model("vidvrd")
    gen = torch.Generator().manual_seed(1235)
    lens = torch.randint(2, 65, (64,), generator=gen)
    lens[0], lens[1] = 96, 64
    x, m = O.synth_pairs(64, c_in(mc), 96, lens, seed=7)
    want = O.mask_vrd(sd, mc, x, m, with_aux=False)
    got = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    close(got["pred_logits"], want["pred_logits"], LOGIT_TOL)
    close(got["pred_masks"], want["pred_masks"], MASK_TOL)


def test_chunking_and_batch_independence():
    """Pairs are independent end to end: any chunking / batch composition gives the same rows."""
    model, mc, _, _ = get_model("vidvrd")
    x, m = O.synth_pairs(10, c_in(mc), 96, [96, 50, 7, 96, 33, 2, 64, 64, 95, 11], seed=21)
    xd, md = x.to(DEV), m.to(DEV)
    whole = model._mask_vrd(xd, md, with_aux=False)
    old = model.pair_chunk
    try:
        model.pair_chunk = 3
        parts = model._mask_vrd(xd, md, with_aux=False)
    finally:
        model.pair_chunk = old
    close(parts["pred_logits"], whole["pred_logits"], 1e-5)
    close(parts["pred_masks"], whole["pred_masks"], 1e-4)
    one = model._mask_vrd(xd[4:5], md[4:5], with_aux=False)
    close(one["pred_logits"], whole["pred_logits"][4:5], 1e-5)
    # padding invariance for L < T_pad: same pair padded to 144 instead of 96
    x2 = torch.zeros(1, x.shape[1], 144)
    x2[..., :96] = x[1:2]
    m2 = torch.zeros(1, 1, 144, dtype=torch.bool)
    m2[..., :50] = True
    pad = model._mask_vrd(x2.to(DEV), m2.to(DEV), with_aux=False)
    close(pad["pred_logits"], whole["pred_logits"][1:2], 5e-5)
    close(pad["pred_masks"][..., :50], whole["pred_masks"][1:2, :, :50], 5e-4)


BF16X3_TIE = {"bf16x3": 5e-6}       # forward_test rankings: ties of the reference's own scores the 17-bit mode may permute


def test_forward_test_matches_reference_golden(precision):
    model, mc, ic, _ = get_model("vidvrd")
    with open(os.path.join(GOLDEN, "forward_test_vidvrd.json")) as f:
        ref = json.load(f)
    data = synth_proposal(6, c_in(mc), 20, 130, seed=4321)
    dev_data = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in data.items()}
    res = model(dev_data)
    assert len(res["triplets"]) == len(ref["triplets"]) == ic["n_max_pair"]
    # identical ranking, records and box tracks in the reference-grade modes; scores within 5e-6 (measured: 1e-7 / 7e-7).  The
    # bf16x3 mode (17-bit products) may permute ranks whose reference scores lie closer together than its own score error
    from golden_cases import compare_forward_test
    compare_forward_test(res, ref, ic["n_max_pair"], 5e-6, slack=0, tie_tol=BF16X3_TIE.get(precision, 0.0))


def _on_device(data):
    return {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in data.items()}


def test_forward_test_many_slices_matches_reference_golden(precision):
    """494 pairs = three slices of the reference's max_so_pair loop (models/maskvrd.py:208-227) whose long pairs are
    padded to 192 / 240 / 240 frames: the length-bucketed batching of MaskVRD.forward_test against the REFERENCE."""
    from golden_cases import SLICES, compare_forward_test
    model, mc, ic, _ = get_model("vidvrd")
    with open(os.path.join(GOLDEN, "forward_test_vidvrd_slices.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **SLICES)
    assert len(data["sids"]) == ref["n_pairs"] > 2 * mc["max_so_pair"]
    res = model(_on_device(data))
    compare_forward_test(res, ref, ic["n_max_pair"], 5e-6, slack=0, tie_tol=BF16X3_TIE.get(precision, 0.0))


def test_forward_test_vidor_x_matches_reference_golden(precision):
    """vidor_x.yaml end to end (CLIP slabs, Q = 10, topk 6, feat_stride 4, so_offset in 0..3; 14 pairs at T 512 and 6
    at T_long 640) against the REFERENCE's forward_test."""
    from golden_cases import VIDOR_X, compare_forward_test
    model, mc, ic, _ = get_model("vidor_x")
    with open(os.path.join(GOLDEN, "forward_test_vidor_x.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **VIDOR_X)
    assert data["so_offset"].tolist() == ref["so_offset"]
    res = model(_on_device(data))
    compare_forward_test(res, ref, ic["n_max_pair"], 5e-6, slack=0, tie_tol=BF16X3_TIE.get(precision, 0.0))


@pytest.mark.parametrize("name", ["vidor", "vidor_local"])
def test_forward_test_vidor_variants_match_reference_golden(name, precision):
    """forward_test under vidor.yaml and vidor_local.yaml against the REFERENCE's results (identity of ranking, records
    and tracks; vidor_local: 180 candidates, fewer than n_max_pair)."""
    from golden_cases import FORWARD_TEST_VARIANTS, compare_forward_test
    model, mc, ic, _ = get_model(name)
    with open(os.path.join(GOLDEN, f"forward_test_{name}.json")) as f:
        ref = json.load(f)
    data = synth_proposal(c_in=c_in(mc), **FORWARD_TEST_VARIANTS[name])
    res = model(_on_device(data))
    compare_forward_test(res, ref, ic["n_max_pair"], 5e-6, slack=0, tie_tol=2e-6)


def test_mask_vrd_b256_matches_reference_golden(precision):
    """256 pairs x T_pad 288 with ragged lengths: the batch size at which the model selects the 256 x 256 LDS-DMA GEMM
    kernel, the padding maps and vrd_gemm_batch (ops.SKIP_MIN_ROWS rows), compared with the REFERENCE's outputs for
    every 16th pair."""
    from golden_cases import B256, b256_lengths
    from vrdone_amd import _hip, ops
    model, mc, _, _ = get_model("vidvrd")
    g = np.load(os.path.join(GOLDEN, "mask_vrd_vidvrd_b256.npz"))
    lens = b256_lengths()
    x, m = O.synth_pairs(B256["B"], c_in(mc), B256["T"], lens, seed=B256["seed"])
    assert 2 * B256["B"] * B256["T"] >= ops.SKIP_MIN_ROWS
    e = B256["every"]
    for tight in (False, True):        # at the batch's own padded length (padding maps), and in tight-padding buckets
        _hip.prof_enable(True)
        _hip.prof_reset()
        try:
            model.tight_padding = tight
            out = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
        finally:
            model.tight_padding = True
        torch.cuda.synchronize()
        prof = _hip.prof_read()
        _hip.prof_enable(False)
        close(out["pred_logits"][::e], g["pred_logits"], LOGIT_TOL)
        close(out["pred_masks"][::e], g["pred_masks"], MASK_TOL)
        if not tight and precision in ("bf16x3", "f16x3"):       # the kernel this case exists for did run, and skipped padded tiles
            assert prof["gemm_x3_big"]["launches"] > 0 and prof["gemm_x3_big"]["flops_skipped"] > 0


def test_mask_vrd_cfg2_at_size_matches_reference_golden(precision):
    """BASELINE config 2 at its size: 1024 pairs x 128 frames, which the eval batching pads to T_pad 144 -- not a multiple of
    32, so the last 32-row block of every pair straddles two sequences in the flat row space, the flash kernels see a
    partial last key tile and the banded kernel a ragged last strip.  Every 64th pair against the REFERENCE's output for
    the same batch; all pairs: finite, -10 on padded frames, independent of the batch composition (first 64 pairs alone)."""
    from golden_cases import CFG2, cfg2_lengths
    model, mc, _, _ = get_model("vidvrd")
    g = np.load(os.path.join(GOLDEN, "mask_vrd_vidvrd_cfg2.npz"))
    lens = cfg2_lengths()
    x, m = O.synth_pairs(CFG2["B"], c_in(mc), CFG2["T"], lens, seed=CFG2["seed"])
    x, m = x.to(DEV), m.to(DEV)
    out = model._mask_vrd(x, m, with_aux=False)
    e = CFG2["every"]
    close(out["pred_logits"][::e], g["pred_logits"], LOGIT_TOL)
    close(out["pred_masks"][::e], g["pred_masks"], MASK_TOL)
    assert bool(torch.isfinite(out["pred_logits"]).all()) and bool(torch.isfinite(out["pred_masks"]).all())
    assert bool((out["pred_masks"].transpose(1, 2)[~m[:, 0]] == -10.0).all())
    part = model._mask_vrd(x[:64].contiguous(), m[:64].contiguous(), with_aux=False)
    # (a 64-pair batch runs the small-tile GEMM kernels and the 32-query flash kernel: same products, other summation order)
    assert float((part["pred_logits"] - out["pred_logits"][:64]).abs().max()) <= 5e-5
    assert float((part["pred_masks"] - out["pred_masks"][:64]).abs().max()) <= 5e-4


def test_sharded_forward_test_with_real_processes():
    """Two fresh `torchrun` ranks (gloo, sharing the GPU) run scripts/sharded_eval_check.py: MaskVRD.shard_pairs() with the
    REAL all-gather on device tensors against the single-process forward_test, every field on every rank.  The ranks are
    child processes of this one (never an exec of a process that has touched the GPU)."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, BENCH_REHEARSAL="1", OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VRDONE_PRECISION", None)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(repo, "scripts", "sharded_eval_check.py")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world_size"] == 2 and line["equal_on_rank"] == [True, True] and line["triplets"] > 0


@pytest.mark.parametrize("name,B,T,lens", [("vidvrd", 3, 96, [96, 50, 7]), ("vidvrd", 3, 288, [288, 201, 30]), ("vidor_x", 2, 128, [128, 77])])
def test_absolute_position_encoding_matches_reference_golden(name, B, T, lens, precision):
    """`use_abs_pe: True` (no shipped config): position rows added in the last visual-embedding LayerNorm launch (CLIP variant:
    as a residual of the visual/CLIP fusion GEMM); table as is below max_len, re-interpolated from it on.  Eval and -- same
    values, differentiable path -- train mode with stochastic depth off."""
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, ic, keys = load_case(name)
    mc = dict(mc, use_abs_pe=True)
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]), strict=True)
    model = model.to(DEV).eval()
    g = np.load(os.path.join(GOLDEN, "abs_pe.npz"))
    x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=4321 + T)
    out = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    close(out["pred_logits"], g[f"{name}/T{T}_pred_logits"], LOGIT_TOL)
    close(out["pred_masks"], g[f"{name}/T{T}_pred_masks"], MASK_TOL)
    assert model.backbone.entity_reach() is None          # position rows count the pair's frames: nothing to share per tracklet
    if T <= mc["max_seq_len"]:
        # the autograd path (fresh tensors, torch.cat joins) computes the same numbers
        with torch.enable_grad():
            xg = x.to(DEV).requires_grad_(True)
            out2 = model._mask_vrd(xg, m.to(DEV), with_aux=False)
        close(out2["pred_logits"], g[f"{name}/T{T}_pred_logits"], LOGIT_TOL)


@pytest.mark.parametrize("name,B,T,lens", [("vidvrd", 3, 96, [96, 50, 7]), ("vidor_local", 2, 128, [128, 77])])
def test_relative_position_encoding_matches_reference_golden(name, B, T, lens, precision):
    """`use_rel_pe: True` (no shipped config): the stem / branch blocks' banded attention adds a learnable bias per (head,
    window slot) to its scores (the REL instantiation of the local-attention kernels); eval and the autograd path."""
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, ic, keys = load_case(name)
    with open(os.path.join(GOLDEN, "rel_pe_keys.json")) as f:
        extra = json.load(f)[name]
    mc = dict(mc, use_rel_pe=True)
    model = MaskVRD(mc, device=DEV)
    assert sorted(k for k, _ in model.named_parameters() if k.endswith("rel_pe")) == sorted(k for k, _ in extra)
    sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    sd.update(O.synth_state_dict(extra))
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    g = np.load(os.path.join(GOLDEN, "rel_pe.npz"))
    x, m = O.synth_pairs(B, c_in(mc), T, lens, seed=8765 + T)
    out = model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    close(out["pred_logits"], g[f"{name}/T{T}_pred_logits"], LOGIT_TOL)
    close(out["pred_masks"], g[f"{name}/T{T}_pred_masks"], MASK_TOL)
    with torch.enable_grad():
        out2 = model._mask_vrd(x.to(DEV).requires_grad_(True), m.to(DEV), with_aux=False)
        close(out2["pred_logits"], g[f"{name}/T{T}_pred_logits"], LOGIT_TOL)
        out2["pred_logits"].square().sum().backward()
    grads = [p.grad for k, p in model.named_parameters() if k.endswith("rel_pe")]
    assert all(gr is not None and gr.shape == (1, 1, mc["n_head"], mc["n_mha_win_size"]) and float(gr.abs().max()) > 0 for gr in grads)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_test_equals_single_process(world, monkeypatch):
    """MaskVRD.shard_pairs(): every rank runs its round-robin share of the length-sorted pairs and the ranks exchange
    compact candidates.  Here the ranks run one after another in this process (the collective is replaced by computing
    the other ranks' shares; the collective itself is covered by tests/test_parallel_cpu.py on gloo and by
    scripts/sharded_eval_check.py with real processes): every rank's result must equal the unsharded one exactly."""
    from golden_cases import SLICES
    from vrdone_amd import parallel
    model, mc, ic, _ = get_model("vidvrd")
    data = _on_device(synth_proposal(c_in=c_in(mc), **SLICES))
    want = model(data)
    feats = data["so_features_list"]
    lens = [int(f.shape[1]) for f in feats]
    order, t_pad = model.eval_plan(lens)
    P = len(lens)
    per = (P + world - 1) // world
    try:
        model.shard_pairs()
        for rank in range(world):
            def fake_all_gather(t, w, group=None):
                parts = []
                for r in range(w):
                    c = t if r == rank else model.pair_candidates(feats, lens, order[r::w], t_pad, model.topk)
                    if c.shape[0] < per:
                        c = torch.cat([c, c.new_zeros(per - c.shape[0], *c.shape[1:])], dim=0)
                    parts.append(c)
                return torch.stack(parts)
            monkeypatch.setattr(parallel, "rank_world", lambda group=None: (rank, world))
            monkeypatch.setattr(parallel, "_all_gather", fake_all_gather)
            got = model(data)
            for key in ("triplets", "pred_durations", "so_tids", "triple_scores", "triple_scores_avg", "so_trajs"):
                assert got[key] == want[key], (rank, key)
    finally:
        model.shard_pairs(enable=False)


def test_sharded_forward_test_from_tracklets(monkeypatch):
    """The same with a pair_source: every rank runs the entity stage for the tracklets of ITS pairs only
    (PairSource.stream_plan over its share) and the results still equal the unsharded call."""
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    from vrdone_amd import parallel
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model("vidvrd")
    vid_kw, dl_kw = PROPOSAL_CASES["vidvrd"]
    prop = prepare_test_proposal(P.synth_raw_video(**vid_kw), dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
    src = prop["pair_source"]
    want = model(prop)
    order, t_pad = model.eval_plan(src.lens)
    world = 3
    per = (len(src) + world - 1) // world
    try:
        model.shard_pairs()
        for rank in range(world):
            def fake_all_gather(t, w, group=None):
                parts = []
                for r in range(w):
                    c = t if r == rank else model.pair_candidates(None, src.lens, order[r::w], t_pad, model.topk, source=src)
                    if c.shape[0] < per:
                        c = torch.cat([c, c.new_zeros(per - c.shape[0], *c.shape[1:])], dim=0)
                    parts.append(c)
                return torch.stack(parts)
            monkeypatch.setattr(parallel, "rank_world", lambda group=None: (rank, world))
            monkeypatch.setattr(parallel, "_all_gather", fake_all_gather)
            assert model(prop) == want, rank
    finally:
        model.shard_pairs(enable=False)


@pytest.mark.parametrize("name", ["vidvrd", "strided"])
def test_gather_pairs_matches_the_reference_dataloader(name):
    """vrd_gather_pairs (per-tracklet rows on the device -> backbone operand buffers, box features computed on the way)
    against the pair matrices of the reference dataloader's `_test_getitem` (oracle.proposal restates it and is pinned
    bit for bit by tests/golden/proposal.npz): visual rows and the non-logarithmic box features bit-equal, the three
    log features within 2e-6."""
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    from vrdone_amd import ops
    from vrdone_amd.proposals import prepare_test_proposal
    vid_kw, dl_kw = PROPOSAL_CASES[name]
    raw = P.synth_raw_video(**vid_kw)
    want = P.test_getitem(raw, **dl_kw)["so_features_list"]
    prop = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
    src = prop["pair_source"]
    Pn, T = len(src), max(src.lens) + 3
    sel = torch.arange(Pn, device=DEV)
    vis, clip, so_box, ent, mask = ops.gather_pairs(src, sel, T, 5, 8, False)
    assert clip is None and mask.sum(1).tolist() == src.lens
    V = src.n_visual
    for p, f in enumerate(want):
        L = f.shape[1]
        ft = f.T.contiguous()
        assert torch.equal(vis[p, :L].cpu(), ft[:, :V]) and torch.equal(vis[Pn + p, :L].cpu(), ft[:, V:2 * V])
        box = ft[:, 2 * V:]
        got_so, got_s, got_o = so_box[p, :L].cpu(), ent[p, :L].cpu(), ent[Pn + p, :L].cpu()
        assert torch.equal(got_so[:, :2], box[:, :2])
        np.testing.assert_allclose(got_so[:, 2:].numpy(), box[:, 2:5].numpy(), rtol=0, atol=2e-6)
        assert torch.equal(got_s, box[:, 5:13]) and torch.equal(got_o, box[:, 13:21])
        for buf in (vis[p, L:], vis[Pn + p, L:], so_box[p, L:], ent[p, L:], ent[Pn + p, L:]):
            assert not bool(buf.any())
    # pair rows (bf16x3 operand format) decode to the same values to 2^-16
    visp, _, _, _, _ = ops.gather_pairs(src, sel, T, 5, 8, True)
    assert float((visp.float() - vis).abs().max()) <= 2.0 ** -15 * float(vis.abs().max())


def test_forward_test_from_tracklet_features_equals_pair_matrices(precision):
    """The whole eval call fed with per-tracklet features (proposals.prepare_test_proposal -> pair_source) returns what it
    returns when fed the reference dataloader's per-pair matrices."""
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model("vidvrd")
    vid_kw, dl_kw = PROPOSAL_CASES["vidvrd"]
    raw = P.synth_raw_video(**vid_kw)
    a = model(_on_device(P.test_getitem(raw, **dl_kw)))
    prop = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
    b = model({k: (v if k == "pair_source" else ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV))) for k, v in prop.items()})
    np.testing.assert_allclose(a["triple_scores_avg"], b["triple_scores_avg"], rtol=0, atol=2e-6 if precision != "bf16x3" else 2e-5)
    if precision == "bf16x3":
        # the 17-bit mode may rank two records differently where their scores lie closer together than its own score error (the
        # two input forms round the features differently): the same records, each at a rank whose score is within that error
        rec = lambda r, i: (tuple(r["triplets"][i]), tuple(r["so_tids"][i]), tuple(r["pred_durations"][i]),           # noqa: E731
                            tuple(map(tuple, r["so_trajs"][i][0])) if r["so_trajs"] else ())
        n = len(a["triplets"])
        assert len(b["triplets"]) == n
        where = {}
        for j in range(n):
            where.setdefault(rec(b, j), []).append(j)
        sc = a["triple_scores_avg"]
        lost = [i for i in range(n) if not any(abs(sc[j] - sc[i]) <= 2e-5 for j in where.get(rec(a, i), []))]
        # (a record whose score ties with the n_max_pair-th may fall off the end of one list)
        assert len(lost) <= 2 and all(abs(sc[i] - sc[-1]) <= 2e-5 for i in lost), lost
    else:
        assert a["triplets"] == b["triplets"] and a["pred_durations"] == b["pred_durations"] and a["so_tids"] == b["so_tids"]
        assert a["so_trajs"] == b["so_trajs"]
    # the plain-tensor form of the same source (what a dataset returns under the reference's eval loop), moved to the device
    # the way utils.dict_to_device does
    flat = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], None)
    c = model({k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV) if torch.is_tensor(v) else v) for k, v in flat.items()})
    assert c == b


@pytest.mark.parametrize("cfg_name,case,full", [("vidvrd", "vidvrd", False), ("vidvrd", "strided", True), ("vidor_x", "strided", False),
                                                ("vidor_x", "vidvrd", True)])
def test_entity_stage_once_per_tracklet_is_bit_equal(cfg_name, case, full, precision):
    """The backbone's entity stage (embeddings, visual/box fusion, first stem block) run once per tracklet and pieced
    together per pair (MaskVRD._entity_streams / vrd_assemble_pairs) gives, bit for bit, the rows of running it on every
    pair's subject and object: tracklets of different spans (pair windows that start / end inside a tracklet), pairs
    shorter than an edge piece, feat_stride 1 and 4 with an offset, attention windows 7 and 9."""
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    from vrdone_amd import ops
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model(cfg_name)
    bb = model.backbone
    vid_kw, dl_kw = PROPOSAL_CASES[case]
    raw = P.synth_raw_video(**dict(vid_kw, n_visual=bb.n_visual, n_clip=bb.n_clip))
    prop = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
    src = prop["pair_source"]
    ids = sorted(range(len(src)), key=lambda i: src.lens[i])
    shared = model._entity_streams(src, ids)
    assert shared is not None and shared[3] == bb.entity_reach() == 3 + bb.mha_win_size[0] // 2
    from vrdone_amd.proposals import PairSource
    lens = src.lens_dev.clone()
    piece, buf = shared[2]
    # windows cut short (of long pairs, so that they stay inside the tracklets): edge pieces that overlap / cover the pair
    lens[torch.tensor(ids[-8:-2])] = torch.tensor([2, 3, piece - 1, piece, piece + 1, buf], dtype=torch.int32, device=DEV)
    if full:        # pairs that fill their T frames (no padded frame behind them): cut the long pairs' windows to T
        T = (max(src.lens) - 1) // 24 * 24
        lens.clamp_(max=T)
    else:
        T = -(-max(src.lens) // 96) * 96
    src = PairSource(src.vis, src.clip, src.boxes, src.s_row, src.o_row, lens, src.stride, src.wh, src.first_row)
    assert max(src.lens) >= 2 * buf and (not full or sum(n == T for n in src.lens) >= 2)
    sel = torch.tensor(ids, device=DEV)
    so, so_box, mask = model._shared_entity_rows(src, sel, shared, 0, T)
    vis, clip, so_box2, ent, mask2 = ops.gather_pairs(src, sel, T, bb.n_bbox_so, bb.n_bbox_entity, ops.pair_mode())
    want = bb.entity_stage(vis, clip, ent, torch.cat([mask2, mask2], dim=0))
    assert torch.equal(mask, mask2) and torch.equal(so_box, so_box2)
    assert torch.equal(so, want), f"max abs diff {float((so - want).abs().max()):.3g}"
    # and the frames at a window edge do differ from the tracklet's own somewhere (the case exercises the pieces)
    rows, stream_row = shared[0].view(-1, so.shape[-1]), shared[1].reshape(-1).tolist()
    assert any(not torch.equal(rows[r], want[e, 0]) for e, r in enumerate(stream_row))


@pytest.mark.parametrize("cfg_name,case", [("vidvrd", "vidvrd"), ("vidor_x", "strided")])
def test_forward_test_sharing_tracklets_changes_nothing(cfg_name, case, precision):
    from golden_cases import PROPOSAL_CASES
    from oracle import proposal as P
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model(cfg_name)
    vid_kw, dl_kw = PROPOSAL_CASES[case]
    raw = P.synth_raw_video(**dict(vid_kw, n_visual=model.backbone.n_visual, n_clip=model.backbone.n_clip))
    prop = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], DEV)
    old_stride = model.feat_stride
    model.feat_stride = dl_kw["feat_stride"]
    try:
        a = model(prop)
        model.share_tracklets = False
        b = model(prop)
    finally:
        model.share_tracklets, model.feat_stride = True, old_stride
    assert a == b


def test_forward_test_matches_oracle_small():
    model, mc, ic, sd = get_model("vidvrd")
    data = synth_proposal(4, c_in(mc), 10, 110, seed=99)
    want = O.forward_test(sd, mc, ic, data)
    dev_data = {k: ([t.to(DEV) for t in v] if isinstance(v, list) else v.to(DEV)) for k, v in data.items()}
    got = model(dev_data)
    assert len(got["triplets"]) == len(want["triplets"])
    np.testing.assert_allclose(got["triple_scores_avg"], want["triple_scores_avg"], atol=2e-5, rtol=0)
    same = sum(a == b and c == d for a, b, c, d in zip(got["triplets"], want["triplets"],
                                                        got["pred_durations"], want["pred_durations"]))
    assert same >= len(want["triplets"]) - 4


def test_full_size_properties():
    """The north-star shape itself (BASELINE metric: 2048 pairs x 256 frames, T_pad 288 -- the bench workload): finite outputs,
    masked frames filled with -10, rows equal to a small-batch run (other kernels: no 256 x 256 tiles, no padding maps), and
    -- batch-composition independence at full size -- the reversed batch gives the reversed outputs BIT FOR BIT."""
    model, mc, _, _ = get_model("vidvrd")
    B, T = 2048, 288
    gen = torch.Generator(device=DEV).manual_seed(5)
    lens = torch.randint(2, 257, (B,), generator=torch.Generator().manual_seed(6))
    lens[:4] = torch.tensor([256, 201, 97, 2])
    lens[-1] = 288
    m = (torch.arange(T)[None] < lens[:, None])[:, None].to(DEV)
    x = torch.randn(B, c_in(mc), T, device=DEV, generator=gen).mul_(m)
    out = model._mask_vrd(x, m, with_aux=False)
    assert out["pred_logits"].shape == (B, 9, 133) and out["pred_masks"].shape == (B, 9, T)
    assert bool(torch.isfinite(out["pred_logits"]).all()) and bool(torch.isfinite(out["pred_masks"]).all())
    assert bool((out["pred_masks"].masked_select(~m.expand(-1, 9, -1)) == -10.0).all())
    small = model._mask_vrd(x[:4].contiguous(), m[:4].contiguous(), with_aux=False)
    close(small["pred_logits"], out["pred_logits"][:4], 1e-5)
    close(small["pred_masks"], out["pred_masks"][:4], 1e-4)
    rev = model._mask_vrd(x.flip(0), m.flip(0), with_aux=False)
    assert torch.equal(rev["pred_logits"].flip(0), out["pred_logits"])
    assert torch.equal(rev["pred_masks"].flip(0), out["pred_masks"])


@pytest.mark.parametrize("B,T", [(384, 512), (4096, 512)])
def test_full_size_properties_vidor_x(B, T):
    """BASELINE config 5 (vidor_x.yaml: C_in 3093 with the CLIP slabs, 8 heads, window 9), at its full 4096 pairs x 512
    frames (26 GB of input, two 2048-pair chunks) and at 384 pairs: finite outputs, -10 on masked frames, and rows equal to a
    3-pair run of the small-shape kernels."""
    model, mc, _, _ = get_model("vidor_x")
    gen = torch.Generator(device=DEV).manual_seed(15)
    lens = torch.randint(2, T + 1, (B,), generator=torch.Generator().manual_seed(16))
    lens[:3] = torch.tensor([512, 333, 65])
    m = (torch.arange(T)[None] < lens[:, None])[:, None].to(DEV)
    x = torch.randn(B, c_in(mc), T, device=DEV, generator=gen).mul_(m)
    out = model._mask_vrd(x, m, with_aux=False)
    Q = mc["predictor"]["num_queries"]
    assert out["pred_logits"].shape == (B, Q, mc["num_classes"] + 1) and out["pred_masks"].shape == (B, Q, T)
    assert bool(torch.isfinite(out["pred_logits"]).all()) and bool(torch.isfinite(out["pred_masks"]).all())
    assert bool((out["pred_masks"].masked_select(~m.expand(-1, Q, -1)) == -10.0).all())
    small = model._mask_vrd(x[:3].contiguous(), m[:3].contiguous(), with_aux=False)
    close(small["pred_logits"], out["pred_logits"][:3], 1e-5)
    close(small["pred_masks"], out["pred_masks"][:3], 1e-4)
    del x, out
    torch.cuda.empty_cache()


def test_sharded_forward_test_at_the_8_gpu_job_size(monkeypatch):
    """BASELINE config 4's job (8192 pairs x 256 frames over 8 ranks; here 91 tracklets = 8190 ordered pairs minus the vIoU
    duplicates, 1024 per rank): the 8 ranks run one after another in this process on one GPU (each rank's candidates computed
    once; the collective itself is covered on gloo and by scripts/sharded_eval_check.py), every rank's result equal to the
    unsharded call."""
    from vrdone_amd import parallel, synth
    from vrdone_amd.proposals import prepare_test_proposal
    model, mc, ic, _ = get_model("vidvrd")
    raw = synth.synth_raw_video(91, mc["visual_dim"], 200, 256, seed=11)
    prop = prepare_test_proposal(raw, 1, 0, 2, DEV)
    src = prop["pair_source"]
    assert len(src) > 8000
    want = model(prop)
    assert want is not None and len(want["triplets"]) > 0
    order, t_pad = model.eval_plan(src.lens)
    world = 8
    per = (len(src) + world - 1) // world
    cands = {}

    def share(r):
        if r not in cands:
            c = model.pair_candidates(None, src.lens, order[r::world], t_pad, model.topk, source=src)
            if c.shape[0] < per:
                c = torch.cat([c, c.new_zeros(per - c.shape[0], *c.shape[1:])], dim=0)
            cands[r] = c
        return cands[r]
    try:
        model.shard_pairs()
        for rank in range(world):
            def fake_all_gather(t, w, group=None):
                return torch.stack([t if r == rank else share(r) for r in range(w)])
            monkeypatch.setattr(parallel, "rank_world", lambda group=None: (rank, world))
            monkeypatch.setattr(parallel, "_all_gather", fake_all_gather)
            assert model(prop) == want, rank
    finally:
        model.shard_pairs(enable=False)


def test_extension_is_loaded_and_profiled():
    """The HIP library is the code that ran: its per-family event profile sees the launches."""
    from vrdone_amd import _hip
    model, mc, _, _ = get_model("vidvrd")
    x, m = O.synth_pairs(2, c_in(mc), 96, [96, 40], seed=3)
    _hip.prof_enable(True)
    _hip.prof_reset()
    model._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
    torch.cuda.synchronize()
    prof = _hip.prof_read()
    _hip.prof_enable(False)
    gemm = {k: sum(prof[f][k] for f in ("gemm_f32_mfma", "gemm_x3_mfma", "gemm_x3_dma"))
            for k in ("launches", "ms", "flops")}
    assert gemm["launches"] > 50 and gemm["ms"] > 0 and gemm["flops"] > 1e9
    # 8 SOS self/cross attentions + the predictor's 4 x (self, cross) on the MFMA kernels; 5 banded attentions
    assert prof["attn_flash"]["launches"] == 16 and prof["attn_small"]["launches"] == 0
    assert prof["local_attn"]["launches"] == 5


def test_pack_pairs_equals_padded_batch(precision):
    """Batching straight from the frame-major per-pair matrices == zero-padded (B, C_in, T) batch + unpack."""
    from vrdone_amd import ops
    model, mc, _, _ = get_model("vidvrd")
    data = synth_proposal(4, c_in(mc), 10, 110, seed=5)
    feats = [f.to(DEV) for f in data["so_features_list"]]
    feats = [f.t().contiguous().t() for f in feats]              # (C, L) views of contiguous (L, C) matrices
    ids = list(range(len(feats)))
    T = 144
    bb = model.backbone
    table, lens = ops.pair_table(feats)
    *parts, m2 = ops.pack_pairs(table, lens, T, bb.n_visual, bb.n_clip, bb.n_bbox_so, bb.n_bbox_entity, ops.pair_mode())
    assert ops.pair_table([f.contiguous() for f in feats]) is None          # not frame-major -> generic batching path
    got = model._heads(*bb.cl_parts(*parts, m2), False)
    x, m = model._batch(feats, ids, T)
    want = model._mask_vrd(x, m, with_aux=False)
    assert torch.equal(m2, m[:, 0])
    assert torch.equal(got["pred_logits"], want["pred_logits"]) and torch.equal(got["pred_masks"], want["pred_masks"])


@pytest.mark.parametrize("name,T", [("vidvrd", 96), ("vidor_x", 512), ("vidor_local", 512)])
def test_forward_training_loss_values_match_reference_golden(name, T, precision):
    """Training-mode forward under no_grad = batching + HIP network + matcher + losses: the loss dict of the
    reference (its matcher / losses run on its own predictions, scripts/make_golden.py) within 1e-3 relative,
    with the same Hungarian matches; with autograd recording the same call is a training step: total_loss.backward()
    gives every parameter a finite gradient (vidvrd, vidor_x with the CLIP slabs, vidor_local with banded SOS layers)."""
    from oracle.synth import synth_relations
    model, mc, _, _ = get_model(name)
    g = np.load(os.path.join(GOLDEN, f"mask_vrd_{name}.npz"))
    with open(os.path.join(GOLDEN, f"criterion_{name}.json")) as f:
        want = json.load(f)[f"T{T}"]
    lens = g[f"T{T}_lengths"].tolist()
    x, m = O.synth_pairs(len(lens), c_in(mc), T, lens, seed=1234 + T)
    gp, gm, gs = synth_relations(lens, T, mc["num_classes"], seed=want["seed"])
    data = {"so_features_list": [x[i, :, :n].contiguous() for i, n in enumerate(lens)],
            "preds_list": gp, "masks_list": gm}
    if mc.get("with_fuzzy", False):
        data["segs_list"] = gs
    model.train()
    try:
        got = model(data)
        xb, mb = model.preprocessing(data["so_features_list"])
        assert xb.shape == (len(lens), c_in(mc), mc["max_seq_len"]) and torch.equal(mb.cpu(), m)
        with torch.enable_grad():
            model.zero_grad(set_to_none=True)
            step = model(data)
            assert step["total_loss"].requires_grad
            step["total_loss"].backward()
        missing = [n for n, p in model.named_parameters() if p.grad is None or not bool(torch.isfinite(p.grad).all())]
        assert missing == [], missing
    finally:
        model.zero_grad(set_to_none=True)
        model.eval()
    assert list(got) == list(want["losses"])
    for k, v in want["losses"].items():
        assert got[k].is_cuda and abs(float(got[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(got[k]), v)
    out = model._mask_vrd(x.to(DEV), m.to(DEV))
    idx, _ = model.bipartite_match(out["pred_logits"], [t.to(DEV) for t in gp], out["pred_masks"],
                                   [t.to(DEV) for t in gm], [t.to(DEV) for t in gs] if "segs_list" in data else None,
                                   _mask=out["output_mask"])
    # a pair of < 16 frames has one valid frame at the predictor's T/8 level and prices its queries almost identically
    # (near-ties decided by 1e-5 differences): the matches are compared on pairs with enough frames to separate the
    # queries, the loss values above on all of them
    for n, (i, j), w in zip(lens, idx, want["indices"]):
        if n >= 16:
            assert [i.tolist(), j.tolist()] == w


def test_forward_training_24_pair_batch(precision):
    """BASELINE config 3 shape: 24 ragged pairs at T_pad 96 through model.train() + no_grad."""
    from oracle.synth import synth_relations
    model, mc, _, _ = get_model("vidvrd")
    with open(os.path.join(GOLDEN, "criterion_vidvrd.json")) as f:
        want = json.load(f)["train24"]
    lens = want["lengths"]
    x, _ = O.synth_pairs(len(lens), c_in(mc), 96, lens, seed=3)
    gp, gm, gs = synth_relations(lens, 96, mc["num_classes"], max_rel=4, seed=want["seed"])
    data = {"so_features_list": [x[i, :, :n].contiguous() for i, n in enumerate(lens)],
            "preds_list": gp, "masks_list": gm, "segs_list": gs}
    model.train()
    try:
        got = model(data)
    finally:
        model.eval()
    assert list(got) == list(want["losses"])
    for k, v in want["losses"].items():
        assert abs(float(got[k]) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(got[k]), v)


@pytest.mark.parametrize("name,B,T", [("vidvrd", 256, 288), ("vidor_x", 160, 512), ("vidor_local", 160, 512)])
def test_padding_skip_changes_nothing(name, B, T):
    """Batches large enough for the padding maps (GEMM block lists, flash key tiles, depthwise-conv strips) on ragged
    lengths: predictions identical to the last bit with the maps switched off (padded rows then hold the reference's
    constants instead of filler, and no valid row ever reads them)."""
    from vrdone_amd import ops
    model, mc, _, _ = get_model(name)
    lens = torch.randint(2, T + 1, (B,), generator=torch.Generator().manual_seed(77))
    lens[:3] = torch.tensor([T, T - 33, 2])
    m = (torch.arange(T)[None] < lens[:, None])[:, None].to(DEV)
    x = torch.randn(B, c_in(mc), T, device=DEV, generator=torch.Generator(device=DEV).manual_seed(78)) * m
    old = ops._skip_padding
    try:
        ops._skip_padding = True
        got = model._mask_vrd(x, m)
        ops._skip_padding = False
        want = model._mask_vrd(x, m)
    finally:
        ops._skip_padding = old
    assert torch.equal(got["pred_logits"], want["pred_logits"]) and torch.equal(got["pred_masks"], want["pred_masks"])
    for a, b in zip(got["aux_outputs"], want["aux_outputs"]):
        assert torch.equal(a["pred_logits"], b["pred_logits"]) and torch.equal(a["pred_masks"], b["pred_masks"])


@pytest.mark.parametrize("B,T,mode", [(1082, 96, "ragged"), (315, 240, "ragged"), (450, 288, "short"), (429, 288, "tail")])
def test_padding_skip_on_odd_shapes(B, T, mode):
    """Row counts that are not whole tiles / segments, T_pad not a multiple of 32, mostly-padding batches: identical
    bits with the padding maps off, and the first rows identical to a 3-pair batch (small-shape kernels, no maps)."""
    from vrdone_amd import ops
    model, mc, _, _ = get_model("vidvrd")
    g = torch.Generator().manual_seed(B + T)
    lens = {"ragged": torch.randint(1, T + 1, (B,), generator=g), "short": torch.randint(1, T // 4, (B,), generator=g),
            "tail": torch.full((B,), T - 32)}[mode]
    lens[0] = T
    m = (torch.arange(T)[None] < lens[:, None])[:, None].to(DEV)
    x = torch.randn(B, c_in(mc), T, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * m
    old = ops._skip_padding
    try:
        ops._skip_padding = True
        got = model._mask_vrd(x, m, with_aux=False)
        ops._skip_padding = False
        want = model._mask_vrd(x, m, with_aux=False)
    finally:
        ops._skip_padding = old
    assert torch.equal(got["pred_logits"], want["pred_logits"]) and torch.equal(got["pred_masks"], want["pred_masks"])
    small = model._mask_vrd(x[:3].contiguous(), m[:3].contiguous(), with_aux=False)
    close(small["pred_logits"], got["pred_logits"][:3], 1e-5)
    close(small["pred_masks"], got["pred_masks"][:3], 1e-4)
