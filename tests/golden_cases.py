"""Parameters of the round-2 golden cases (scripts/make_golden_r2.py holds the same numbers: the generator runs only in
the build container and is not imported by tests)."""
import torch

SLICES = dict(n_tracklets=23, min_len=30, max_len=250, seed=2718, sort_by_length=True)
VIDOR_X = dict(n_tracklets=5, min_len=150, max_len=800, seed=1618, feat_stride=4, random_offset=True)
# scripts/make_golden_r2.py --only-forward-test-variants
FORWARD_TEST_VARIANTS = {"vidor": dict(n_tracklets=5, min_len=150, max_len=800, seed=2618, feat_stride=4, random_offset=True),
                         "vidor_local": dict(n_tracklets=5, min_len=150, max_len=700, seed=3618, feat_stride=4, random_offset=True)}
B256 = dict(B=256, T=288, seed=31415, every=16)
CFG2 = dict(B=1024, T=144, frames=128, seed=27182, every=64)       # scripts/make_golden_r2.py --only-cfg2


def cfg2_lengths():
    lens = [CFG2["frames"]] * CFG2["B"]
    for i, n in zip(range(0, CFG2["B"], CFG2["every"]), [128, 127, 144, 143, 97, 96, 65, 64, 33, 32, 2, 128, 100, 113, 129, 140]):
        lens[i] = n
    return lens


def b256_lengths():
    g = torch.Generator().manual_seed(B256["seed"])
    lens = torch.randint(2, B256["T"] + 1, (B256["B"],), generator=g)
    lens[::16] = torch.tensor([288, 287, 256, 255, 200, 129, 97, 96, 64, 33, 32, 31, 17, 3, 2, 288])
    return lens.tolist()


def compare_forward_test(res, ref, n_max_pair, score_tol, slack, tie_tol=0.0):
    """A forward_test result against a stored reference result: same number of triplets, sorted scores within
    score_tol, and the same (triplet, tracklets, duration) records and box-track digests up to `slack` entries (the
    ranking can only differ where two scores are closer than the arithmetic noise; slack = 0 demands identity -- which
    holds for the goldens in both precision modes, measured score error 1e-7 (f32) / 1e-6 (bf16x3)).  tie_tol: ranks inside a tie
    of the reference's own scores may be permuted (vidor_local returns ALL 180 candidates, the last of them with scores that
    differ by less than the arithmetic noise)."""
    import numpy as np
    assert len(res["triplets"]) == len(ref["triplets"]) <= n_max_pair          # (fewer candidates than n_max_pair: all of them)
    np.testing.assert_allclose(res["triple_scores_avg"], ref["triple_scores_avg"], atol=score_tol, rtol=0)
    n = len(ref["triplets"])
    key = lambda r, i: (tuple(r["triplets"][i]), tuple(r["so_tids"][i]), tuple(r["pred_durations"][i]))   # noqa: E731
    # a rank may differ only inside a tie: the record found at rank i stands, in the reference, at a rank whose score is
    # within tie_tol of rank i's (tie_tol = 0: no rank may differ beyond `slack`)
    sc = ref["triple_scores_avg"]
    where = {}
    for j in range(n):
        where.setdefault(key(ref, j), []).append(j)
    off = [i for i in range(n) if key(res, i) != key(ref, i)]
    untied = [i for i in off if not any(abs(sc[j] - sc[i]) <= tie_tol for j in where.get(key(res, i), []))] if tie_tol > 0 else off
    assert len(untied) <= slack, f"{len(untied)} ranks differ outside ties: {untied[:8]}"
    got, want = {key(res, i) for i in range(n)}, {key(ref, i) for i in range(n)}
    assert len(got & want) >= len(want) - slack
    dig = {(len(t[0]), round(float(np.sum(np.asarray(t, dtype=np.float64))), 3)) for t in res["so_trajs"]}
    wdig = {(int(a), round(b, 3)) for a, b in ref["so_trajs_digest"]}
    assert len(dig & wdig) >= len(wdig) - slack


# ---- training step (scripts/make_golden_train.py)
TRAIN = dict(B=24, T=96, seed_len=2024, seed_x=3, seed_gt=2025)
TRAIN_VIDOR = dict(B=6, T=512, seed_len=3024, seed_x=5, seed_gt=3025)      # scripts/make_golden_train.py --vidor
# scripts/make_golden_train.py --vidor-variants: the CLIP backbone (vidor_x.yaml), the banded SOS attention (vidor_local.yaml)
TRAIN_SPECS = {"vidor": TRAIN_VIDOR, "vidor_x": dict(B=4, T=512, seed_len=4024, seed_x=11, seed_gt=4025),
               "vidor_local": dict(B=4, T=512, seed_len=5024, seed_x=9, seed_gt=5025)}


def train_batch(mc, c_in, device="cpu", spec=TRAIN):
    """The training batch of a train_step golden (24 pairs x 96 frames for vidvrd.yaml, 6 x 512 for vidor.yaml), in the
    dataloader's training format (dataloaders/vidvrd.py:451-457): so_features_list (C_in, L_i), preds_list, masks_list
    (N_i, max_seq_len), segs_list."""
    from oracle import vrd_oracle as O
    from oracle.synth import synth_relations
    B, T = spec["B"], spec["T"]
    lens = torch.randint(2, T + 1, (B,), generator=torch.Generator().manual_seed(spec["seed_len"])).tolist()
    x, m = O.synth_pairs(B, c_in, T, lens, seed=spec["seed_x"])
    gp, gm, gs = synth_relations(lens, T, mc["num_classes"], max_rel=4, seed=spec["seed_gt"])
    to = lambda ts: [t.to(device) for t in ts]      # noqa: E731
    data = {"so_features_list": to([x[i, :, :n].contiguous() for i, n in enumerate(lens)]),
            "preds_list": to(gp), "masks_list": to(gm), "segs_list": to(gs)}
    return lens, x, m, data


def replay_matching(model, recorded, tie_tol=1e-5):
    """Make model.bipartite_match return the reference's recorded assignments (final head, then the auxiliary layers, in
    call order) instead of its own: both sides then differentiate the same loss function even where a cost matrix
    has a near-tie (pairs shorter than 16 frames have ONE valid frame at the predictor's T/8 level: their queries cost
    the same to ~1e-5, tests/test_oracle_golden.py::test_criterion_train24).  Returns a list that collects, per call,
    the pairs on which the model's own matching differs from the recorded one.
    A differing pair has to BE such a tie: under the model's own cost matrix (the one its own assignment minimises) the
    recorded assignment of that pair costs at most tie_tol more than the model's -- a different prediction, or a broken
    cost kernel, would show as a gap (asserted here, per call and pair)."""
    from vrdone_amd.models import losses
    real = model.bipartite_match
    state = {"call": 0}
    differing = []

    def own_costs(pred_logits, gt_preds, pred_masks, gt_masks, gt_segs, _mask):
        dev = pred_logits.device
        sizes = [len(p) for p in gt_preds]
        owner = torch.repeat_interleave(torch.arange(len(sizes), device=dev), torch.tensor(sizes, device=dev))
        segs, scale_range = model._fuzzy([t.to(dev) for t in gt_segs] if gt_segs is not None else None)
        cc, cm, cd = losses.pair_costs(pred_logits, pred_masks, _mask[:, 0], torch.cat([t.to(dev) for t in gt_preds]),
                                       torch.cat([t.to(dev) for t in gt_masks]), owner, segs, scale_range)
        cost = model.cost_factor['cost_class'] * cc + model.cost_factor['cost_mask'] * cm + model.cost_factor['cost_dice'] * cd
        return cost.detach().double().cpu().split(sizes, dim=0)           # per pair: (N_i, Q)

    def match(pred_logits, gt_preds, pred_masks, gt_masks, gt_segs, _mask):
        idx, lm = real(pred_logits, gt_preds, pred_masks, gt_masks, gt_segs, _mask=_mask)
        rec = recorded[state["call"]]
        state["call"] += 1
        diff = [n for n, ((i, j), r) in enumerate(zip(idx, rec)) if not (i.tolist() == r[0] and j.tolist() == r[1])]
        if diff:
            blocks = own_costs(pred_logits, gt_preds, pred_masks, gt_masks, gt_segs, _mask)
            for n in diff:
                total = lambda q, r: float(sum(blocks[n][rr, qq] for qq, rr in zip(q, r)))       # noqa: E731
                own, ref = total(idx[n][0].tolist(), idx[n][1].tolist()), total(rec[n][0], rec[n][1])
                assert abs(own - ref) <= tie_tol * max(1.0, abs(own)), \
                    f"matcher call {state['call'] - 1}, pair {n}: own assignment costs {own:.7f}, the reference's {ref:.7f} -- not a tie"
        differing.append(diff)
        return [(torch.tensor(r[0], dtype=torch.int64), torch.tensor(r[1], dtype=torch.int64)) for r in rec], lm
    model.bipartite_match = match
    return differing


def compare_grads(named_grads, golden_npz, meta, case, rtol, atol_frac=1e-6, median_tol=None, outlier_tol=None, max_outliers=3,
                  outlier_scope=None):
    """Every parameter's gradient against the stored reference gradient: full tensor when it has <= 2048 elements, the
    stride-`sample_stride` sample otherwise; error measured relative to the l2 norm of the stored entries (plus
    atol_frac of the largest gradient norm of the model, for gradients that are ~0).  rtol bounds the worst parameter,
    median_tol the median over parameters; at most `max_outliers` parameters may exceed `outlier_tol` (the worst-case bound
    has to leave room for a max-pool arg-max that flips on a rounding difference; this one keeps that room from hiding a
    systematic error); `outlier_scope`: regular expression of the parameter names that count applies to (default: all).
    Returns (worst, median) relative error."""
    import re
    import numpy as np
    stride = meta["sample_stride"]
    stats = meta["cases"][case]["grad_stats"]
    biggest = max(s[2] for s in stats.values())
    worst, worst_name, errs, outliers = 0.0, None, [], []
    for name, g in named_grads:
        assert g is not None, f"{name} received no gradient"
        g = g.detach().float().cpu()
        assert bool(torch.isfinite(g).all()), name
        want = golden_npz[f"{case}/{name}"]
        got = (g if g.numel() <= 2048 else g.flatten()[::stride]).numpy()
        assert got.shape == want.shape, name
        err = float(np.linalg.norm(got.astype(np.float64) - want)) / (float(np.linalg.norm(want)) + atol_frac * biggest)
        errs.append(err)
        if outlier_tol is not None and err > outlier_tol and (outlier_scope is None or re.match(outlier_scope, name)):
            outliers.append((name, err))
        if err > worst:
            worst, worst_name = err, name
        # whole-tensor checksum: the l2 norm of the full gradient
        l2 = float(g.double().norm())
        assert abs(l2 - stats[name][2]) <= rtol * stats[name][2] + atol_frac * biggest, (name, l2, stats[name][2])
    assert worst <= rtol, f"{worst_name}: relative gradient error {worst:.3e} > {rtol:.1e}"
    median = float(np.median(errs))
    assert median_tol is None or median <= median_tol, f"median relative gradient error {median:.3e} > {median_tol:.1e}"
    assert len(outliers) <= max_outliers, (f"{len(outliers)} parameters exceed a relative gradient error of {outlier_tol:.1e} "
                                           f"(allowed: {max_outliers}): {outliers[:8]}")
    return worst, median


# ---- eval-time pair construction (scripts/make_golden_r2.py proposal_case)
PROPOSAL_CASES = {   # name -> (oracle.proposal.synth_raw_video kwargs, dataloader settings)
    "vidvrd": (dict(n_tracklets=8, video_len=120, min_len=12, max_len=100, seed=5),
               dict(feat_stride=1, stride_offset=0, proposal_min_frames=2)),
    "strided": (dict(n_tracklets=6, video_len=400, min_len=30, max_len=380, seed=6),
                dict(feat_stride=4, stride_offset=2, proposal_min_frames=5)),
}
