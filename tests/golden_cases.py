"""Parameters of the round-2 golden cases (scripts/make_golden_r2.py holds the same numbers: the generator runs only in
the build container and is not imported by tests)."""
import torch

SLICES = dict(n_tracklets=23, min_len=30, max_len=250, seed=2718, sort_by_length=True)
VIDOR_X = dict(n_tracklets=5, min_len=150, max_len=800, seed=1618, feat_stride=4, random_offset=True)
B256 = dict(B=256, T=288, seed=31415, every=16)


def b256_lengths():
    g = torch.Generator().manual_seed(B256["seed"])
    lens = torch.randint(2, B256["T"] + 1, (B256["B"],), generator=g)
    lens[::16] = torch.tensor([288, 287, 256, 255, 200, 129, 97, 96, 64, 33, 32, 31, 17, 3, 2, 288])
    return lens.tolist()


def compare_forward_test(res, ref, n_max_pair, score_tol, slack):
    """A forward_test result against a stored reference result: same number of triplets, sorted scores within
    score_tol, and the same (triplet, tracklets, duration) records and box-track digests up to `slack` entries (the
    ranking can only differ where two scores are closer than the arithmetic noise; slack = 0 demands identity)."""
    import numpy as np
    assert len(res["triplets"]) == len(ref["triplets"]) == n_max_pair
    np.testing.assert_allclose(res["triple_scores_avg"], ref["triple_scores_avg"], atol=score_tol, rtol=0)
    n = len(ref["triplets"])
    same = sum(a == b for a, b in zip(res["triplets"], ref["triplets"]))
    assert same >= n - slack, f"{n - same} ranks differ"
    key = lambda r, i: (tuple(r["triplets"][i]), tuple(r["so_tids"][i]), tuple(r["pred_durations"][i]))   # noqa: E731
    got, want = {key(res, i) for i in range(n)}, {key(ref, i) for i in range(n)}
    assert len(got & want) >= len(want) - slack
    dig = {(len(t[0]), round(float(np.sum(np.asarray(t, dtype=np.float64))), 3)) for t in res["so_trajs"]}
    wdig = {(int(a), round(b, 3)) for a, b in ref["so_trajs_digest"]}
    assert len(dig & wdig) >= len(wdig) - slack
