"""world_size-2 rehearsal of the pair-sharded path on the gloo backend (CPU): shard arithmetic and
the all-gather of per-pair predictions, including uneven shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vrdone_amd.parallel import gather_candidates, gather_predictions, shard_range


def test_shard_range_partitions():
    for n in (1, 7, 8, 2048, 2049, 8192):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        logits = torch.randn(n_pairs, 9, 133, generator=g)
        masks = torch.randn(n_pairs, 9, 96, generator=g)
        lo, hi = shard_range(n_pairs, rank, world)
        got_l, got_m = gather_predictions(logits[lo:hi], masks[lo:hi], n_pairs, world)
        q.put((rank, bool(torch.equal(got_l, logits) and torch.equal(got_m, masks))))
    finally:
        dist.destroy_process_group()


def _cand_worker(rank, world, port, n_pairs, q):
    """Compact forward_test candidates (Q = 9, k = 8: 18 floats per query, integer fields bit-cast) of the pairs
    order[rank::world]; the gathered tensor must be the full record list in `order`, bit for bit, on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(5)
        full = torch.randint(-2 ** 31, 2 ** 31 - 1, (n_pairs, 9, 18), generator=g, dtype=torch.int64).to(torch.int32)
        mine = full[rank::world].contiguous().view(torch.float32)
        got = gather_candidates(mine, n_pairs)
        q.put((rank, bool(torch.equal(got.view(torch.int32), full))))
    finally:
        dist.destroy_process_group()


def _run(target, world, n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


@pytest.mark.parametrize("world,n_pairs", [(2, 494), (2, 7), (3, 20), (3, 2)])
def test_gather_candidates(world, n_pairs):
    _run(_cand_worker, world, n_pairs)


def test_eval_plan_follows_the_reference_slice_rule():
    """MaskVRD.eval_plan gives every pair the padded length the reference's slice loop gives it (oracle.preprocess_eval
    restates models/maskvrd.py:363-414 per slice), and its order is a permutation grouped by that length."""
    from conftest import load_case
    from oracle import vrd_oracle as O
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, _ = load_case("vidvrd")
    model = MaskVRD(mc, device="cpu")
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(2, 260, (431,), generator=g).tolist()
    feats = [torch.empty(1, n) for n in lens]
    step = mc["max_so_pair"]
    model.tight_padding = False         # first the reference's own padded lengths ...
    order, t_pad = model.eval_plan(lens)
    for s0 in range(0, len(lens), step):
        for part in O.preprocess_eval(mc, feats[s0:s0 + step]):
            if part is not None:
                x, _, ids = part
                assert all(t_pad[s0 + i] == x.shape[-1] for i in ids)
    t_ref = t_pad
    model.tight_padding = True          # ... then the tight ones: never longer than the reference's, a padded frame left at every
    model.row_space = False             # pyramid level unless the reference has none either, multiples of 32.  Bucket by bucket:
    assert model.eval_plan(lens)[1] == t_ref    # (431 pairs are fewer rows than one bucket: that policy leaves them alone)
    model.TIGHT_MIN_ROWS = 8192
    down = 8

    def check(t_pad):
        for L, t, T in zip(lens, t_pad, t_ref):
            assert t <= T and t % 32 == 0 and (t == T or t // down > -(-L // down))
        assert sum(t_pad) < 0.8 * sum(t_ref)
    order, t_coarse = model.eval_plan(lens)
    check(t_coarse)
    del model.row_space                 # the default, all buckets in one row space: buckets as fine as 4 k rows
    order, t_pad = model.eval_plan(lens)
    check(t_pad)
    assert sum(t_pad) <= sum(t_coarse) and len(set(t_pad)) >= len(set(t_coarse))
    assert sorted(order) == list(range(len(lens)))
    keys = [(t_pad[i], lens[i]) for i in order]
    assert keys == sorted(keys)
    for world in (2, 3, 8):        # round-robin shares: every rank gets the same number of pairs of each padded length, +-1
        for T in set(t_pad):
            counts = [sum(t_pad[i] == T for i in order[r::world]) for r in range(world)]
            assert max(counts) - min(counts) <= 1


@pytest.mark.parametrize("n_pairs", [8, 7])
def test_gather_predictions_world2(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_tight_plan_from_attached_lengths_equals_the_one_read_back_from_the_mask():
    """Masks built by MaskVRD._batch carry the lengths they were built from, so the tight-padding plan needs no read-back;
    a mask tensor from elsewhere (a clone: no attribute) gets the same plan from its own bits; editing the mask in place
    invalidates the attached lengths (version counter).  Host logic only: runs on CPU tensors."""
    from conftest import load_case
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, _ = load_case("vidvrd")
    model = MaskVRD(mc, device="cpu").eval()
    g = torch.Generator().manual_seed(9)
    lens = [96, 95, 94] + torch.randint(2, 90, (200,), generator=g).tolist()
    feats = [torch.zeros(4, n) for n in lens]
    model.ROWS_MIN_ROWS = 256
    x, m = model._batch(feats, range(len(feats)), 96)
    assert m._vrd_lens[1] == lens
    plan = model._tight_plan(m, m.reshape(len(lens), 96))
    other = m.clone()
    assert not hasattr(other, "_vrd_lens")
    plan2 = model._tight_plan(other, other.reshape(len(lens), 96))
    assert plan["rows"] and plan2["rows"] and len(plan["buckets"]) == len(plan2["buckets"]) >= 3
    for (t, idx, n, flat), (t2, idx2, n2, flat2) in zip(plan["buckets"], plan2["buckets"]):
        assert (t, n, flat) == (t2, n2, flat2) and torch.equal(idx, idx2)
        assert all((lens[i] <= t - 2) == flat and t % 32 == 0 for i in idx.tolist())
    assert [f for _, _, _, f in plan["buckets"]] == sorted((f for _, _, _, f in plan["buckets"]), reverse=True)   # flat buckets first
    m[0, 0, 50:] = False                                   # in place: pair 0 is 50 frames long now
    plan3 = model._tight_plan(m, m.reshape(len(lens), 96))
    t_of_0 = [t for t, idx, _, _ in plan3["buckets"] if 0 in idx.tolist()]
    assert t_of_0 == [64]
