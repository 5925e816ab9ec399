"""world_size-2 rehearsal of the pair-sharded path on the gloo backend (CPU): shard arithmetic and
the all-gather of per-pair predictions, including uneven shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vrdone_amd.parallel import gather_predictions, shard_range


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
