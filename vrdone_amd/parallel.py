"""Pair-sharded multi-GPU inference.  Pairs are independent end to end (no cross-pair op on the
path), so ranks take disjoint sets of pairs with replicated weights and no data-path
collective; the one exchange step is an all-gather of per-pair results in front of the
global top-n_max_pair selection (reference models/maskvrd.py:319-328).  Backend "nccl" is RCCL
over xGMI on the MI355X box; the same code runs on "gloo" in the CPU tests and in the single-GPU rehearsal
(scripts/sharded_eval_check.py), where device tensors are staged through the host for the collective.

Two payloads:
  gather_candidates   compact per-pair candidates of MaskVRD.forward_test, (Q, 2k + 2) floats per pair (648 B for
                      vidvrd): what the product path exchanges (MaskVRD.shard_pairs()).
  gather_predictions  raw pred_logits + pred_masks (15.2 kB per pair at T_pad 288): BASELINE config 4's "all-gather of
                      per-pair masks", used by bench.py --gpus N.
"""
import os

import torch
import torch.distributed as dist


def forced():
    """VRDONE_FORCE_COLLECTIVE=1: run the exchange step even in a group of ONE rank (an all-gather with itself), so that a
    single-GPU box executes the RCCL calls of the N > 1 path (tests/test_gpu_model.py::test_rccl_path_on_one_rank)."""
    return os.environ.get("VRDONE_FORCE_COLLECTIVE") == "1" and dist.is_available() and dist.is_initialized()


def rank_world(group=None):
    """(rank, world size) of `group`; (0, 1) without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_range(n_pairs, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: the first n_pairs % world ranks get one extra pair."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _all_gather(t, world, group=None):
    """(n, ...) -> (world, n, ...), same n on every rank.  RCCL takes device tensors directly; any other backend
    (gloo) gets a host copy."""
    t = t.contiguous()
    n = t.shape[0]
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        full = t.new_empty(world * n, *t.shape[1:])          # concatenated along dim 0 (the form gloo accepts too)
        dist.all_gather_into_tensor(full, t, group=group)
        return full.view(world, n, *t.shape[1:])
    host = t.cpu()
    host_full = host.new_empty(world * n, *host.shape[1:])
    dist.all_gather_into_tensor(host_full, host, group=group)
    return host_full.to(t.device).view(world, n, *t.shape[1:])


def gather_candidates(cand, n_pairs, group=None):
    """cand: this rank's (ceil-or-floor(n_pairs / world), ...) records for the pairs order[rank::world] of a common
    ordering (round-robin sharding).  Returns the (n_pairs, ...) records in that ordering, identical on every rank.
    ONE collective: shards are padded to ceil(n_pairs / world) rows; position j * world + r of the interleaved result is
    rank r's j-th record, so the padding lands behind the last real record and needs no index table."""
    rank, world = rank_world(group)
    if world == 1 and not forced():
        return cand
    per = (n_pairs + world - 1) // world
    mine = (n_pairs - rank + world - 1) // world
    assert cand.shape[0] == mine, f"rank {rank} holds {cand.shape[0]} records, its share of {n_pairs} is {mine}"
    if mine < per:
        cand = torch.cat([cand, cand.new_zeros(per - mine, *cand.shape[1:])], dim=0)
    full = _all_gather(cand, world, group)                     # (world, per, ...)
    return full.transpose(0, 1).reshape(world * per, *cand.shape[1:])[:n_pairs]


def gather_predictions(pred_logits, pred_masks, n_pairs, world, group=None):
    """All-gather per-pair predictions of contiguous shards (shard_range) into full (n_pairs, ...) tensors.
    One collective per tensor; uneven shards are padded to the largest shard for the exchange and the padding rows
    (the tail of the short shards) are dropped by slicing, shard by shard."""
    if world == 1 and not forced():
        return pred_logits, pred_masks
    per = (n_pairs + world - 1) // world
    sizes = [shard_range(n_pairs, r, world) for r in range(world)]
    outs = []
    for t in (pred_logits, pred_masks):
        t = t.contiguous()
        if t.shape[0] < per:
            t = torch.cat([t, t.new_zeros(per - t.shape[0], *t.shape[1:])], dim=0)
        full = _all_gather(t, world, group)                    # (world, per, ...)
        if n_pairs % world == 0:
            full = full.reshape(world * per, *t.shape[1:])
        else:
            full = torch.cat([full[r, :hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
        outs.append(full)
    return tuple(outs)
