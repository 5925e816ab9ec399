"""Pair-sharded multi-GPU inference.  Pairs are independent end to end (no cross-pair op on the
path), so ranks take contiguous blocks of pairs with replicated weights and no data-path
collective; the one exchange step is an all-gather of the per-pair predictions in front of the
global top-n_max_pair selection (reference models/maskvrd.py:319-328).  Backend "nccl" is RCCL
over xGMI on the MI355X box; the same code runs on "gloo" in the CPU tests."""
import torch
import torch.distributed as dist


def shard_range(n_pairs, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: the first n_pairs % world ranks get one extra pair."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_predictions(pred_logits, pred_masks, n_pairs, world):
    """All-gather per-pair predictions of every rank's shard into full (n_pairs, ...) tensors.
    One collective per tensor; uneven shards are padded to the largest shard for the exchange."""
    if world == 1:
        return pred_logits, pred_masks
    per = (n_pairs + world - 1) // world
    outs = []
    for t in (pred_logits, pred_masks):
        t = t.contiguous()
        if t.shape[0] < per:
            t = torch.cat([t, t.new_zeros(per - t.shape[0], *t.shape[1:])], dim=0)
        full = t.new_empty(world * per, *t.shape[1:])
        dist.all_gather_into_tensor(full, t)
        if n_pairs % world:
            keep = torch.cat([torch.arange(r * per, r * per + (shard_range(n_pairs, r, world)[1] - shard_range(n_pairs, r, world)[0]))
                              for r in range(world)]).to(full.device)
            full = full.index_select(0, keep)
        outs.append(full)
    return tuple(outs)
