"""Deterministic synthetic proposals for the eval path (TEST INFRASTRUCTURE ONLY,
same rules as vrd_oracle.py).  Shapes follow the producer of the reference's eval
inputs, dataloaders/vidvrd.py:706-715 (keys sids, oids, so_features_list, bboxes_list,
cat_ids, cat_scores, traj_durations, so_offset)."""
import torch


def synth_proposal(n_tracklets, c_in, min_len, max_len, seed=4321, feat_stride=1, video_len=None, random_offset=False,
                   sort_by_length=False):
    """All ordered pairs of n tracklets whose durations overlap by >= 2 feature steps.
    Feature tensors are handed over the way the reference dataloader does: an (L, C)
    row-major tensor viewed as (C, L) (dataloaders/vidvrd.py:693).
    random_offset: each pair starts its temporal sub-sampling at a random frame offset in [0, feat_stride), as the
    dataloader does with `random_stride` (dataloaders/vidor.py:660, :678-692: feat[offset::feat_stride]) -- so_offset != 0.
    sort_by_length: tracklets in ascending length, so the pairs of early subjects are short and the slices of the
    reference's max_so_pair loop (models/maskvrd.py:208) pad their long pairs to different lengths."""
    g = torch.Generator().manual_seed(seed)
    video_len = video_len or (max_len * feat_stride + 16)
    durs, boxes = [], []
    for _ in range(n_tracklets):
        L = int(torch.randint(min_len * feat_stride, max_len * feat_stride + 1, (1,), generator=g))
        L = min(L, video_len)
        st = int(torch.randint(0, video_len - L + 1, (1,), generator=g))
        durs.append([st, st + L])
        xy = torch.rand(L, 2, generator=g) * 100.0
        wh = torch.rand(L, 2, generator=g) * 50.0 + 1.0
        boxes.append(torch.cat([xy, xy + wh], dim=1))
    if sort_by_length:
        order = sorted(range(n_tracklets), key=lambda i: durs[i][1] - durs[i][0])
        durs, boxes = [durs[i] for i in order], [boxes[i] for i in order]
    sids, oids, feats, offs = [], [], [], []
    for s in range(n_tracklets):
        for o in range(n_tracklets):
            if s == o:
                continue
            a, b = max(durs[s][0], durs[o][0]), min(durs[s][1], durs[o][1])
            off = int(torch.randint(0, feat_stride, (1,), generator=g)) if random_offset else 0
            n_steps = (b - a - off + feat_stride - 1) // feat_stride if b - off > a else 0
            if n_steps < 2:
                continue
            sids.append(s)
            oids.append(o)
            offs.append(off)
            feats.append(torch.randn(n_steps, c_in, generator=g).permute(1, 0))
    return {
        "sids": torch.tensor(sids), "oids": torch.tensor(oids),
        "so_features_list": feats, "bboxes_list": boxes,
        "cat_ids": torch.randint(1, 36, (n_tracklets,), generator=g),
        "cat_scores": torch.rand(n_tracklets, generator=g),
        "traj_durations": torch.tensor(durs), "so_offset": torch.tensor(offs),
    }


def synth_relations(lengths, t_pad, n_classes, max_rel=3, seed=777):
    """Ground truth the way the training dataloader hands it over (dataloaders/vidvrd.py:402-455): per pair
    preds (N_i,) int64 in [1, n_classes], segs (N_i, 2) int64 [start, end) inside the pair's length, and
    masks (N_i, t_pad) float32 with ones on [start, end)."""
    g = torch.Generator().manual_seed(seed)
    preds, masks, segs = [], [], []
    for L in lengths:
        n = int(torch.randint(1, max_rel + 1, (1,), generator=g))
        p = torch.randint(1, n_classes + 1, (n,), generator=g)
        sg = []
        for _ in range(n):
            a = int(torch.randint(0, max(L - 1, 1), (1,), generator=g))
            b = int(torch.randint(a + 1, L + 1, (1,), generator=g))
            sg.append([a, b])
        sg = torch.tensor(sg, dtype=torch.int64)
        m = torch.zeros(n, t_pad)
        for r, (a, b) in enumerate(sg.tolist()):
            m[r, a:b] = 1
        preds.append(p)
        masks.append(m)
        segs.append(sg)
    return preds, masks, segs
