"""Eval-time pair construction on the device (with the host half of 8f-2).

The reference's test dataloader (dataloaders/vidvrd.py:552-715, dataloaders/vidor.py:640-735) clamps the tracklet
boxes to the frame, drops tracklets shadowed by a same-category tracklet (vIoU > 0.9), and then builds, per ordered
(subject, object) pair, an (L, C_in) matrix on the host by slicing / concatenating the two tracklets' per-frame features
and computing 21 box-feature channels -- every tracklet is copied into 2 (N - 1) pair matrices, which then travel to the
device one by one (utils/misc.py:98-112).

`prepare_test_proposal` keeps the cheap, irregular part on the host (clamp, de-dup, the pair list and its offsets) and
uploads each tracklet's rows ONCE; `MaskVRD.forward_test` then gathers pair rows and computes their box features on the
device (`vrd_gather_pairs`), straight into the backbone's operand buffers.  The returned dict is the reference's eval
proposal with `pair_source` in place of `so_features_list`; everything else (`sids`, `oids`, `cat_ids`, `cat_scores`,
`traj_durations`, `bboxes_list`, `so_offset`) is what `_test_getitem` returns.
"""
import numpy as np
import torch


class PairSource:
    """Per-tracklet rows on the device + per-pair tables: pair p = `lens[p]` frames, frame t = row s_row[p] + t*stride of
    the subject's tracklet and o_row[p] + t*stride of the object's in the concatenated (sum L, .) arrays."""

    def __init__(self, vis, clip, boxes, s_row, o_row, lens, stride, wh, first_row=None):
        self.vis, self.clip, self.boxes = vis, clip, boxes
        self.s_row, self.o_row, self.lens_dev = s_row, o_row, lens
        self.lens = lens.tolist()
        self.stride, self.wh = int(stride), (float(wh[0]), float(wh[1]))
        self.n_visual = vis.shape[1]
        self.n_clip = 0 if clip is None else clip.shape[1]
        # first_row (n_tracklets + 1,): tracklet k owns rows [first_row[k], first_row[k + 1]) -- what lets the model run its
        # per-entity stage once per tracklet (stream_plan); None: pairs only
        self.first_row = None if first_row is None else np.asarray(first_row, dtype=np.int64)
        self._rows_host = None

    def __len__(self):
        return len(self.lens)

    FIELDS = ("tracklet_visual", "tracklet_boxes", "tracklet_first_row", "pair_s_row", "pair_o_row", "pair_lens", "video_wh")

    def fields(self):
        """The source as plain proposal entries -- tensors and one int, the only value types the reference's eval loop
        lets through (`utils.dict_to_device`, utils/misc.py:98-112, raises on anything else): what a dataset running in
        DataLoader workers returns (CPU tensors); `from_fields` puts the object back together in `forward_test`."""
        d = {"tracklet_visual": self.vis, "tracklet_boxes": self.boxes, "tracklet_first_row": torch.as_tensor(self.first_row),
             "pair_s_row": self.s_row, "pair_o_row": self.o_row, "pair_lens": self.lens_dev,
             "video_wh": torch.tensor(self.wh, dtype=torch.float32), "pair_stride": self.stride}
        if self.clip is not None:
            d["tracklet_clip"] = self.clip
        return d

    @classmethod
    def from_fields(cls, d, device):
        to = lambda t: t.to(device)            # noqa: E731
        w, h = d["video_wh"].tolist()
        return cls(to(d["tracklet_visual"]), to(d["tracklet_clip"]) if "tracklet_clip" in d else None, to(d["tracklet_boxes"]),
                   to(d["pair_s_row"]), to(d["pair_o_row"]), to(d["pair_lens"]), int(d["pair_stride"]), (w, h),
                   first_row=d["tracklet_first_row"].cpu().numpy())

    def stream_plan(self, ids):
        """The sub-sampled tracklets ("streams") the pairs `ids` read from.  A pair's subject frames are rows
        s_row + t*stride of one tracklet: frames phase, phase + stride, ... of it, from frame number j0 on, with
        phase = (s_row - tracklet start) % stride.  Returns (start row (n,), length (n,)) of the distinct
        (tracklet, phase) streams, and per pair in `ids` the (stream index, j0) of its subject and of its object:
        numpy arrays stream (2, len(ids)), j0 (2, len(ids))."""
        assert self.first_row is not None
        if self._rows_host is None:
            self._rows_host = (self.s_row.cpu().numpy(), self.o_row.cpu().numpy())
        ids = np.asarray(ids, dtype=np.int64)
        rows = np.stack([self._rows_host[0][ids], self._rows_host[1][ids]])              # (2, n)
        trk = np.searchsorted(self.first_row, rows, side="right") - 1
        rel = rows - self.first_row[trk]
        phase, j0 = rel % self.stride, rel // self.stride
        key, stream = np.unique(trk * self.stride + phase, return_inverse=True)
        k_trk, k_phase = key // self.stride, key % self.stride
        start = self.first_row[k_trk] + k_phase
        length = -(-(self.first_row[k_trk + 1] - start) // self.stride)
        return start, length.astype(np.int32), stream.reshape(rows.shape), j0


def _clamped(boxes, w, h):
    b = boxes.clone()
    b[:, 0:2].clamp_(min=0)
    b[:, 2].clamp_(max=w - 1)
    b[:, 3].clamp_(max=h - 1)
    if not bool(((b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])).all()):
        raise ValueError("a tracklet box is empty after clamping to the frame")
    return b


def shadowed_tracklets(boxes, spans, cat_ids, threshold=0.9):
    """Indices of tracklets to drop (reference dataloaders/vidvrd.py:577-636): walking tracklets i < j of one category
    whose durations overlap, j goes when i spans j's whole duration and the per-frame box intersections (+1 pixel
    convention) sum to more than `threshold` of j's summed box area; i goes (and its scan stops) in the mirrored case."""
    n = len(boxes)
    gone = np.zeros(n, dtype=bool)
    for i in range(n):
        for j in range(i + 1, n):
            if gone[j] or cat_ids[i] != cat_ids[j]:
                continue
            lo, hi = max(spans[i][0], spans[j][0]), min(spans[i][1], spans[j][1])
            if hi <= lo:
                continue
            bi = boxes[i][lo - spans[i][0]:hi - spans[i][0]]
            bj = boxes[j][lo - spans[j][0]:hi - spans[j][0]]
            wh = (torch.minimum(bi[:, 2:], bj[:, 2:]) - torch.maximum(bi[:, :2], bj[:, :2]) + 1).clamp_(min=0)
            inter = float((wh[:, 0] * wh[:, 1]).sum())
            a_i = float((bi[:, 2] - bi[:, 0] + 1).mul(bi[:, 3] - bi[:, 1] + 1).sum())
            a_j = float((bj[:, 2] - bj[:, 0] + 1).mul(bj[:, 3] - bj[:, 1] + 1).sum())
            if inter / a_j > threshold and spans[i][0] <= spans[j][0] and spans[i][1] >= spans[j][1]:
                gone[j] = True
            elif inter / a_i > threshold and spans[j][0] <= spans[i][0] and spans[j][1] >= spans[i][1]:
                gone[i] = True
                break
    return np.nonzero(gone)[0].tolist()


def load_test_video(info_pkl, features_pkl):
    """The two per-video pickles of the reference's test split -> the dict its `_test_getitem` (and
    `prepare_test_proposal`) takes; restates `_prepare_test`, dataloaders/vidvrd.py:459-550.
      info_pkl      {'traj_proposal': {num_proposals, cat_ids (N,), scores (N,), bboxes_list [N x (L_i, 4)],
                     traj_durations (N, 2) with INCLUSIVE end frames, video_wh}}
      features_pkl  {frame id: {'frame_id', 'tids' [tracklet ids present], 'visual_features' [one (V,) row per tid]}}
    Returns {} for a video with fewer than two tracklets or without a pair of tracklets that share a frame."""
    import pickle
    with open(info_pkl, "rb") as f:
        prop = pickle.load(f)["traj_proposal"]
    if prop["num_proposals"] < 2:
        return {}
    spans = np.asarray(prop["traj_durations"]).astype(np.int64).copy()
    spans[:, 1] += 1                                           # [start, end)
    n = len(spans)
    s_id, o_id = np.meshgrid(np.arange(n), np.arange(n))       # the reference's pair order: object-major
    s_id, o_id = s_id.flatten(), o_id.flatten()
    keep = s_id != o_id
    s_id, o_id = s_id[keep], o_id[keep]
    overlap = np.minimum(spans[s_id, 1], spans[o_id, 1]) > np.maximum(spans[s_id, 0], spans[o_id, 0])
    if not overlap.any():
        return {}
    s_id, o_id = s_id[overlap], o_id[overlap]
    with open(features_pkl, "rb") as f:
        frames = pickle.load(f)
    rows = [[] for _ in range(n)]
    for fid in sorted(frames):
        rec = frames[fid]
        assert rec["frame_id"] == fid
        for i, tid in enumerate(rec["tids"]):
            assert spans[tid][0] <= fid < spans[tid][1]
            rows[tid].append(np.asarray(rec["visual_features"][i]))
    assert all(len(r) == spans[t][1] - spans[t][0] for t, r in enumerate(rows)), "a tracklet misses frames in the feature file"
    return {"sids": torch.as_tensor(s_id, dtype=torch.int64), "oids": torch.as_tensor(o_id, dtype=torch.int64),
            "cat_ids": torch.as_tensor(np.asarray(prop["cat_ids"]), dtype=torch.int64),
            "cat_scores": torch.as_tensor(np.asarray(prop["scores"]), dtype=torch.float32),
            "bboxes_list": [torch.as_tensor(np.asarray(b), dtype=torch.float32) for b in prop["bboxes_list"]],
            "traj_durations": torch.as_tensor(spans, dtype=torch.int64),
            "visual_features_list": [torch.as_tensor(np.stack(r, axis=0), dtype=torch.float32) for r in rows],
            "video_wh": prop["video_wh"]}


def prepare_test_proposal(raw, feat_stride, stride_offset, proposal_min_frames, device, viou_threshold=0.9):
    """raw: the dict `_prepare_test` hands to `_test_getitem` (sids, oids, cat_ids, cat_scores, bboxes_list,
    traj_durations [start, end), visual_features_list, optional clip_features_list, video_wh).  Returns {} when no pair
    survives, else the eval proposal with a device-resident `pair_source` -- or, with device=None, with the source as
    plain CPU tensor entries (PairSource.fields): the form for a dataset's `_test_getitem` under the reference's unmodified
    eval loop (DataLoader workers, `utils.dict_to_device`)."""
    w, h = raw["video_wh"]
    boxes = [_clamped(b, w, h) for b in raw["bboxes_list"]]
    spans = [(int(a), int(e)) for a, e in raw["traj_durations"].tolist()]
    dropped = set(shadowed_tracklets(boxes, spans, raw["cat_ids"].tolist(), viou_threshold))
    first_row = np.concatenate([[0], np.cumsum([len(b) for b in boxes])])            # tracklet -> first row of its frames
    sids, oids, s_row, o_row, lens = [], [], [], [], []
    for s, o in zip(raw["sids"].tolist(), raw["oids"].tolist()):
        if s in dropped or o in dropped:
            continue
        lo, hi = max(spans[s][0], spans[o][0]), min(spans[s][1], spans[o][1])
        shared = hi - lo
        steps = len(range(stride_offset, shared, feat_stride)) if shared > 0 else 0
        if shared < proposal_min_frames or steps < 2:
            continue
        sids.append(s)
        oids.append(o)
        s_row.append(first_row[s] + lo - spans[s][0] + stride_offset)
        o_row.append(first_row[o] + lo - spans[o][0] + stride_offset)
        lens.append(steps)
    if not sids:
        return {}
    clip = raw.get("clip_features_list")
    flat = device is None
    device = "cpu" if flat else device
    src = PairSource(
        torch.cat(raw["visual_features_list"], dim=0).to(device=device, dtype=torch.float32).contiguous(),
        None if clip is None else torch.cat(clip, dim=0).to(device=device, dtype=torch.float32).contiguous(),
        torch.cat(boxes, dim=0).to(device=device, dtype=torch.float32).contiguous(),
        torch.tensor(s_row, dtype=torch.int64, device=device), torch.tensor(o_row, dtype=torch.int64, device=device),
        torch.tensor(lens, dtype=torch.int32, device=device), feat_stride, (w, h), first_row=first_row)
    out = {"sids": torch.tensor(sids), "oids": torch.tensor(oids), "cat_ids": raw["cat_ids"], "cat_scores": raw["cat_scores"],
           "traj_durations": raw["traj_durations"], "bboxes_list": boxes,
           "so_offset": torch.full((len(sids),), stride_offset, dtype=torch.int64)}
    if flat:
        out.update(src.fields())
    else:
        out["pair_source"] = src
    return out


# --------------------------------------------------------------------------------------------------------------------
# Training side: the per-video ground-truth cache and the per-step sample construction of the reference's
# training dataloader.  Host code (the sample construction draws from Python's `random`, call for call like the reference,
# so that a seeded run sees the same crops); the model side takes the resulting lists as `forward_training` input.
# --------------------------------------------------------------------------------------------------------------------
def load_train_video(anno_json, gt_features_pkl, entity_cat_name_to_id, pred_cat_name_to_id):
    """One training video's cache entry; restates `_prepare_train`, dataloaders/vidvrd.py:172-322.
      anno_json        VidVRD annotation: {'height', 'width', 'trajectories' [per frame: [{'tid', 'bbox': {xmin, ymin, xmax,
                       ymax}}]], 'subject/objects' [{'tid', 'category'}], 'relation_instances' [{'subject_tid',
                       'object_tid', 'predicate', 'begin_fid', 'end_fid'}]}
      gt_features_pkl  {1-based frame id: {'frame_id', 'tids' array, 'visual_features' (n, V) array}} of the ground-truth boxes
    Trajectories are renumbered 0..n-1 in tid order and cut into intervals of consecutive frames; relation instances of
    one (subject, object, predicate) that overlap in time are merged; every relation is keyed by (subject, object, subject
    interval, object interval).  Returns {} for a video without relations, else {'video_hw', 'relation_merged' {key:
    [{'predicate', 'begin_fid', 'end_fid'}]}, 'relation_keys' [[...]], 'visual_features' / 'entity_bboxes' {index: [per
    interval tensor]}, 'entity_classes', 'traj_intervals' {index: [[start, end)]}}."""
    import copy
    import json
    import pickle
    from collections import defaultdict
    with open(anno_json) as f:
        anno = json.load(f)
    if len(anno["relation_instances"]) == 0:
        return {}
    with open(gt_features_pkl, "rb") as f:
        frames = pickle.load(f)
    present = defaultdict(list)
    for fid, frame in enumerate(anno["trajectories"]):
        for box in frame:
            present[box["tid"]].append(fid)
    tids = sorted(present)
    index_of = {tid: i for i, tid in enumerate(tids)}
    frame_keys = sorted(frames)
    visual, bboxes, intervals = {}, {}, {}
    for tid in tids:
        fids = np.asarray(sorted(present[tid]))
        cut = np.nonzero(np.diff(fids) > 1)[0]
        starts = fids[np.concatenate([[0], cut + 1])]
        ends = fids[np.concatenate([cut, [len(fids) - 1]])] + 1
        spans = [[int(a), int(e)] for a, e in zip(starts, ends)]
        idx = index_of[tid]
        intervals[idx] = spans
        vis_i, box_i = [], []
        for a, e in spans:
            rows = []
            for k in frame_keys:                         # frame ids in the feature file start at 1
                if k - 1 < a:
                    continue
                if k - 1 >= e:
                    break
                rec = frames[k]
                assert rec["frame_id"] == k
                at = np.nonzero(np.asarray(rec["tids"]) == tid)[0]
                assert len(at) == 1
                rows.append(np.asarray(rec["visual_features"])[at])
            vis_i.append(torch.tensor(np.concatenate(rows, axis=0)))
            bb = [[b["bbox"]["xmin"], b["bbox"]["ymin"], b["bbox"]["xmax"], b["bbox"]["ymax"]]
                  for frame in anno["trajectories"][a:e] for b in frame if b["tid"] == tid]
            assert len(bb) == e - a
            box_i.append(torch.tensor(bb, dtype=torch.float32))
        visual[idx], bboxes[idx] = vis_i, box_i
    classes = {index_of[so["tid"]]: entity_cat_name_to_id[so["category"]] for so in anno["subject/objects"]}

    # merge instances of the same (subject, object, predicate) that overlap in time, in order of their first frame
    insts = sorted(copy.deepcopy(anno["relation_instances"]), key=lambda r: r["begin_fid"])
    merged, seen = [], [False] * len(insts)
    for i, base in enumerate(insts):
        if seen[i]:
            continue
        seen[i] = True
        for j in range(i + 1, len(insts)):
            other = insts[j]
            if (other["subject_tid"], other["object_tid"], other["predicate"]) != (base["subject_tid"], base["object_tid"], base["predicate"]):
                continue
            assert other["begin_fid"] > base["begin_fid"]
            if other["begin_fid"] <= base["end_fid"]:
                assert other["end_fid"] > base["end_fid"]
                base["end_fid"] = other["end_fid"]
                seen[j] = True
        merged.append(copy.deepcopy(base))
    merged.sort(key=lambda r: r["begin_fid"])

    # (the keys are collected in a set and listed in ITS iteration order, like the reference: `pair_duration` slices index
    # that list, and for tuples of ints the order is a deterministic function of the insertions)
    relation_merged, keys = defaultdict(list), set()
    for r in merged:
        s, o = index_of[r["subject_tid"]], index_of[r["object_tid"]]
        bf, ef = r["begin_fid"], r["end_fid"]
        s_iv = [k for k, (a, e) in enumerate(intervals[s]) if a <= bf and e >= ef]
        o_iv = [k for k, (a, e) in enumerate(intervals[o]) if a <= bf and e >= ef]
        assert len(s_iv) == 1 and len(o_iv) == 1, "a relation must lie inside one interval of each trajectory"
        key = (s, o, s_iv[0], o_iv[0])
        assert max(intervals[s][s_iv[0]][0], intervals[o][o_iv[0]][0]) < min(intervals[s][s_iv[0]][1], intervals[o][o_iv[0]][1])
        relation_merged[key].append({"predicate": pred_cat_name_to_id[r["predicate"]], "begin_fid": bf, "end_fid": ef})
        keys.add(key)
    return {"video_hw": (anno["height"], anno["width"]), "relation_merged": relation_merged,
            "relation_keys": [list(k) for k in keys], "visual_features": visual, "entity_bboxes": bboxes,
            "entity_classes": classes, "traj_intervals": intervals}


def truncate_feats(so_feat, pred, segment, max_seq_len, trunc_thresh=0.5, max_times=10, rng=None):
    """Random max_seq_len crop of a pair that keeps at least one relation >= trunc_thresh inside (reference
    utils/misc.py:219-273, which derives from ActionFormer); None when ten draws find none.  so_feat (C, L), segment
    (N, 2) in feature steps.  Draws `rng.randint(0, L - max_seq_len)` per try, like the reference's `random.randint`."""
    import random
    rng = rng or random
    L = so_feat.shape[1]
    if L <= max_seq_len:
        return so_feat, pred, segment
    for _ in range(max_times):
        st = rng.randint(0, L - max_seq_len)
        ed = st + max_seq_len
        left = torch.clamp(segment[:, 0], min=st).float()
        right = torch.clamp(segment[:, 1], max=ed).float()
        inter = (right - left).clamp(min=0)
        keep = inter / (segment[:, 1] - segment[:, 0]).abs() >= trunc_thresh
        if int(keep.sum()) > 0:
            return so_feat[:, st:ed], pred[keep], torch.stack((left[keep], right[keep]), dim=1) - st
    return None


def _so_box_features(sb, ob):
    """utils/misc.py:158-178 (what vrd_gather_pairs computes on the device for the eval path)."""
    s_cx, s_cy = (sb[:, 2] + sb[:, 0]) / 2, (sb[:, 3] + sb[:, 1]) / 2
    o_cx, o_cy = (ob[:, 2] + ob[:, 0]) / 2, (ob[:, 3] + ob[:, 1]) / 2
    s_w, s_h, o_w, o_h = sb[:, 2] - sb[:, 0], sb[:, 3] - sb[:, 1], ob[:, 2] - ob[:, 0], ob[:, 3] - ob[:, 1]
    return torch.stack([(s_cx - o_cx) / o_cx, (s_cy - o_cy) / o_cy, torch.log(s_w / o_w), torch.log(s_h / o_h),
                        torch.log(s_w * s_h / (o_w * o_h))], dim=1)


def _entity_box_features(b, w, h):
    """utils/misc.py:181-217: normalised centre / size and their frame-to-frame differences (first frame extrapolated)."""
    x0, x1, y0, y1 = b[:, 0] / w, b[:, 2] / w, b[:, 1] / h, b[:, 3] / h
    geo = torch.stack([(x1 + x0) / 2, (y1 + y0) / 2, x1 - x0, y1 - y0], dim=1)
    d = geo[1:] - geo[:-1]
    first = d[0:1] - (d[1:2] - d[0:1]) if len(d) > 1 else d[0:1]
    d = torch.cat([first, d], dim=0)
    return torch.stack([geo[:, 0], d[:, 0], geo[:, 1], d[:, 1], geo[:, 2], d[:, 2], geo[:, 3], d[:, 3]], dim=1)


def train_getitem(video, feat_stride, max_seq_len, cut_max_preds=False, proposal_max_preds=0, pair_duration=None, rng=None):
    """The training sample of one video (or of its relation keys [pair_duration[0], pair_duration[1])): restates
    `_train_getitem`, dataloaders/vidvrd.py:324-457.  Per relation key: a random sub-sampling offset, the subject / object
    features of the frames both intervals share, the 21 box-feature channels, the relations as [ceil((fid - start -
    offset) / stride)] segments, a random max_seq_len crop, 0/1 masks.  Returns {} when nothing survives, else
    {'so_features_list' [(C_in, L)], 'preds_list', 'masks_list' [(N, max_seq_len)], 'segs_list'}.  A cache entry with
    'clip_features' {index: [per interval (L, Cc)]} (the VidOR loader with CLIP features, dataloaders/vidor.py:346-415) puts
    the subject / object CLIP rows behind the visual ones."""
    import copy
    import random
    rng = rng or random
    if not video:
        return {}
    video = copy.deepcopy(video)
    merged, keys = video["relation_merged"], video["relation_keys"]
    if pair_duration is not None:
        keys = keys[pair_duration[0]:pair_duration[1]]
        merged = {k: v for k, v in merged.items() if list(k) in keys}
    h, w = video["video_hw"]
    boxes = {t: [_clamped(b, w, h) for b in per] for t, per in video["entity_bboxes"].items()}
    feats_out, preds_out, masks_out, segs_out = [], [], [], []
    for key in merged:
        offset = rng.randint(0, feat_stride - 1)
        s, o, si, oi = key
        if cut_max_preds and proposal_max_preds < len(merged[key]):
            continue
        s_iv, o_iv = video["traj_intervals"][s][si], video["traj_intervals"][o][oi]
        lo, hi = max(s_iv[0], o_iv[0]), min(s_iv[1], o_iv[1])
        pick = lambda per, iv: per[lo - iv[0]:hi - iv[0]][offset::feat_stride]      # noqa: E731
        s_feat, o_feat = pick(video["visual_features"][s][si], s_iv), pick(video["visual_features"][o][oi], o_iv)
        if s_feat.shape[0] < 2:
            continue
        sb, ob = pick(boxes[s][si], s_iv), pick(boxes[o][oi], o_iv)
        clip = video.get("clip_features")
        wide = [s_feat, o_feat] if clip is None else [s_feat, o_feat, pick(clip[s][si], s_iv), pick(clip[o][oi], o_iv)]
        so_feat = torch.cat(wide + [_so_box_features(sb, ob), _entity_box_features(sb, w, h), _entity_box_features(ob, w, h)],
                            dim=-1).permute(1, 0)
        preds, segs = [], []
        for r in merged[key]:
            left = np.ceil((r["begin_fid"] - lo - offset) / feat_stride)
            right = np.ceil((r["end_fid"] - lo - offset) / feat_stride)
            if left < right:
                preds.append(r["predicate"])
                segs.append([left, right])
        if not preds:
            continue
        cropped = truncate_feats(so_feat, torch.tensor(preds, dtype=torch.int64), torch.tensor(np.array(segs), dtype=torch.int64),
                                 max_seq_len, rng=rng)
        if cropped is None:
            continue
        so_feat, preds, segs = cropped
        segs_out.append(segs)
        masks = torch.zeros(len(segs), max_seq_len, dtype=torch.float32)
        for m, (a, e) in zip(masks, segs.to(torch.int64).tolist()):
            assert 0 <= a < e <= max_seq_len
            m[a:e] = 1
        feats_out.append(so_feat)
        preds_out.append(preds)
        masks_out.append(masks)
    if not feats_out:
        return {}
    return {"so_features_list": feats_out, "preds_list": preds_out, "masks_list": masks_out, "segs_list": segs_out}


def train_policy(video_num_pairs, num_pairs):
    """Steps of at most num_pairs relation keys, videos cut across steps where needed: [[(video, (first, last))]] per step;
    restates `apply_policy`, dataloaders/vidvrd.py:100-135."""
    policy, current = [[]], 0
    for name, n in video_num_pairs:
        if n + current < num_pairs:
            policy[-1].append([name, (0, n)])
            current += n
            continue
        start = 0
        while n + current >= num_pairs:
            take = num_pairs - current
            policy[-1].append([name, (start, start + take)])
            n -= take
            start += take
            current = 0
            policy.append([])
        if n > 0:
            policy[-1].append([name, (start, start + n)])
            current += n
    return policy
