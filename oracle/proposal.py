"""CPU restatement of the reference's eval-time pair construction (TEST INFRASTRUCTURE ONLY, same rules as
vrd_oracle.py: imported by tests/, never by vrdone_amd/).

Follows dataloaders/vidvrd.py:552-715 (`_test_getitem`; dataloaders/vidor.py:640-735 is the same code plus the CLIP
slabs): clamp the tracklet boxes to the frame, drop tracklets that a same-category tracklet covers with vIoU > 0.9,
then, for every remaining ordered (subject, object) pair, slice both tracklets to the frames they share, sub-sample
with feat_stride from stride_offset, and concatenate per frame
    [s_vis | o_vis | (s_clip | o_clip) | so_box 5 | s_box 8 | o_box 8]          
with the box features of utils/misc.py:158-217.  Pinned by tests/golden/proposal_*.npz, which
scripts/make_golden_r2.py produces by calling the reference's own `_test_getitem`.
"""
import torch

TO_REMOVE = 1


def so_box_features(sbbox, obbox):
    """utils/misc.py:158-178: [(s_cx-o_cx)/o_cx, (s_cy-o_cy)/o_cy, log(s_w/o_w), log(s_h/o_h), log(s_area/o_area)]."""
    s_cx, s_cy = (sbbox[:, 2] + sbbox[:, 0]) / 2, (sbbox[:, 3] + sbbox[:, 1]) / 2
    s_w, s_h = sbbox[:, 2] - sbbox[:, 0], sbbox[:, 3] - sbbox[:, 1]
    o_cx, o_cy = (obbox[:, 2] + obbox[:, 0]) / 2, (obbox[:, 3] + obbox[:, 1]) / 2
    o_w, o_h = obbox[:, 2] - obbox[:, 0], obbox[:, 3] - obbox[:, 1]
    return torch.stack([(s_cx - o_cx) / o_cx, (s_cy - o_cy) / o_cy, torch.log(s_w / o_w), torch.log(s_h / o_h),
                        torch.log((s_w * s_h) / (o_w * o_h))], dim=1)


def entity_box_features(bboxes, w, h):
    """utils/misc.py:181-217: on boxes normalised by the frame size, [cx, d cx, cy, d cy, w, d w, h, d h] with d = first
    difference along the (sub-sampled) frames; the first frame's difference is extrapolated linearly from the next two
    (d0 - (d1 - d0)), or copied when there is only one."""
    b = bboxes.clone()
    b[:, 0:4:2] /= w
    b[:, 1:4:2] /= h
    cols = [(b[:, 2] + b[:, 0]) / 2, (b[:, 3] + b[:, 1]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]]
    feats = []
    for v in cols:
        d = v[1:] - v[:-1]
        first = d[0] - (d[1] - d[0]) if len(d) > 1 else d[0]
        feats += [v, torch.cat([first.reshape(1), d])]
    return torch.stack(feats, dim=1)


def clamp_boxes(bboxes_list, w, h):
    """dataloaders/vidvrd.py:567-575."""
    out = []
    for b in bboxes_list:
        b = b.clone()
        b[:, 0].clamp_(min=0)
        b[:, 1].clamp_(min=0)
        b[:, 2].clamp_(max=w - 1)
        b[:, 3].clamp_(max=h - 1)
        assert bool((b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all())
        out.append(b)
    return out


def dedup_tracklets(bboxes_list, durations, cat_ids, viou_threshold=0.9):
    """dataloaders/vidvrd.py:577-636: tracklet j is dropped when a same-category tracklet i (i < j examined first)
    covers its whole duration and the box intersection summed over their shared frames exceeds viou_threshold of j's
    summed area (areas and intersections with the +1 pixel convention); symmetric rule for i.  Returns the kept ids."""
    n = len(bboxes_list)
    valid = [True] * n
    dur = [list(map(int, d)) for d in durations.tolist()]
    for i in range(n):
        for j in range(i + 1, n):
            if not valid[j] or int(cat_ids[i]) != int(cat_ids[j]):
                continue
            if dur[j][0] >= dur[i][1] or dur[j][1] <= dur[i][0]:
                continue
            a, e = max(dur[i][0], dur[j][0]), min(dur[i][1], dur[j][1])
            bi = bboxes_list[i][a - dur[i][0]: a - dur[i][0] + e - a]
            bj = bboxes_list[j][a - dur[j][0]: a - dur[j][0] + e - a]
            area_i = (bi[:, 2] - bi[:, 0] + TO_REMOVE) * (bi[:, 3] - bi[:, 1] + TO_REMOVE)
            area_j = (bj[:, 2] - bj[:, 0] + TO_REMOVE) * (bj[:, 3] - bj[:, 1] + TO_REMOVE)
            wh = (torch.min(bi[:, 2:], bj[:, 2:]) - torch.max(bi[:, :2], bj[:, :2]) + TO_REMOVE).clamp(min=0.0)
            inter = (wh[:, 0] * wh[:, 1]).sum()
            if inter / area_j.sum() > viou_threshold and dur[i][0] <= dur[j][0] and dur[i][1] >= dur[j][1]:
                valid[j] = False
            elif inter / area_i.sum() > viou_threshold and dur[j][0] <= dur[i][0] and dur[j][1] >= dur[i][1]:
                valid[i] = False
                break
    return [i for i in range(n) if valid[i]]


def test_getitem(input_dict, feat_stride=1, stride_offset=0, proposal_min_frames=2, viou_threshold=0.9):
    """The reference's `_test_getitem` (dataloaders/vidvrd.py:552-715) for a fixed stride offset.  input_dict: sids, oids
    (P,), traj_durations (N, 2) [start, end), bboxes_list, visual_features_list (optionally clip_features_list),
    cat_ids, cat_scores, video_wh.  Returns the eval proposal dict MaskVRD.forward_test consumes, or {}."""
    w_, h_ = input_dict["video_wh"]
    boxes = clamp_boxes(input_dict["bboxes_list"], w_, h_)
    dur = input_dict["traj_durations"]
    vis = input_dict["visual_features_list"]
    clip = input_dict.get("clip_features_list")
    keep = set(dedup_tracklets(boxes, dur, input_dict["cat_ids"], viou_threshold))
    sids, oids, feats, offs = [], [], [], []
    for s, o in zip(input_dict["sids"].tolist(), input_dict["oids"].tolist()):
        if s not in keep or o not in keep:
            continue
        a, e = max(int(dur[s][0]), int(dur[o][0])), min(int(dur[s][1]), int(dur[o][1]))
        n, sd, od = e - a, a - int(dur[s][0]), a - int(dur[o][0])
        if vis[s][sd:sd + n].shape[0] < proposal_min_frames:
            continue
        sl = slice(stride_offset, None, feat_stride)
        s_feat, o_feat = vis[s][sd:sd + n][sl], vis[o][od:od + n][sl]
        if s_feat.shape[0] < 2:
            continue
        sb, ob = boxes[s][sd:sd + n][sl], boxes[o][od:od + n][sl]
        parts = [s_feat, o_feat]
        if clip is not None:
            parts += [clip[s][sd:sd + n][sl], clip[o][od:od + n][sl]]
        parts += [so_box_features(sb, ob), entity_box_features(sb, w_, h_), entity_box_features(ob, w_, h_)]
        sids.append(s)
        oids.append(o)
        offs.append(stride_offset)
        feats.append(torch.cat(parts, dim=-1).permute(1, 0))
    if not sids:
        return {}
    return {"sids": torch.tensor(sids), "oids": torch.tensor(oids), "cat_ids": input_dict["cat_ids"],
            "cat_scores": input_dict["cat_scores"], "traj_durations": dur, "bboxes_list": boxes,
            "so_features_list": feats, "so_offset": torch.tensor(offs, dtype=torch.int64)}


def synth_raw_video(n_tracklets=8, video_len=120, min_len=12, max_len=100, n_visual=1024, n_clip=0, seed=5, wh=(640, 360)):
    """A synthetic `_prepare_test` output (dataloaders/vidvrd.py:459-550): tracklets with random durations, boxes that
    now and then stick out of the frame (the clamp must act), two same-category tracklets of which one shadows the
    other (the vIoU de-dup must drop it), per-tracklet visual (and CLIP) features, all ordered pairs that overlap."""
    g = torch.Generator().manual_seed(seed)
    W, H = wh
    durs, boxes, vis, clip = [], [], [], []
    for i in range(n_tracklets):
        L = int(torch.randint(min_len, max_len + 1, (1,), generator=g))
        st = int(torch.randint(0, video_len - L + 1, (1,), generator=g))
        durs.append([st, st + L])
        xy = torch.rand(L, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7]) - 8.0
        wh_ = torch.rand(L, 2, generator=g) * torch.tensor([W * 0.4, H * 0.4]) + 12.0
        boxes.append(torch.cat([xy, xy + wh_], dim=1))
        vis.append(torch.randn(L, n_visual, generator=g))
        if n_clip:
            clip.append(torch.randn(L, n_clip, generator=g))
    cat_ids = torch.arange(1, n_tracklets + 1)
    # tracklet 1 shadows tracklet 0: same category, inside its duration, nearly the same boxes
    a, e = durs[0]
    inner = [a + 2, e - 3]
    durs[1] = inner
    boxes[1] = boxes[0][2:2 + inner[1] - inner[0]].clone() + 0.5
    vis[1] = torch.randn(inner[1] - inner[0], n_visual, generator=g)
    if n_clip:
        clip[1] = torch.randn(inner[1] - inner[0], n_clip, generator=g)
    cat_ids[1] = cat_ids[0]
    sids, oids = [], []
    for s in range(n_tracklets):
        for o in range(n_tracklets):
            if s != o and min(durs[s][1], durs[o][1]) > max(durs[s][0], durs[o][0]):
                sids.append(s)
                oids.append(o)
    out = {"sids": torch.tensor(sids), "oids": torch.tensor(oids), "cat_ids": cat_ids,
           "cat_scores": torch.rand(n_tracklets, generator=g), "bboxes_list": boxes,
           "traj_durations": torch.tensor(durs, dtype=torch.int64), "visual_features_list": vis, "video_wh": (W, H)}
    if n_clip:
        out["clip_features_list"] = clip
    return out


def write_synth_pickles(directory, video_name="vid0", n_tracklets=6, video_len=60, n_visual=32, seed=9):
    """Write the two per-video files `_prepare_test` reads (dataloaders/vidvrd.py:466-467,510-511): <dir>/info/<name>.pkl
    with the trajectory proposals (inclusive end frames) and <dir>/feat/<name>.pkl with the per-frame RoI features.
    Returns (info path, features path)."""
    import os
    import pickle
    g = torch.Generator().manual_seed(seed)
    durs, boxes = [], []
    for _ in range(n_tracklets):
        L = int(torch.randint(3, video_len // 2, (1,), generator=g))
        st = int(torch.randint(0, video_len - L + 1, (1,), generator=g))
        durs.append([st, st + L - 1])                                        # inclusive end, as stored on disk
        xy = torch.rand(L, 2, generator=g) * 200.0
        boxes.append(torch.cat([xy, xy + torch.rand(L, 2, generator=g) * 80.0 + 5.0], dim=1))
    feats = {t: torch.randn(durs[t][1] - durs[t][0] + 1, n_visual, generator=g) for t in range(n_tracklets)}
    frames = {}
    for fid in range(video_len):
        tids = [t for t in range(n_tracklets) if durs[t][0] <= fid <= durs[t][1]]
        if tids:
            frames[fid] = {"frame_id": fid, "tids": tids, "visual_features": [feats[t][fid - durs[t][0]].numpy() for t in tids]}
    info = {"traj_proposal": {"num_proposals": n_tracklets, "cat_ids": torch.randint(1, 36, (n_tracklets,), generator=g),
                              "scores": torch.rand(n_tracklets, generator=g), "bboxes_list": boxes,
                              "traj_durations": torch.tensor(durs), "video_wh": (640, 360)}}
    os.makedirs(os.path.join(directory, "info"), exist_ok=True)
    os.makedirs(os.path.join(directory, "feat"), exist_ok=True)
    paths = (os.path.join(directory, "info", video_name + ".pkl"), os.path.join(directory, "feat", video_name + ".pkl"))
    for path, obj in zip(paths, (info, frames)):
        with open(path, "wb") as f:
            pickle.dump(obj, f)
    return paths


def write_synth_train_files(directory, video_name="vid0", n_tracks=5, video_len=90, n_visual=16, seed=21, wh=(320, 240)):
    """Write the two per-video files `_prepare_train` reads (dataloaders/vidvrd.py:175-186): <dir>/anno/<name>.json
    (VidVRD annotation: per-frame trajectories, subject/objects, relation instances) and <dir>/gtfeat/<name>.pkl (RoI
    features of the ground-truth boxes per 1-based frame id).  The video is built to hit the loader's cases: a trajectory
    with a gap (two intervals), relation instances of one (subject, object, predicate) that overlap in time (merged),
    several relations on one pair, boxes that stick out of the frame (clamped), a pair longer than a short max_seq_len.
    Returns (annotation dir, feature dir, entity name -> id, predicate name -> id)."""
    import json
    import os
    import pickle
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    W, H = wh
    cats = ["dog", "person", "ball", "bicycle", "sofa"]
    preds = ["chase", "watch", "ride", "next_to"]
    spans = []
    for t in range(n_tracks):
        a = int(torch.randint(0, video_len // 3, (1,), generator=g))
        e = int(torch.randint(2 * video_len // 3, video_len + 1, (1,), generator=g))
        spans.append([(a, e)])
    a, e = spans[1][0]
    spans[1] = [(a, a + (e - a) // 2 - 3), (a + (e - a) // 2 + 2, e)]          # trajectory 1 disappears for five frames
    tids = [3 * t + 2 for t in range(n_tracks)]                                # ids are not 0..n-1
    trajectories = [[] for _ in range(video_len)]
    frames = {}
    for t, tid in enumerate(tids):
        for (a, e) in spans[t]:
            for f in range(a, e):
                xy = torch.rand(2, generator=g) * torch.tensor([W * 0.8, H * 0.8]) - 6.0
                wh_ = torch.rand(2, generator=g) * torch.tensor([W * 0.35, H * 0.35]) + 10.0
                x0, y0, x1, y1 = [round(float(v), 2) for v in (xy[0], xy[1], xy[0] + wh_[0], xy[1] + wh_[1])]
                trajectories[f].append({"tid": tid, "bbox": {"xmin": x0, "ymin": y0, "xmax": x1, "ymax": y1}})
                rec = frames.setdefault(f + 1, {"frame_id": f + 1, "tids": [], "visual_features": []})
                rec["tids"].append(tid)
                rec["visual_features"].append(torch.randn(n_visual, generator=g).numpy())
    for rec in frames.values():
        rec["tids"] = np.asarray(rec["tids"])
        rec["visual_features"] = np.stack(rec["visual_features"], axis=0).astype(np.float32)

    def inst(s, o, p, b, e_):
        return {"subject_tid": tids[s], "object_tid": tids[o], "predicate": preds[p], "begin_fid": b, "end_fid": e_}
    lo, hi = max(spans[0][0][0], spans[2][0][0]), min(spans[0][0][1], spans[2][0][1])
    third = (hi - lo) // 3
    rel = [inst(0, 2, 0, lo + 1, lo + third), inst(0, 2, 0, lo + third - 2, lo + 2 * third),     # overlap -> one instance
           inst(0, 2, 1, lo + 2, hi - 1), inst(2, 0, 3, lo, lo + 6),
           inst(3, 4, 2, max(spans[3][0][0], spans[4][0][0]) + 1, min(spans[3][0][1], spans[4][0][1]) - 1)]
    b1 = spans[1][1]                                                                          # inside trajectory 1's second interval
    rel.append(inst(1, 0, 1, max(b1[0], spans[0][0][0]) + 1, min(b1[1], spans[0][0][1]) - 1))
    anno = {"video_id": video_name, "height": H, "width": W, "trajectories": trajectories,
            "subject/objects": [{"tid": tid, "category": cats[t % len(cats)]} for t, tid in enumerate(tids)],
            "relation_instances": rel}
    anno_dir, feat_dir = os.path.join(directory, "anno"), os.path.join(directory, "gtfeat")
    os.makedirs(anno_dir, exist_ok=True)
    os.makedirs(feat_dir, exist_ok=True)
    with open(os.path.join(anno_dir, video_name + ".json"), "w") as f:
        json.dump(anno, f)
    with open(os.path.join(feat_dir, video_name + ".pkl"), "wb") as f:
        pickle.dump(frames, f)
    return anno_dir, feat_dir, {c: i + 1 for i, c in enumerate(cats)}, {p_: i + 1 for i, p_ in enumerate(preds)}
