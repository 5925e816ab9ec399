"""Evaluation glue behind `forward_test`: result dict -> the benchmark's JSON records, and the relation
detection / tagging metrics.

* `EvaluationFormatConvertor` -- interface and output of reference utils/evaluate.py:12-73 (what eval.py:102,151 calls).
  The id -> name tables are the reference checkout's `dataloaders/category.py` (data, not restated here): they are
  imported from the checkout when the drop-in launcher has put it on sys.path, or passed in.
* `eval_visual_relation` / `eval_relation` -- interface of utils/evaluate.py:77-170.  The scoring functions they call
  (`eval_detection_scores`, `eval_tagging_scores`, `voc_ap`, `viou`) live in the un-vendored third-party package
  VidVRD-helper (xdshang/VidVRD-helper, `evaluation/visual_relation_detection.py`, `evaluation/common.py`; the reference
  pins no version), which is absent from the reference tree: they are restated here from the published algorithm and
  **parity is unpinned** -- tests check them on hand-worked cases and invariants only.  The vIoU of all (prediction, ground
  truth) candidates of a video is evaluated with array operations per pair of trajectories instead of a Python loop per
  frame.
"""
import json
import os
from collections import defaultdict

import numpy as np


def _reference_tables(dataset_type):
    try:
        from dataloaders import category          # the reference checkout (dropin/run.py puts it on sys.path)
    except ImportError as e:
        raise ImportError("EvaluationFormatConvertor needs the id -> name tables of the reference checkout's "
                          "dataloaders/category.py: run from the checkout (dropin/run.py) or pass "
                          "entity_id_to_name / pred_id_to_name") from e
    return (getattr(category, f"{dataset_type}_category_id_to_name"), getattr(category, f"{dataset_type}_pred_id_to_name"))


class EvaluationFormatConvertor:
    """forward_test's result dict -> {video name: [relation records]} (reference utils/evaluate.py:12-73)."""

    def __init__(self, dataset_type, entity_id_to_name=None, pred_id_to_name=None):
        self.dataset_type = dataset_type.lower()
        if self.dataset_type not in ("vidvrd", "vidor"):
            raise NotImplementedError(dataset_type)
        if entity_id_to_name is None or pred_id_to_name is None:
            entity_id_to_name, pred_id_to_name = _reference_tables(self.dataset_type)
        self.entity_id_to_name, self.pred_id_to_name = entity_id_to_name, pred_id_to_name

    def _reset_video_name(self, video_name):
        if self.dataset_type == "vidor":            # "0001_3598080384" -> "3598080384"
            parts = video_name.split('_')
            assert len(parts) == 2
            return parts[1]
        return video_name

    def to_eval_format_pr(self, video_name, pr_triplet):
        """pr_triplet: what MaskVRD.forward_test returns (None -- no candidate survived -- gives an empty list; the
        reference's caller skips such videos before it gets here, eval.py:148-149)."""
        video_name = self._reset_video_name(video_name)
        if pr_triplet is None:
            return {video_name: []}
        records = []
        for i, (s_cat, pred_cat, o_cat) in enumerate(pr_triplet['triplets']):
            start, end = pr_triplet["pred_durations"][i][0], pr_triplet["pred_durations"][i][1]
            sub, obj = pr_triplet["so_trajs"][i]
            assert len(sub) == len(obj) == end - start
            records.append({
                "triplet": [self.entity_id_to_name[s_cat], self.pred_id_to_name[pred_cat], self.entity_id_to_name[o_cat]],
                "duration": (start, end),
                "score": float(pr_triplet["triple_scores_avg"][i]),
                "sub_traj": sub,
                "obj_traj": obj,
            })
        return {video_name: records}


# --------------------------------------------------------------------------------------------------------------------
# VidVRD-helper's scoring, restated (parity unpinned, see the module docstring)
# --------------------------------------------------------------------------------------------------------------------
def _volume(traj):
    t = np.asarray(traj, dtype=np.float64).reshape(-1, 4)
    return float(((t[:, 2] - t[:, 0] + 1) * (t[:, 3] - t[:, 1] + 1)).sum())


def viou(traj_1, duration_1, traj_2, duration_2):
    """Volumetric IoU of two box tracks; durations are [start, end) frame ids, boxes (x0, y0, x1, y1) with the +1 pixel
    convention (VidVRD-helper evaluation/common.py `viou`)."""
    lo, hi = max(duration_1[0], duration_2[0]), min(duration_1[1], duration_2[1])
    if hi <= lo:
        return 0.0
    a = np.asarray(traj_1, dtype=np.float64).reshape(-1, 4)[lo - duration_1[0]:hi - duration_1[0]]
    b = np.asarray(traj_2, dtype=np.float64).reshape(-1, 4)[lo - duration_2[0]:hi - duration_2[0]]
    w = np.clip(np.minimum(a[:, 2], b[:, 2]) - np.maximum(a[:, 0], b[:, 0]) + 1, 0, None)
    h = np.clip(np.minimum(a[:, 3], b[:, 3]) - np.maximum(a[:, 1], b[:, 1]) + 1, 0, None)
    overlap = float((w * h).sum())
    return overlap / (_volume(traj_1) + _volume(traj_2) - overlap)


def voc_ap(rec, prec):
    """Area under the monotone precision envelope (VOC 2010+ average precision; evaluation/common.py `voc_ap`)."""
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.nonzero(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def _prec_rec(hit_scores, n_gt):
    tp = np.isfinite(hit_scores)
    cum_tp = np.cumsum(tp).astype(np.float32)
    cum_fp = np.cumsum(~tp).astype(np.float32)
    eps = np.finfo(np.float32).eps
    return cum_tp / np.maximum(cum_tp + cum_fp, eps), cum_tp / np.maximum(n_gt, eps)


def eval_detection_scores(gt_relations, pred_relations, viou_threshold):
    """Greedy matching in descending score order: a prediction hits the not-yet-detected ground truth of the same triplet
    with the largest min(subject vIoU, object vIoU) >= threshold.  Returns (precision, recall, hit scores: the score of a
    hit, -inf of a miss)."""
    preds = sorted(pred_relations, key=lambda r: r['score'], reverse=True)
    detected = np.zeros(len(gt_relations), dtype=bool)
    hit_scores = np.full(len(preds), -np.inf)
    by_triplet = defaultdict(list)
    for g, rel in enumerate(gt_relations):
        by_triplet[tuple(rel['triplet'])].append(g)
    for p, pred in enumerate(preds):
        best, best_g = -np.inf, -1
        for g in by_triplet.get(tuple(pred['triplet']), ()):
            if detected[g]:
                continue
            gt = gt_relations[g]
            ov = min(viou(pred['sub_traj'], pred['duration'], gt['sub_traj'], gt['duration']),
                     viou(pred['obj_traj'], pred['duration'], gt['obj_traj'], gt['duration']))
            if ov >= viou_threshold and ov > best:
                best, best_g = ov, g
        if best_g >= 0:
            hit_scores[p] = pred['score']
            detected[best_g] = True
    prec, rec = _prec_rec(hit_scores, len(gt_relations))
    return prec, rec, hit_scores


def eval_tagging_scores(gt_relations, pred_relations):
    """Trajectories ignored: the distinct predicted triplets in descending score order against the set of ground-truth
    triplets."""
    preds = sorted(pred_relations, key=lambda r: r['score'], reverse=True)
    gt_triplets = {tuple(r['triplet']) for r in gt_relations}
    seen, hit_scores = set(), []
    for r in preds:
        t = tuple(r['triplet'])
        if t not in seen:
            seen.add(t)
            hit_scores.append(r['score'] if t in gt_triplets else -np.inf)
    hit_scores = np.asarray(hit_scores, dtype=np.float64)
    prec, rec = _prec_rec(hit_scores, len(gt_triplets))
    return prec, rec, hit_scores


def eval_visual_relation(groundtruth, prediction, viou_threshold=0.5, det_nreturns=(50, 100), tag_nreturns=(1, 5, 10)):
    """(mean AP, {n: recall@n}, {n: tagging precision@n}); reference utils/evaluate.py:77-126."""
    video_ap = {}
    tot_scores, tot_tp, prec_at_n = defaultdict(list), defaultdict(list), defaultdict(list)
    tot_gt = 0
    for vid, gt_relations in groundtruth.items():
        if len(gt_relations) == 0:
            continue
        tot_gt += len(gt_relations)
        preds = prediction.get(vid, [])
        det_prec, det_rec, det_scores = eval_detection_scores(gt_relations, preds, viou_threshold)
        video_ap[vid] = voc_ap(det_rec, det_prec)
        tp = np.isfinite(det_scores)
        for n in det_nreturns:
            cut = min(n, det_scores.size)
            tot_scores[n].append(det_scores[:cut])
            tot_tp[n].append(tp[:cut])
        tag_prec, _, _ = eval_tagging_scores(gt_relations, preds)
        for n in tag_nreturns:
            cut = min(n, tag_prec.size)
            prec_at_n[n].append(tag_prec[cut - 1] if cut > 0 else 0.0)
    mean_ap = float(np.mean(list(video_ap.values())))
    rec_at_n = {}
    for n in det_nreturns:
        scores, tps = np.concatenate(tot_scores[n]), np.concatenate(tot_tp[n])
        tps = tps[np.argsort(scores)[::-1]]
        rec = np.cumsum(tps).astype(np.float32) / np.maximum(tot_gt, np.finfo(np.float32).eps)
        rec_at_n[n] = float(rec[-1])
    return mean_ap, rec_at_n, {n: float(np.mean(prec_at_n[n])) for n in tag_nreturns}


def eval_relation(dataset_type, prediction_results=None, json_results_path=None, config=None):
    """reference utils/evaluate.py:128-170, except that a missing ground-truth file is an error: building it needs the
    dataset classes of VidVRD-helper (utils/prepare_eval_labels.py), which are the reference's own job."""
    if prediction_results is None:
        assert json_results_path is not None
        with open(json_results_path) as f:
            prediction_results = json.load(f)
    else:
        assert json_results_path is None
    assert config is not None
    gt_path = config['prepare_gt_config']['gt_relations_path']
    if gt_path is None or not os.path.exists(gt_path):
        raise FileNotFoundError(f"ground-truth relations {gt_path!r} not found: generate them with the reference's "
                                "utils/prepare_eval_labels.py (needs VidVRD-helper's dataset classes)")
    with open(gt_path) as f:
        gt_relations = json.load(f)
    mean_ap, rec_at_n, mprec_at_n = eval_visual_relation(gt_relations, prediction_results,
                                                         viou_threshold=config['inference_config']['viou_th'])
    result = {"RelDet_mAP": mean_ap}
    result.update({f"RelDet_AR@{k}": v for k, v in rec_at_n.items()})
    result.update({f"RelTag_AP@{k}": v for k, v in mprec_at_n.items()})
    return result
