"""Evaluation glue: record conversion and the VidVRD-helper metrics restated in vrdone_amd/evaluate.py
(third-party, un-vendored: parity unpinned -- hand-worked cases and invariants)."""
import os
import sys

import numpy as np
import pytest

from vrdone_amd import evaluate as E


def _result():
    box = [[0.0, 0.0, 9.0, 9.0]]
    return {"triplets": [[1, 2, 0], [0, 1, 1]], "pred_durations": [[4, 7], [0, 1]], "triple_scores_avg": [0.75, 0.5],
            "so_trajs": [[box * 3, box * 3], [box, box]], "so_tids": [[0, 1], [1, 0]], "triple_scores": [[1, 1, 1]] * 2}


def test_convertor_builds_the_benchmark_records():
    conv = E.EvaluationFormatConvertor("VidOR", entity_id_to_name={0: "dog", 1: "child"}, pred_id_to_name={1: "chase", 2: "watch"})
    out = conv.to_eval_format_pr("0001_3598080384", _result())
    assert list(out) == ["3598080384"]                                # vidor names lose their folder prefix
    first, second = out["3598080384"]
    assert first == {"triplet": ["child", "watch", "dog"], "duration": (4, 7), "score": 0.75,
                     "sub_traj": [[0.0, 0.0, 9.0, 9.0]] * 3, "obj_traj": [[0.0, 0.0, 9.0, 9.0]] * 3}
    assert second["triplet"] == ["dog", "chase", "child"] and second["duration"] == (0, 1)
    assert E.EvaluationFormatConvertor("vidvrd", {0: "a"}, {0: "b"}).to_eval_format_pr("ILSVRC2015_train_00005015", None) == {
        "ILSVRC2015_train_00005015": []}
    with pytest.raises(NotImplementedError):
        E.EvaluationFormatConvertor("coco", {}, {})
    bad = _result()
    bad["so_trajs"][0][0] = bad["so_trajs"][0][0][:2]                  # trajectory shorter than the duration
    with pytest.raises(AssertionError):
        conv.to_eval_format_pr("0001_1", bad)


@pytest.mark.skipif(not os.path.isdir("/root/reference/dataloaders"), reason="needs the reference checkout's category tables")
def test_convertor_takes_its_tables_from_the_checkout():
    sys.path.insert(0, "/root/reference")
    try:
        conv = E.EvaluationFormatConvertor("vidvrd")
        assert len(conv.entity_id_to_name) == 36 and len(conv.pred_id_to_name) == 133        # incl. the background entries
        assert len(E.EvaluationFormatConvertor("vidor").entity_id_to_name) == 81
    finally:
        sys.path.remove("/root/reference")
        for m in [m for m in sys.modules if m == "dataloaders" or m.startswith("dataloaders.")]:
            del sys.modules[m]


def test_viou_hand_worked():
    a = [[0, 0, 9, 9]] * 4            # frames 0..3, 100 px each
    b = [[5, 0, 14, 9]] * 4           # frames 2..5, overlaps a in 5 columns on frames 2, 3
    assert E.viou(a, (0, 4), b, (2, 6)) == pytest.approx(2 * 50 / (400 + 400 - 100))
    assert E.viou(a, (0, 4), a, (0, 4)) == 1.0
    assert E.viou(a, (0, 4), b, (4, 8)) == 0.0                        # durations touch but do not overlap
    assert E.viou(a, (0, 4), [[20, 20, 29, 29]] * 4, (0, 4)) == 0.0   # same frames, disjoint boxes
    assert E.viou(b, (2, 6), a, (0, 4)) == E.viou(a, (0, 4), b, (2, 6))


def test_voc_ap_hand_worked():
    # hits at ranks 1 and 3 of 2 ground truths: envelope precision 1 up to recall 0.5, 2/3 up to recall 1
    prec, rec = np.array([1.0, 0.5, 2 / 3]), np.array([0.5, 0.5, 1.0])
    assert E.voc_ap(rec, prec) == pytest.approx(0.5 * 1.0 + 0.5 * (2 / 3))
    assert E.voc_ap(np.array([]), np.array([])) == 0.0


def _rel(triplet, dur, box, score=None):
    r = {"triplet": list(triplet), "duration": list(dur), "sub_traj": [box] * (dur[1] - dur[0]), "obj_traj": [box] * (dur[1] - dur[0])}
    if score is not None:
        r["score"] = score
    return r


def test_detection_matching_is_greedy_by_score_and_one_to_one():
    box, off = [0, 0, 9, 9], [3, 0, 12, 9]             # off: IoU with box = 70 / 130 = 0.538
    gt = [_rel("abc", (0, 4), box), _rel("abc", (10, 14), box), _rel("xyz", (0, 4), box)]
    preds = [_rel("abc", (0, 4), off, 0.9),           # hits gt 0 (vIoU 0.54)
             _rel("abc", (0, 4), box, 0.8),           # gt 0 already taken, gt 1 does not overlap in time -> miss
             _rel("abc", (10, 14), box, 0.7),         # hits gt 1
             _rel("xyz", (0, 4), [50, 50, 59, 59], 0.6),   # right triplet, wrong place -> miss
             _rel("qqq", (0, 4), box, 0.5)]           # unknown triplet -> miss
    prec, rec, hits = E.eval_detection_scores(gt, list(reversed(preds)), 0.5)       # input order must not matter
    assert hits.tolist() == [0.9, -np.inf, 0.7, -np.inf, -np.inf]
    np.testing.assert_allclose(rec, [1 / 3, 1 / 3, 2 / 3, 2 / 3, 2 / 3], rtol=1e-6)
    np.testing.assert_allclose(prec, [1, 1 / 2, 2 / 3, 2 / 4, 2 / 5], rtol=1e-6)
    # a higher threshold turns the 0.54 match into a miss, and then the second prediction takes gt 0
    _, _, hits = E.eval_detection_scores(gt, preds, 0.6)
    assert hits.tolist() == [-np.inf, 0.8, 0.7, -np.inf, -np.inf]


def test_tagging_counts_each_triplet_once():
    box = [0, 0, 9, 9]
    gt = [_rel("abc", (0, 4), box), _rel("abc", (5, 9), box), _rel("xyz", (0, 4), box)]
    preds = [_rel("qqq", (0, 1), box, 0.9), _rel("abc", (0, 1), box, 0.8), _rel("abc", (2, 3), box, 0.7), _rel("xyz", (0, 1), box, 0.1)]
    prec, rec, hits = E.eval_tagging_scores(gt, preds)
    assert hits.tolist() == [-np.inf, 0.8, 0.1]
    np.testing.assert_allclose(prec, [0, 1 / 2, 2 / 3], rtol=1e-6)
    np.testing.assert_allclose(rec, [0, 1 / 2, 1], rtol=1e-6)


def test_eval_relation_end_to_end(tmp_path):
    import json
    box = [0, 0, 9, 9]
    gt = {"v1": [_rel("abc", (0, 4), box), _rel("xyz", (2, 6), box)], "v2": [_rel("abc", (0, 2), box)], "v3": []}
    perfect = {v: [dict(r, score=1.0 - 0.1 * i) for i, r in enumerate(rels)] for v, rels in gt.items()}
    path = tmp_path / "gt.json"
    path.write_text(json.dumps(gt))
    cfg = {"prepare_gt_config": {"gt_relations_path": str(path)}, "inference_config": {"viou_th": 0.5}}
    res = E.eval_relation("vidvrd", prediction_results=perfect, config=cfg)
    assert set(res) == {"RelDet_mAP", "RelDet_AR@50", "RelDet_AR@100", "RelTag_AP@1", "RelTag_AP@5", "RelTag_AP@10"}
    assert res["RelDet_mAP"] == 1.0 and res["RelDet_AR@50"] == 1.0 and res["RelTag_AP@1"] == 1.0
    # predictions only for v1, one of its two relations: AP 0.5 on v1, 0 on v2 -> mAP 0.25; recall 1 of 3
    res = E.eval_relation("vidvrd", prediction_results={"v1": perfect["v1"][:1]}, config=cfg)
    assert res["RelDet_mAP"] == pytest.approx(0.25) and res["RelDet_AR@100"] == pytest.approx(1 / 3)
    assert res["RelTag_AP@1"] == pytest.approx(0.5) and res["RelTag_AP@5"] == pytest.approx(0.5)
    pred_file = tmp_path / "pred.json"
    pred_file.write_text(json.dumps(perfect))
    assert E.eval_relation("vidvrd", json_results_path=str(pred_file), config=cfg)["RelDet_mAP"] == 1.0
    with pytest.raises(FileNotFoundError):
        E.eval_relation("vidvrd", prediction_results=perfect, config={"prepare_gt_config": {"gt_relations_path": str(tmp_path / "none.json")},
                                                                     "inference_config": {"viou_th": 0.5}})
