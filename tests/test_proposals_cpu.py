"""Host half of the device-side pair construction (vrdone_amd/proposals.py): box clamp, vIoU
de-dup, pair list, row offsets and lengths against the oracle restatement of the reference dataloader's `_test_getitem`
(itself pinned bit for bit by tests/golden/proposal.npz).  CPU only: the tables are built on the host and the tensors
stay on 'cpu' here; the gather kernel is tested in tests/test_gpu_model.py."""
import numpy as np
import pytest
import torch

from golden_cases import PROPOSAL_CASES
from oracle import proposal as P


@pytest.mark.parametrize("name", list(PROPOSAL_CASES))
def test_pair_tables_match_the_reference_dataloader(name, golden_dir):
    from vrdone_amd.proposals import prepare_test_proposal
    vid_kw, dl_kw = PROPOSAL_CASES[name]
    raw = P.synth_raw_video(**vid_kw)
    got = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], "cpu")
    g = np.load(f"{golden_dir}/proposal.npz")
    assert got["sids"].tolist() == g[f"{name}/sids"].tolist() and got["oids"].tolist() == g[f"{name}/oids"].tolist()
    assert got["so_offset"].tolist() == g[f"{name}/so_offset"].tolist()
    src = got["pair_source"]
    assert src.lens == g[f"{name}/lens"].tolist()
    np.testing.assert_array_equal(torch.cat(got["bboxes_list"]).numpy(), g[f"{name}/boxes_clamped"])
    # the row tables address exactly the frames the reference slices: re-gather on the host and compare the visual sums
    want = P.test_getitem(raw, **dl_kw)["so_features_list"]
    V = src.n_visual
    for p, f in enumerate(want):
        rows_s = src.s_row[p] + torch.arange(src.lens[p]) * src.stride
        rows_o = src.o_row[p] + torch.arange(src.lens[p]) * src.stride
        assert torch.equal(src.vis[rows_s], f[:V].T) and torch.equal(src.vis[rows_o], f[V:2 * V].T)


def test_no_pair_survives():
    from vrdone_amd.proposals import prepare_test_proposal
    raw = P.synth_raw_video(n_tracklets=3, video_len=40, min_len=4, max_len=6, seed=1)
    raw["sids"], raw["oids"] = raw["sids"][:0], raw["oids"][:0]
    assert prepare_test_proposal(raw, 1, 0, 2, "cpu") == {}


def test_load_test_video_reads_the_reference_pickle_layout(tmp_path, golden_dir):
    """vrdone_amd.proposals.load_test_video (restating `_prepare_test`, dataloaders/vidvrd.py:459-550) on the two per-video
    pickles: pair order, [start, end) durations and per-tracklet feature stacks against the reference's own `_prepare_test`
    run on the same files (tests/golden/proposal.npz, keys pickles/*), and against the reference itself when it is here."""
    import os
    from vrdone_amd.proposals import load_test_video
    info, feat = P.write_synth_pickles(str(tmp_path))
    got = load_test_video(info, feat)
    g = np.load(f"{golden_dir}/proposal.npz")
    assert got["sids"].tolist() == g["pickles/sids"].tolist() and got["oids"].tolist() == g["pickles/oids"].tolist()
    np.testing.assert_array_equal(got["traj_durations"].numpy(), g["pickles/traj_durations"])
    np.testing.assert_array_equal(torch.cat(got["visual_features_list"]).numpy(), g["pickles/visual_features"])
    np.testing.assert_array_equal(torch.cat(got["bboxes_list"]).numpy(), g["pickles/bboxes"])
    assert tuple(got["video_wh"]) == (640, 360)
    if os.path.isdir("/root/reference/dataloaders"):
        import subprocess, sys, json
        code = (f"import sys, json; sys.path.insert(0, '/root/reference')\n"
                f"from dataloaders.vidvrd import VidVRD\n"
                f"ds = object.__new__(VidVRD); ds.split = 'test'; ds.info_dir = {os.path.dirname(info)!r}; ds.test_boxfeatures_dir = {os.path.dirname(feat)!r}\n"
                f"out = ds._prepare_test('vid0')\n"
                f"print(json.dumps([out['sids'].tolist(), out['oids'].tolist(), out['traj_durations'].tolist(), float(sum(v.double().sum() for v in out['visual_features_list']))]))\n")
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-1500:]
        sids, oids, durs, fsum = json.loads(res.stdout.strip().splitlines()[-1])
        assert got["sids"].tolist() == sids and got["oids"].tolist() == oids and got["traj_durations"].tolist() == durs
        assert abs(float(sum(v.double().sum() for v in got["visual_features_list"])) - fsum) < 1e-9


def test_plain_tensor_form_passes_the_reference_eval_loop():
    """prepare_test_proposal(device=None) returns only value types the reference's `utils.dict_to_device`
    (utils/misc.py:98-112) accepts -- tensors, ints, strings, lists of tensors -- and PairSource.from_fields puts the same
    source back together."""
    from vrdone_amd.proposals import PairSource, prepare_test_proposal
    vid_kw, dl_kw = PROPOSAL_CASES["strided"]
    raw = P.synth_raw_video(**dict(vid_kw, n_clip=16))
    flat = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], None)
    obj = prepare_test_proposal(raw, dl_kw["feat_stride"], dl_kw["stride_offset"], dl_kw["proposal_min_frames"], "cpu")
    assert "pair_source" not in flat and set(PairSource.FIELDS) <= set(flat) and "tracklet_clip" in flat
    for k, v in flat.items():
        assert isinstance(v, (torch.Tensor, int, str)) or (isinstance(v, list) and all(isinstance(t, torch.Tensor) for t in v)), k
    a, b = PairSource.from_fields(flat, "cpu"), obj["pair_source"]
    assert a.lens == b.lens and a.stride == b.stride and a.wh == b.wh and (a.first_row == b.first_row).all()
    for name in ("vis", "clip", "boxes", "s_row", "o_row"):
        assert torch.equal(getattr(a, name), getattr(b, name))
    for k in ("sids", "oids", "so_offset"):
        assert torch.equal(flat[k], obj[k])


# --------------------------------------------------------------------------------------------------------------------
# training side: tests/golden/train_data.* = the reference dataloader's own `_prepare_train`,
# `_train_getitem` (seeded `random`) and `apply_policy` on the synthetic files of oracle.proposal.write_synth_train_files
# (scripts/make_golden_r2.py --only-train-data)
# --------------------------------------------------------------------------------------------------------------------
TRAIN_DATA_CASES = {   # name -> (feat_stride, max_seq_len, cut_max_preds, proposal_max_preds, pair_duration, random seed)
    "stride1": (1, 96, False, 0, None, 3),
    "stride1_crop": (1, 24, False, 0, None, 4),
    "stride4": (4, 96, False, 0, None, 5),
    "cut": (1, 96, True, 1, (1, 3), 6),
}


@pytest.fixture(scope="module")
def train_video(tmp_path_factory):
    from vrdone_amd.proposals import load_train_video
    tmp = tmp_path_factory.mktemp("train")
    anno_dir, feat_dir, ent, pred = P.write_synth_train_files(str(tmp))
    return load_train_video(f"{anno_dir}/vid0.json", f"{feat_dir}/vid0.pkl", ent, pred)


def test_train_cache_entry_matches_the_reference_dataloader(train_video, golden_dir):
    import json
    g = json.load(open(f"{golden_dir}/train_data.json"))
    arrs = np.load(f"{golden_dir}/train_data.npz")
    v = train_video
    assert v["relation_keys"] == g["relation_keys"]                      # incl. the set's iteration order (pair_duration indexes it)
    assert [[list(k), r] for k, r in v["relation_merged"].items()] == g["relation_merged"]
    assert {str(k): iv for k, iv in v["traj_intervals"].items()} == g["traj_intervals"]
    assert {str(k): c for k, c in v["entity_classes"].items()} == g["entity_classes"]
    assert list(v["video_hw"]) == g["video_hw"]
    assert any(len(iv) > 1 for iv in v["traj_intervals"].values())       # the gapped trajectory became two intervals
    assert any(len(r) > 1 for r in v["relation_merged"].values())        # several relations on one pair
    np.testing.assert_array_equal(torch.cat([t for k in sorted(v["visual_features"]) for t in v["visual_features"][k]]).numpy(), arrs["visual"])
    np.testing.assert_array_equal(torch.cat([t for k in sorted(v["entity_bboxes"]) for t in v["entity_bboxes"][k]]).numpy(), arrs["boxes"])


@pytest.mark.parametrize("name", list(TRAIN_DATA_CASES))
def test_train_samples_match_the_reference_dataloader(name, train_video, golden_dir):
    """Same draws from `random`, same crops: features, masks, predicates and segments equal the reference's bit for bit."""
    import json
    import random
    from vrdone_amd.proposals import train_getitem
    stride, max_len, cut, max_preds, dur, seed = TRAIN_DATA_CASES[name]
    g = json.load(open(f"{golden_dir}/train_data.json"))["samples"][name]
    arrs = np.load(f"{golden_dir}/train_data.npz")
    random.seed(seed)
    got = train_getitem(train_video, stride, max_len, cut, max_preds, dur)
    assert len(got.get("so_features_list", [])) == g["n"] > 0
    assert [int(f.shape[1]) for f in got["so_features_list"]] == g["lens"]
    assert [p.tolist() for p in got["preds_list"]] == g["preds"]
    assert [s.tolist() for s in got["segs_list"]] == g["segs"]
    for i, (f, m) in enumerate(zip(got["so_features_list"], got["masks_list"])):
        np.testing.assert_array_equal(f.numpy(), arrs[f"{name}/feat{i}"])
        np.testing.assert_array_equal(m.numpy(), arrs[f"{name}/mask{i}"])


def test_train_policy_matches_the_reference(golden_dir):
    import json
    from vrdone_amd.proposals import train_policy
    g = json.load(open(f"{golden_dir}/train_data.json"))["policy"]
    got = train_policy(g["video_num_pairs"], g["num_pairs"])
    assert [[[n, list(r)] for n, r in step] for step in got] == g["policy"]
    assert all(sum(r[1] - r[0] for _, r in step) <= g["num_pairs"] for step in got)


def test_train_samples_with_clip_features_match_the_vidor_loader(train_video, golden_dir):
    """The VidOR loader's `_train_getitem` with CLIP features (dataloaders/vidor.py:335-478) on the same cache entry plus
    per-interval CLIP rows: [subject visual | object visual | subject CLIP | object CLIP | 21 box channels]."""
    import copy
    import json
    import random
    from vrdone_amd.proposals import train_getitem
    g = json.load(open(f"{golden_dir}/train_data.json"))["samples"]["vidor_clip"]
    arrs = np.load(f"{golden_dir}/train_data.npz")
    video = copy.deepcopy(train_video)
    flat, at, clip = torch.from_numpy(arrs["clip"]), 0, {}
    for k in sorted(video["visual_features"]):
        clip[k] = []
        for t in video["visual_features"][k]:
            clip[k].append(flat[at:at + t.shape[0]])
            at += t.shape[0]
    video["clip_features"] = clip
    random.seed(8)
    got = train_getitem(video, 4, 96)
    assert [int(f.shape[1]) for f in got["so_features_list"]] == g["lens"] and [p.tolist() for p in got["preds_list"]] == g["preds"]
    for i, (f, m) in enumerate(zip(got["so_features_list"], got["masks_list"])):
        np.testing.assert_array_equal(f.numpy(), arrs[f"vidor_clip/feat{i}"])
        np.testing.assert_array_equal(m.numpy(), arrs[f"vidor_clip/mask{i}"])
