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
