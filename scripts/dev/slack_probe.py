import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.chdir("/root/repo")
import numpy as np, torch
import tests.test_gpu_model as T
from golden_cases import SLICES, VIDOR_X
from vrdone_amd import ops
def diffs(res, ref):
    n = len(ref["triplets"])
    same = sum(a == b for a, b in zip(res["triplets"], ref["triplets"]))
    key = lambda r, i: (tuple(r["triplets"][i]), tuple(r["so_tids"][i]), tuple(r["pred_durations"][i]))
    got, want = {key(res, i) for i in range(n)}, {key(ref, i) for i in range(n)}
    dig = {(len(t[0]), round(float(np.sum(np.asarray(t, dtype=np.float64))), 3)) for t in res["so_trajs"]}
    wdig = {(int(a), round(b, 3)) for a, b in ref["so_trajs_digest"]}
    sc = float(np.max(np.abs(np.asarray(res["triple_scores_avg"]) - np.asarray(ref["triple_scores_avg"]))))
    return n - same, len(want) - len(got & want), len(wdig) - len(dig & wdig), sc
for mode in ("f32", "bf16x3"):
    ops.set_precision(mode)
    for name, cfg, fn, kw in (("vidvrd", "vidvrd", "forward_test_vidvrd.json", dict(n_tracklets=6, min_len=20, max_len=130, seed=4321)),
                              ("slices", "vidvrd", "forward_test_vidvrd_slices.json", SLICES), ("vidor_x", "vidor_x", "forward_test_vidor_x.json", VIDOR_X)):
        model, mc, ic, _ = T.get_model(cfg)
        ref = json.load(open(os.path.join(T.GOLDEN, fn)))
        if name == "vidvrd":
            data = T.synth_proposal(6, T.c_in(mc), 20, 130, seed=4321)
        else:
            data = T.synth_proposal(c_in=T.c_in(mc), **kw)
        res = model(T._on_device(data))
        print(mode, name, "rank diffs, record diffs, digest diffs, max score err:", diffs(res, ref))
