"""forward_test wall time on small, real-sized videos (vidvrd: pairs of <= 96 frames mostly): per-pair matrices and per-tracklet
features, row-space form on / off; and how many kernel launches a call issues."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, synth, _hip
from vrdone_amd.models.maskvrd import MaskVRD
from vrdone_amd.proposals import prepare_test_proposal

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
ic = configs.inference_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
model._config_eval(ic)
c_in = configs.input_channels(cfg)
with torch.no_grad():
    for n_trk, lo, hi in ((8, 10, 60), (16, 20, 90), (24, 20, 120), (46, 30, 96)):
        video = synth.synth_video(n_trk, c_in, lo, hi, seed=3, device=dev)
        raw = synth.synth_raw_video(n_trk, cfg["visual_dim"], lo, hi, seed=3)
        prop = prepare_test_proposal(raw, ic["feat_stride"], 0, 2, dev)
        for name, data in (("matrices ", video), ("tracklets", prop)):
            for rs in (True, False, "auto"):      # True / False: the form forced; auto: the default policy (MaskVRD.ROWS_MIN_PAIRS), measured last
                for attr in ("row_space", "ROWS_MIN_PAIRS"):
                    if attr in model.__dict__:
                        delattr(model, attr)
                if rs != "auto":
                    model.row_space = rs
                    model.ROWS_MIN_PAIRS = 0
                ts = []
                for it in range(6):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    res = model(data)
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                _hip.prof_enable(True); _hip.prof_reset()
                model(data); torch.cuda.synchronize()
                _hip.prof_enable(False)
                pr = _hip.prof_read()
                print(f"{len(data['sids']):5d} pairs ({lo}-{hi} frames) {name} row space {str(rs):5s}: {1e3 * sorted(ts[1:])[2]:7.2f} ms per call, "
                      f"{sum(v['launches'] for v in pr.values())} library launches, {sum(v['ms'] for v in pr.values()):.2f} ms of kernels", flush=True)
