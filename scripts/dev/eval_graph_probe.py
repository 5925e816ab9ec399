"""Lab aid: what one bucket of forward_test costs eagerly and as a recorded HIP graph (vrdone_amd/eval_graph.py):
wall time per call of MaskVRD._bucket_candidates for a few (T, pairs), eager against graph.replay(), and the graphs' node counts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vrdone_amd import configs, eval_graph, ops, synth
from vrdone_amd.models.maskvrd import MaskVRD

dev = torch.device("cuda:0")
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
model._config_eval(configs.inference_config("vidvrd"))
c_in = configs.input_channels(cfg)
k = model.topk
with torch.no_grad():
    for T, n in ((32, 16), (64, 48), (96, 128), (128, 256)):
        g = torch.Generator(device=dev).manual_seed(T + n)
        mats = [torch.randn(max(2, T - 3 - i % 7), c_in, device=dev, generator=g) for i in range(n)]
        table = torch.tensor([m.data_ptr() for m in mats], dtype=torch.int64, device=dev)
        lens = torch.tensor([m.shape[0] for m in mats], dtype=torch.int32, device=dev)

        def timed(fn, reps=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / reps
        eager = timed(lambda: model._bucket_candidates(table, lens, T, k))
        rec = eval_graph._Recording(model, T, eval_graph.pad_size(n), k, c_in)
        replay = timed(lambda: rec(table, lens))
        bare = timed(lambda: rec.graph.replay())
        print(f"T {T:4d}  {n:4d} pairs: eager {eager:6.2f} ms per call, recorded {replay:6.2f} ms (graph.replay() alone {bare:6.2f})", flush=True)

# where a bucket's launches go (library families; the tensor ops between them are not counted)
from vrdone_amd import _hip
with torch.no_grad():
    T, n = 64, 48
    g = torch.Generator(device=dev).manual_seed(1)
    mats = [torch.randn(T - 3 - i % 7, c_in, device=dev, generator=g) for i in range(n)]
    table = torch.tensor([m.data_ptr() for m in mats], dtype=torch.int64, device=dev)
    lens = torch.tensor([m.shape[0] for m in mats], dtype=torch.int32, device=dev)
    model._bucket_candidates(table, lens, T, k)
    torch.cuda.synchronize()
    _hip.prof_enable(True); _hip.prof_reset()
    model._bucket_candidates(table, lens, T, k)
    torch.cuda.synchronize()
    _hip.prof_enable(False)
    pr = _hip.prof_read()
    print("launches per family (48 pairs x 64 frames):", {f: (v["launches"], round(v["ms"], 3)) for f, v in sorted(pr.items(), key=lambda kv: -kv[1]["launches"]) if v["launches"]})
    print("total", sum(v["launches"] for v in pr.values()), "launches,", round(sum(v["ms"] for v in pr.values()), 2), "ms")
