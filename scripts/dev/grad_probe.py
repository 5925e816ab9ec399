import json, os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); os.chdir("/root/repo")
import numpy as np, torch
import tests.test_gpu_train as T
from golden_cases import TRAIN_SPECS, train_batch, replay_matching
from vrdone_amd import ops
from vrdone_amd.models.blocks import AffineDropPath
name = sys.argv[1] if len(sys.argv) > 1 else "vidor_x"
for mode in ("f32", "bf16x3"):
    ops.set_precision(mode)
    model, mc, _ = T.build(name)
    meta = json.load(open(os.path.join(T.GOLDEN, f"train_step_{name}.json")))
    g = np.load(os.path.join(T.GOLDEN, f"train_step_{name}.npz"))
    lens, _, _, data = train_batch(mc, T.c_in(mc), device="cuda", spec=TRAIN_SPECS[name])
    model.train()
    for mod in model.modules():
        if isinstance(mod, AffineDropPath): mod.drop_prob = 0.0
    replay_matching(model, meta["cases"]["nodrop"]["indices"])
    with torch.enable_grad():
        loss = model(data); loss["total_loss"].backward()
    print(mode, {k: (float(v), meta["cases"]["nodrop"]["losses"][k]) for k, v in list(loss.items())[:4]})
    stride = meta["sample_stride"]; stats = meta["cases"]["nodrop"]["grad_stats"]; biggest = max(s[2] for s in stats.values())
    errs = []
    for n, p in model.named_parameters():
        gg = p.grad.detach().float().cpu(); want = g[f"nodrop/{n}"]
        got = (gg if gg.numel() <= 2048 else gg.flatten()[::stride]).numpy()
        err = float(np.linalg.norm(got.astype(np.float64) - want)) / (float(np.linalg.norm(want)) + 1e-4 * biggest)
        errs.append((err, n))
    errs.sort()
    print(mode, "median", errs[len(errs)//2][0])
    bygroup = {}
    for e, n in errs:
        key = ".".join(n.split(".")[:3]); bygroup.setdefault(key, []).append(e)
    for k, v in sorted(bygroup.items()):
        print(f"   {k:50s} n={len(v):3d} median {np.median(v):.2e} max {max(v):.2e}")
