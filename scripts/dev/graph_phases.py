"""Dev aid: wall time of the phases of a training step (24-pair batch), eager against HIP graphs."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "scripts"))
from train_step import synthetic_batch, param_groups
from vrdone_amd import configs, synth, train_graph
from vrdone_amd.models.maskvrd import MaskVRD
cfg = configs.model_config("vidvrd")
model = synth.load_synthetic_weights(MaskVRD(cfg, device="cuda")).cuda().train()
data = synthetic_batch(cfg, configs.input_channels(cfg), "cuda", seed=0)
opt = torch.optim.AdamW(param_groups(model, 0.05), lr=1e-5)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for graphs in (False, True):
    model.enable_training_graphs(graphs)
    acc = {}
    for it in range(8):
        t = [sync()]
        x, m = model._train_batch(data["so_features_list"]); t.append(sync())
        if graphs: pred = train_graph.mask_vrd(model, x, m)
        else: pred = model._mask_vrd(x, m, with_aux=model.deep_supervision)
        t.append(sync())
        loss = model.criterion(pred, data)["total_loss"]; t.append(sync())
        opt.zero_grad(set_to_none=True); loss.backward(); t.append(sync())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); opt.step(); t.append(sync())
        if it >= 3:
            for name, a, b in zip(("batch", "network fwd", "criterion", "backward", "clip+adamw"), t[:-1], t[1:]):
                acc.setdefault(name, []).append(1e3 * (b - a))
    print("graphs" if graphs else "eager ", {k: round(sorted(v)[len(v) // 2], 2) for k, v in acc.items()},
          "sum", round(sum(sorted(v)[len(v) // 2] for v in acc.values()), 1))
