#!/usr/bin/env python3
"""A training loop body on the HIP path: own code mirroring the reference's train.py:176-191 (forward -> zero_grad ->
backward -> clip_grad_norm_ -> AdamW step -> EMA update) with its optimizer recipe (utils/train_utils.py:35-95: AdamW,
no weight decay on biases, LayerNorm / scale parameters and embeddings; configs/vidvrd.yaml training_config), on the
24-pair synthetic batch of BASELINE config 3 (6 videos x 4 pairs, T = max_seq_len = 96).

    python scripts/train_step.py --steps 5          # on the GPU box
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def param_groups(model, weight_decay):
    """decay: conv / linear weights; no decay: biases, LayerNorm affine, drop-path scales, embeddings (the split of the
    reference's build_optimizer)."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        is_ln = p.dim() == 3 and p.shape[0] == 1 and p.shape[2] == 1
        (no_decay if (leaf in ("bias", "scale") or is_ln or "query_embed" in name) else decay).append(p)
    return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


def synthetic_batch(cfg, c_in, device, n_pairs=24, seed=0):
    from vrdone_amd import synth
    T = cfg["max_seq_len"]
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(8, T + 1, (n_pairs,), generator=g).tolist()
    x, _ = synth.synth_pairs(n_pairs, c_in, T, lens, seed=seed + 1)
    preds, masks, segs = [], [], []
    for L in lens:
        n = int(torch.randint(1, 4, (1,), generator=g))
        a = torch.randint(0, max(L - 2, 1), (n,), generator=g)
        b = torch.minimum(a + 1 + torch.randint(1, L, (n,), generator=g), torch.tensor(L))
        m = torch.zeros(n, T)
        for r in range(n):
            m[r, a[r]:b[r]] = 1
        preds.append(torch.randint(1, cfg["num_classes"] + 1, (n,), generator=g))
        masks.append(m)
        segs.append(torch.stack([a, b], dim=1))
    to = lambda ts: [t.to(device) for t in ts]      # noqa: E731
    return {"so_features_list": to([x[i, :, :n].contiguous() for i, n in enumerate(lens)]),
            "preds_list": to(preds), "masks_list": to(masks), "segs_list": to(segs)}


def run(steps=3, seed=0, device="cuda", lr=1e-4, weight_decay=0.05, clip=1.0, ema_decay=0.999, verbose=True, drop_path=True,
        graphs=False, config="vidvrd", n_pairs=24, profile=False):
    from vrdone_amd import _hip, configs, synth
    from vrdone_amd.models.maskvrd import MaskVRD
    cfg = configs.model_config(config)
    torch.manual_seed(seed)
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=device)).to(device).train()
    if not drop_path:
        from vrdone_amd.models.blocks import AffineDropPath
        for mod in model.modules():
            if isinstance(mod, AffineDropPath):
                mod.drop_prob = 0.0
    if graphs:
        model.enable_training_graphs()        # forward + backward of the network as two HIP-graph replays (train_graph.py)
    from vrdone_amd.ema import ModelEma
    ema = ModelEma(model, decay=ema_decay)                                # one-launch EMA (vrd_ema_update), same values
    opt = torch.optim.AdamW(param_groups(model, weight_decay), lr=lr)
    data = synthetic_batch(cfg, configs.input_channels(cfg), device, n_pairs=n_pairs, seed=seed)
    start = [p.detach().clone() for p in model.parameters()]
    log = {"total_loss": [], "step_ms": [], "params_without_grad": [], "nonfinite_grads": []}
    for step in range(steps):
        if profile and step == steps - 1:                                 # per-kernel-family time of the last step (HIP events)
            _hip.prof_enable(True)
            _hip.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss_dict = model(data)                                           # train.py:182
        opt.zero_grad(set_to_none=True)
        loss_dict["total_loss"].backward()                                # train.py:186
        named = list(model.named_parameters())
        log["params_without_grad"] += [name for name, p in named if p.grad is None]
        with_grad = [(name, p.grad) for name, p in named if p.grad is not None]
        norms = torch.stack(torch._foreach_norm([g for _, g in with_grad]))       # one multi-tensor launch, one sync
        log["nonfinite_grads"] += [with_grad[i][0] for i in torch.nonzero(~torch.isfinite(norms)).flatten().tolist()]
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)     # train.py:187-188
        opt.step()
        ema.update(model)                                                 # train.py:194 (ModelEma.update, utils/train_utils.py:21-29)
        torch.cuda.synchronize()
        log["step_ms"].append(1e3 * (time.perf_counter() - t0))
        log["total_loss"].append(float(loss_dict["total_loss"].detach()))
        if verbose:
            print(f"step {step}: total_loss {log['total_loss'][-1]:.4f}  ({log['step_ms'][-1]:.1f} ms)", flush=True)
    if profile:
        prof = _hip.prof_read()
        _hip.prof_enable(False)
        log["kernel_ms_last_step"] = {k: round(v["ms"], 3) for k, v in prof.items() if v["launches"]}
        log["kernel_launches_last_step"] = {k: v["launches"] for k, v in prof.items() if v["launches"]}
    with torch.no_grad():          # (multi-tensor ops: a per-parameter expression here was 3 x 521 x 2 launches in the step profile)
        for key, params in (("param_delta_norm", model.parameters()), ("ema_delta_norm", ema.module.parameters())):
            norms = torch._foreach_norm(torch._foreach_sub([p.detach() for p in params], start))
            log[key] = float(torch.linalg.vector_norm(torch.stack(norms)))
    return log


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graphs", action="store_true", help="replay the network's forward / backward as HIP graphs")
    ap.add_argument("--config", default="vidvrd", help="vidvrd (24 pairs x 96 frames) | vidor (48 pairs x 512 frames: --pairs 48)")
    ap.add_argument("--pairs", type=int, default=24)
    ap.add_argument("--profile", action="store_true", help="per-kernel-family HIP-event time of the last step")
    args = ap.parse_args()
    print(json.dumps(run(steps=args.steps, seed=args.seed, graphs=args.graphs, config=args.config, n_pairs=args.pairs, profile=args.profile)))
