#!/usr/bin/env python3
"""DistributedDataParallel around the HIP model, the way the reference's train.py wraps it (train.py:103-108:
DDP(model, device_ids=[rank], find_unused_parameters=False)): every rank differentiates its own shard of a 24-pair
batch; the all-reduced (averaged) gradients must equal the gradient of the mean of the shard losses computed without
DDP, and every parameter must have taken part (find_unused_parameters=False tolerates no unused one).

One-GPU rehearsal (ranks share the card, gradient all-reduce on gloo):
    BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29513 scripts/ddp_step_check.py
Multi-GPU node: drop BENCH_REHEARSAL (RCCL).  Launch from a shell (a process that has not touched the GPU).
`--graphs`: additionally take three DDP steps with MaskVRD.enable_training_graphs() (forward / backward of the network as
HIP-graph replays, vrdone_amd/train_graph.py) and compare their all-reduced gradients with the eager DDP step's.
"""
import json
import os
import sys

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "scripts"))


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist.init_process_group("gloo" if rehearsal else "nccl")

    from train_step import synthetic_batch
    from vrdone_amd import configs, synth
    from vrdone_amd.models.blocks import AffineDropPath
    from vrdone_amd.models.maskvrd import MaskVRD
    cfg = configs.model_config("vidvrd")
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).train()
    for mod in model.modules():                     # deterministic loss: no stochastic depth
        if isinstance(mod, AffineDropPath):
            mod.drop_prob = 0.0
    data = synthetic_batch(cfg, configs.input_channels(cfg), dev, n_pairs=24, seed=0)      # same on every rank
    per = 24 // world

    def shard(r):
        return {k: v[r * per:(r + 1) * per] for k, v in data.items()}

    # reference: mean of the shard losses, no DDP
    model.zero_grad(set_to_none=True)
    total = sum(model(shard(r))["total_loss"] for r in range(world)) / world
    total.backward()
    want = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)

    ddp = DDP(model, device_ids=None if rehearsal else [dev_index], find_unused_parameters=False)
    loss = ddp(shard(rank))["total_loss"]
    loss.backward()
    graph_worst = None
    if "--graphs" in sys.argv:
        # the same step with the network's forward / backward replayed from HIP graphs (MaskVRD.enable_training_graphs):
        # the recording differentiates aliases of the parameters, the replaying autograd.Function hands the gradients to the
        # real ones, so DDP's reducer hooks fire outside the graphs and all-reduce as in the eager step
        eager = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        model.enable_training_graphs()
        for _ in range(3):                           # the first of these records
            model.zero_grad(set_to_none=True)
            ddp(shard(rank))["total_loss"].backward()
        from vrdone_amd import train_graph
        assert len(train_graph.recordings(model)) == 1
        graph_worst = max(float((p.grad - eager[n]).norm()) / (float(eager[n].norm()) + 1e-4 * max(float(g.norm()) for g in eager.values()))
                          for n, p in model.named_parameters())
    worst, missing = 0.0, []
    scale = max(float(g.norm()) for g in want.values())
    for n, p in model.named_parameters():
        if p.grad is None:
            missing.append(n)
            continue
        worst = max(worst, float((p.grad - want[n]).norm()) / (float(want[n].norm()) + 1e-4 * scale))
    flags = [None] * world
    dist.all_gather_object(flags, (worst, missing, graph_worst))
    if rank == 0:
        print(json.dumps({"check": "DDP-averaged gradients == gradient of the mean shard loss; every parameter used",
                          "world_size": world, "backend": dist.get_backend(), "rehearsal_on_one_gpu": rehearsal,
                          "worst_relative_gradient_error_per_rank": [f[0] for f in flags],
                          "parameters_without_gradient": [f[1] for f in flags], "loss_rank0": float(loss.detach()),
                          "graph_replayed_ddp_step_vs_eager_ddp_step": [f[2] for f in flags]}), flush=True)
    dist.destroy_process_group()
    if any(f[0] > 1e-4 or f[1] or (f[2] is not None and f[2] > 2e-4) for f in flags):
        sys.exit(1)


if __name__ == "__main__":
    main()
