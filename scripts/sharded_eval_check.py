#!/usr/bin/env python3
"""Pair-sharded MaskVRD.forward_test with REAL processes (one per rank), checked against the single-process result.

On the one-GPU box (a rehearsal: all ranks share the card, the collective runs on gloo):
    BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29511 scripts/sharded_eval_check.py
On a multi-GPU node drop BENCH_REHEARSAL (one GPU per rank, RCCL).  Launch it from a shell, i.e. from a process that has
not touched the GPU.  Every rank computes the sharded result and the unsharded one and compares them field by field;
rank 0 prints one JSON line.
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if rehearsal:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=dev)

    from vrdone_amd import configs, synth
    from vrdone_amd.models.maskvrd import MaskVRD
    name = "vidvrd"
    cfg = configs.model_config(name)
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
    model._config_eval(configs.inference_config(name))
    video = synth.synth_video(24, configs.input_channels(cfg), 40, 250, seed=11, device=dev)     # same on every rank
    with torch.no_grad():
        want = model(video)
        model.shard_pairs()
        model(video)                                         # warm-up (process-group buffers)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        got = model(video)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
    same = all(got[k] == want[k] for k in ("triplets", "pred_durations", "so_tids", "triple_scores",
                                           "triple_scores_avg", "so_trajs"))
    flags = [None] * world
    dist.all_gather_object(flags, bool(same))
    if rank == 0:
        print(json.dumps({"check": "sharded forward_test == single-process forward_test, every field, every rank",
                          "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                          "rehearsal_on_one_gpu": rehearsal, "pairs": len(video["sids"]),
                          "triplets": len(got["triplets"]), "equal_on_rank": flags, "sharded_ms": round(ms, 2)}), flush=True)
    dist.destroy_process_group()
    if not all(flags):
        sys.exit(1)


if __name__ == "__main__":
    main()
