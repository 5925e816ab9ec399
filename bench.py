#!/usr/bin/env python3
"""Headline benchmark: subject-object pairs/second through MaskVRD._mask_vrd (backbone with SOS
fusion -> 1-D FPN -> mask-segmentation predictor) on synthetic pair x frame tensors.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json metric): configs/vidvrd.yaml model, 2048 pairs x 256 frames, which the
reference's eval batching pads to T_pad = 288 (models/maskvrd.py:378-379); inputs (B, 2069, 288)
fp32 ~ N(0,1) * mask, resident in HBM before the timed region; name-seeded synthetic weights.
A step = one pass of the hot path over the whole 2048-pair batch.  With N GPUs the pair dimension
is sharded (2048/N per rank, strong scaling) and every step ends with an RCCL all-gather of the
per-pair predictions (logits + masks), as BASELINE config 4 describes.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: bf16 / f16 MFMA, dense (not the 2:1-sparsity figure; the f16 forms take the same cycles)
SPLIT_MODES = ("f16x3", "bf16x3")
DTYPE = {"f16x3": "f16x3: f32 products as 3 scaled-f16 MFMA products (~22-bit), f32 accumulate; reference-grade: distance to a float64 run "
                  "of the reference <= 1.4 x the reference's own float32 error (tests/golden/mask_vrd_f64.npz)",
         "bf16x3": "bf16x3 split (~17-bit products), f32 accumulate", "f32": "f32"}
# closed-form algorithmic FLOPs per pair at T_pad (cross-checked with FlopCounterMode)
FLOPS_PER_PAIR = {("vidvrd", 96): 6.386e9, ("vidvrd", 144): 9.650e9, ("vidvrd", 288): 19.897e9,
                  ("vidor_x", 512): 42.038e9}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="vidvrd")
    ap.add_argument("--pairs", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--pair-chunk", type=int, default=0, help="pairs per launch wave inside _mask_vrd (0 = model default)")
    ap.add_argument("--no-prof", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--precision", default=None, choices=["f16x3", "bf16x3", "f32"], help="GEMM product mode of the headline (default: the library default, f16x3)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra runs in the other precision modes")
    ap.add_argument("--no-ragged", action="store_true", help="skip the extra run on ragged pair lengths")
    ap.add_argument("--no-forward-test", action="store_true", help="skip the secondary metric (whole eval call on one synthetic video)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step leg (BASELINE config 3: forward + backward on a 24-pair batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shard-projection", action="store_true",
                    help="skip timing the 1/2, 1/4, 1/8 per-rank shares of the batch on this one GPU (N = 1 only)")
    ap.add_argument("--cpu-pairs", type=int, default=64)
    return ap.parse_args()


def padded_len(cfg, frames, div):
    if frames <= cfg["max_seq_len"]:
        return cfg["max_seq_len"]
    return (frames + div - 1) // div * div


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and would oversubscribe a container share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, quota // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def kernel_source_sha():
    """Content hash of everything the kernels are built from (csrc/*.hip, csrc/*.h, the C-ABI header) plus the host code
    that decides which kernels run on which shapes.  scripts/summarize_traffic.py stamps it into the PMC summary; a
    summary whose stamp differs from the tree that is running describes other kernels and is not reported."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(REPO, "vrdone_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(REPO, "vrdone_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(REPO, "vrdone_amd", "*.py")) +
                   glob.glob(os.path.join(REPO, "vrdone_amd", "models", "*.py")) +
                   [os.path.join(REPO, "include", "vrdone_hip.h")])
    for f in files:
        h.update(os.path.relpath(f, REPO).encode())
        with open(f, "rb") as fh:
            data = fh.read()
        if f.endswith(".py"):
            # host code counts by what it DOES: comments and docstrings do not move a profile's stamp
            import ast
            tree = ast.parse(data)
            for node in ast.walk(tree):
                body = getattr(node, "body", None)
                if isinstance(body, list) and body and isinstance(body[0], ast.Expr) and \
                        isinstance(getattr(body[0], "value", None), ast.Constant) and isinstance(body[0].value.value, str):
                    body[0] = ast.Pass()
            data = ast.dump(tree).encode()
        h.update(data)
    return h.hexdigest()[:16]


def hbm_traffic(kernel, mode):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary for this precision mode
    (profiles/*_hbm_traffic[_f32].json, made by scripts/collect_profiles.sh + scripts/summarize_traffic.py: counters
    cannot be read inside this process).  (None, reason) when there is none or when it was collected on other
    kernel sources than the ones running (its kernel_src_sha stamp differs)."""
    import glob
    suffix = f"_hbm_traffic_{mode}.json"
    if mode == "bf16x3" and not glob.glob(os.path.join(REPO, "profiles", "*" + suffix)):
        suffix = "_hbm_traffic.json"        # (rounds 1-3 named the bf16x3 summary without a mode)
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*" + suffix)))
    if not files:
        return None, "no PMC summary committed for this mode"
    with open(files[-1]) as f:
        d = json.load(f)
    name = os.path.basename(files[-1])
    if d.get("kernel_src_sha") != kernel_source_sha():
        return None, f"{name} is stale: collected on kernel sources {d.get('kernel_src_sha')}, running {kernel_source_sha()}"
    k = d.get("kernels", {}).get(kernel)
    if not k:
        return None, f"{name} has no entry for {kernel}"
    return k["hbm_bytes_per_launch"], f"{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same workload, git {d.get('git_sha', '?')})"


def cpu_baseline(model_cfg, sd_cpu, c_in, frames, t_pad, n_pairs):
    """The oracle (CPU restatement of the reference, plain PyTorch fp32) on a bounded sample of the
    same workload, on this box's host cores."""
    from oracle import vrd_oracle as O
    threads = usable_cores()
    torch.set_num_threads(threads)
    log(f"cpu baseline: {n_pairs} pairs on {threads} threads (os.cpu_count() = {os.cpu_count()})")
    x, m = O.synth_pairs(n_pairs, c_in, t_pad, [frames] * n_pairs, seed=1234)
    with torch.no_grad():
        O.mask_vrd(sd_cpu, model_cfg, x[:2], m[:2], with_aux=True)        # warm-up
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.mask_vrd(sd_cpu, model_cfg, x, m, with_aux=True)            # the reference computes the aux heads too
            times.append(time.perf_counter() - t0)
            log(f"cpu baseline run: {times[-1]:.2f} s")
    med = sorted(times)[1]
    return {"value": n_pairs / med, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n_pairs} pairs x {frames} frames (T_pad {t_pad}), oracle.mask_vrd, median of 3 runs, "
                      f"{med:.2f} s/run; computes the three auxiliary decoder heads too (with_aux=True, as the reference's "
                      "_mask_vrd always does) -- the GPU leg times with_aux=False: last-layer heads only, < 0.3 % of the FLOPs"}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks the way the driver would
    (`python -m torch.distributed.run --nproc-per-node N bench.py ...`) as a CHILD process -- this process has not touched
    the GPU and never does -- relay the child's output (its rank 0 prints the JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")
    # BENCH_REHEARSAL=1: several ranks share the visible GPU(s) over gloo (a single-GPU dry run of the N > 1 path)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # BENCH_FORCE_DIST=1 with --gpus 1: a process group of ONE rank over RCCL, and the exchange step, the barrier and the
    # max-over-ranks reduction run through it like in an N > 1 job (how a single-GPU box executes that code at all)
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if force_dist:
        import socket
        os.environ["VRDONE_FORCE_COLLECTIVE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    elif world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from vrdone_amd import _hip, configs, ops, synth
    from vrdone_amd.models.maskvrd import MaskVRD
    from vrdone_amd.parallel import shard_range, gather_predictions

    cfg = configs.model_config(args.config)
    c_in = configs.input_channels(cfg)
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).eval()
    if args.pair_chunk:
        model.pair_chunk = args.pair_chunk
    t_pad = padded_len(cfg, args.frames, model.max_div_factor)

    log(f"model ready (T_pad {t_pad}, pair_chunk {model.pair_chunk}); generating inputs")
    lo, hi = shard_range(args.pairs, rank, world)
    # every rank generates only its own shard, already in HBM
    x, m = synth.synth_pairs(hi - lo, c_in, t_pad, [args.frames] * (hi - lo), seed=1234 + rank, device=dev)

    batch = {"x": x, "m": m}

    def step():
        # (ragged leg: a NEW mask tensor every call, as an eval loop passes one -- whatever MaskVRD derives from the mask's
        # lengths on the host, and whatever it caches on the tensor, is paid inside the timed region)
        m_step = batch["m"].clone() if batch.get("fresh_mask") else batch["m"]
        out = model._mask_vrd(batch["x"], m_step, with_aux=False)
        if use_dist:
            return gather_predictions(out["pred_logits"], out["pred_masks"], args.pairs, world)
        return out["pred_logits"], out["pred_masks"]

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    LAB_ABLATION = os.environ.get("BENCH_LAB_ABLATION") == "1"
    range_ok = {}

    def run(mode, warmup, steps):
        """W untimed + K timed steps in one GEMM precision mode; returns (seconds, per-family profile)."""
        ops.set_precision(mode)
        with torch.no_grad():
            for i in range(warmup):
                step()
                torch.cuda.synchronize()
                log(f"[{mode}] warmup step {i} done")
            fence()
            # The timed region records HIP events around the launches of the GEMM families only -- what `roofline` needs.  Two
            # events around each of a step's ~600 launches cost 0.8 % of it (142.8 against 141.7 ms, three A/Bs on one box);
            # the other families' times come from one more, untimed step with events around every launch.
            _hip.prof_select(GEMM_FAMILIES)
            _hip.prof_enable(not args.no_prof)
            _hip.prof_reset()
            t0 = time.perf_counter()
            for _ in range(steps):
                logits, masks = step()
            fence()
            elapsed = time.perf_counter() - t0
            _hip.prof_enable(False)
            prof = _hip.prof_read() if (rank == 0 and not args.no_prof) else {}
            if prof and world == 1:
                _hip.prof_select(None)
                _hip.prof_enable(True)
                _hip.prof_reset()
                step()
                torch.cuda.synchronize()
                _hip.prof_enable(False)
                for fam, v in _hip.prof_read().items():
                    if fam not in GEMM_FAMILIES:           # (scaled to the timed region's step count: the tables divide by it)
                        prof[fam] = {k: val * steps for k, val in v.items()}
        # A timed region counts only if its results do: finite logits, and (f16x3) no operand beyond the f16 format's range anywhere
        # in its steps -- a direct caller of _mask_vrd checks the device flag itself, so the bench does.
        # (BENCH_LAB_ABLATION=1: timing-only lab builds of the library, whose results are wrong by design.)
        if not LAB_ABLATION:
            assert logits.shape[0] == args.pairs and bool(torch.isfinite(logits).all())
            bits = ops.f16_range_exceeded(dev)
            assert not bits, f"[{mode}] f16 operand range exceeded inside the timed region ({ops.describe_range(bits)}): the line is void"
        range_ok[mode] = True
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        log(f"[{mode}] timed region: {steps} steps in {float(t.item()):.3f} s")
        return float(t.item()), prof

    GEMM_FAMILIES = ["gemm_f32_mfma", "gemm_x3_mfma", "gemm_x3_dma", "gemm_x3_big"]

    def roofline(mode, prof):
        """Dominant kernel family of the mode: algorithmic FLOPs (2*M*N*K per launch) / HIP-event time."""
        if mode in SPLIT_MODES:    # the split-precision GEMM family that takes the most time (256x256 or 128x256 tile)
            fam = max(("gemm_x3_big", "gemm_x3_dma"), key=lambda f: prof[f]["ms"])
        else:
            fam = "gemm_f32_mfma"
        g = prof[fam]
        n = max(g["launches"], 1)
        avg_ms = g["ms"] / n
        # FLOPs the kernel executed: tiles of padded frames that skip their contraction (padding maps) are not counted,
        # although the reference computes them (SURVEY 8d counts FLOPs at T_pad: that figure is `algorithmic_tflops`)
        executed = g["flops"] - g.get("flops_skipped", 0.0)
        achieved = executed / n / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        if mode in SPLIT_MODES:
            # three 16-bit MFMA products per f32-equivalent product: peak in algorithmic FLOP/s = bf16 / f16 dense peak / 3
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0
            extra = {"mfma_tflops_executed": 3.0 * achieved, "mfma_peak": PEAK_BF16_MFMA_TFLOPS,
                     "note": "peak = 16-bit dense MFMA peak / 3 (a_hi*w_hi + a_hi*w_lo + a_lo*w_hi per product); the kernel family is "
                             "named for its first format: the f16x3 mode runs its F16 = true instantiations (v_mfma_f32_16x16x32_f16)"}
            kern = fam + "_kernel"
        else:
            peak, extra, kern = PEAK_F32_MFMA_TFLOPS, {}, "gemm_f32_mfma_kernel"
        tot = sum(v["ms"] for v in prof.values())
        traffic, src = hbm_traffic(kern, mode)
        if src:
            extra = dict(extra, traffic_source=src)
        return dict({"bound": "mfma", "kernel": kern, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak, "traffic": traffic, "launches": g["launches"], "avg_launch_ms": avg_ms,
                     "flops_per_launch": executed / n, "padding_flops_skipped_per_launch": g.get("flops_skipped", 0.0) / n,
                     "algorithmic_bytes_per_launch": g["bytes"] / n,
                     "share_of_kernel_time": g["ms"] / tot if tot else 0.0}, **extra)

    main_mode = args.precision or ops.get_precision()
    elapsed, prof = run(main_mode, args.warmup, args.steps)
    alts = {}
    if not args.no_alt:
        for alt_mode in ("f16x3", "bf16x3", "f32"):
            if alt_mode != main_mode:
                alts[alt_mode] = run(alt_mode, 1, args.steps)
    ragged = None
    if not args.no_ragged:
        # SURVEY 8d's second input set, reported separately: same pairs, lengths ~ U[2, frames] (seed 1235), padded to
        # the same T_pad like the reference's eval batching pads a slice to its longest pair
        gen = torch.Generator().manual_seed(1235 + rank)
        lens = torch.randint(2, args.frames + 1, (hi - lo,), generator=gen)
        lens[0] = args.frames
        del x, m
        batch["x"], batch["m"] = synth.synth_pairs(hi - lo, c_in, t_pad, lens.tolist(), seed=1234 + rank, device=dev)
        batch["fresh_mask"] = True
        r_elapsed, r_prof = run(main_mode, 1, args.steps)
        batch["fresh_mask"] = False
        ragged = {"lengths": f"U[2, {args.frames}] (seed 1235), mean {float(lens.float().mean()):.1f}, T_pad {t_pad}",
                  "mask": "a fresh mask tensor every step (the tight-padding plan is rebuilt inside the timed region)",
                  "value": args.pairs * args.steps / r_elapsed, "unit": "pairs/s", "ms_per_step": 1e3 * r_elapsed / args.steps,
                  "kernel_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in r_prof.items() if v["launches"]}}

    # What a rank of an N-GPU run computes per step, timed here on ONE GPU: the first pairs/N pairs of the batch for
    # N = 2, 4, 8 (strong scaling: the global batch is fixed).  A projection of the scaling curve, not a measurement of
    # it: it leaves out the all-gather (payload below, < 1 ms over xGMI) and rank-to-rank skew.
    projection = None
    if world == 1 and not args.no_shard_projection:
        if ragged is not None or "x" not in batch:          # the ragged leg replaced the full-length batch
            batch.clear()
            full_x, full_m = synth.synth_pairs(args.pairs, c_in, t_pad, [args.frames] * args.pairs, seed=1234, device=dev)
        else:
            full_x, full_m = batch["x"], batch["m"]
        shares = {}
        for n in (2, 4, 8):
            if args.pairs % n:
                continue
            batch["x"], batch["m"] = full_x[:args.pairs // n].contiguous(), full_m[:args.pairs // n].contiguous()
            ops.set_precision(main_mode)
            with torch.no_grad():
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(max(args.steps, 3)):
                    step()
                torch.cuda.synchronize()
            shares[str(n)] = {"pairs_per_rank": args.pairs // n, "ms_per_step": 1e3 * (time.perf_counter() - t0) / max(args.steps, 3)}
        q, k1 = cfg["predictor"]["num_queries"], cfg["num_classes"] + 1
        one = 1e3 * elapsed / args.steps
        # BASELINE config 4 (8192 pairs x 256 frames, 1024 per rank at N = 8): the whole job on this one GPU (four launch waves
        # of pair_chunk pairs) against one rank's 1024 pairs
        cfg4 = None
        if args.pairs == 2048 and "2" in shares:
            del full_x, full_m
            batch.clear()
            torch.cuda.empty_cache()
            batch["x"], batch["m"] = synth.synth_pairs(8192, c_in, t_pad, [args.frames] * 8192, seed=1234, device=dev)
            with torch.no_grad():
                step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
            whole = 1e3 * (time.perf_counter() - t0) / 2
            cfg4 = {"job": "8192 pairs x 256 frames (BASELINE config 4)", "ms_per_step_n1": whole,
                    "ms_per_step_rank_of_8": shares["2"]["ms_per_step"], "pairs_per_rank_of_8": 1024,
                    "projected_speedup_n8": round(whole / shares["2"]["ms_per_step"], 2),
                    "allgather_payload_bytes_per_step": 8192 * (cfg["predictor"]["num_queries"] * (cfg["num_classes"] + 1) +
                                                                cfg["predictor"]["num_queries"] * t_pad) * 4}
            batch.clear()
            full_x = full_m = None
        projection = {"kind": "projection, not a measurement: per-rank share of the same batch timed on one GPU; no collective, no skew",
                      "ms_per_step_n1": one, "ranks": shares,
                      "projected_speedup": {n: round(one / v["ms_per_step"], 2) for n, v in shares.items()},
                      "allgather_payload_bytes_per_step": args.pairs * (q * k1 + q * t_pad) * 4}
        if cfg4 is not None:
            projection["cfg4_8192_pairs"] = cfg4
        del full_x, full_m

    ft = None
    if not args.no_forward_test and args.config == "vidvrd":
        # secondary metric: the whole eval call MaskVRD.forward_test -- batching of the dataloader's
        # per-pair matrices, the path, device post-processing, result lists -- on one synthetic video of 46 tracklets;
        # with N > 1 the product's pair-sharded form (MaskVRD.shard_pairs: compact-candidate all-gather), every rank
        # holding the same video
        batch.clear()
        torch.cuda.empty_cache()
        ops.set_precision(main_mode)
        model._config_eval(configs.inference_config(args.config))
        if world > 1:
            model.shard_pairs()
        video = synth.synth_video(46, c_in, 200, args.frames, seed=7, device=dev)
        times = []
        with torch.no_grad():
            for _ in range(4):
                fence()
                t0 = time.perf_counter()
                res = model(video)
                fence()
                times.append(time.perf_counter() - t0)
        wall = sorted(times[1:])[1]
        ft = {"pairs": len(video["sids"]), "triplets": len(res["triplets"]), "ms": 1e3 * wall,
              "value": len(video["sids"]) / wall, "unit": "pairs/s", "sharded_over_ranks": world,
              "note": "wall time of one eval call incl. the Python result lists; median of 3 after a warm-up"}
        if world == 1:
            # the same call fed from PER-TRACKLET features on the host (what the dataloader holds before it builds pair
            # matrices): host preparation (box clamp, vIoU de-dup, pair tables) + upload of every tracklet once
            # (proposals.prepare_test_proposal), then forward_test gathering pair rows / box features on the device
            from vrdone_amd.proposals import prepare_test_proposal
            del video
            torch.cuda.empty_cache()
            ic = configs.inference_config(args.config)
            raw = synth.synth_raw_video(46, cfg["visual_dim"], 200, args.frames, seed=7)
            prep, fwd = [], []
            with torch.no_grad():
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    prop = prepare_test_proposal(raw, ic["feat_stride"], 0, 2, dev)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    res2 = model(prop)
                    torch.cuda.synchronize()
                    prep.append(t1 - t0)
                    fwd.append(time.perf_counter() - t1)
            host_mb = sum(v.numel() for v in raw["visual_features_list"]) * 4 / 1e6
            pair_mb = sum(prop["pair_source"].lens) * c_in * 4 / 1e6
            ft["from_tracklets"] = {"pairs": len(prop["sids"]), "triplets": len(res2["triplets"]),
                                    "prepare_ms": 1e3 * sorted(prep[1:])[1], "forward_test_ms": 1e3 * sorted(fwd[1:])[1],
                                    "uploaded_MB": round(host_mb, 1), "pair_matrices_would_be_MB": round(pair_mb, 1),
                                    "note": "host features -> result: prepare = clamp + de-dup + pair tables + upload of each "
                                            "tracklet once; the reference uploads one (L, C_in) matrix per pair"}

    # BASELINE config 3: one training step's forward + backward through the HIP path (vidvrd.yaml, 24 pairs in T_pad 96,
    # synthetic ground truth; optimizer excluded -- that is the reference's own code).  Rank 0, N = 1 only.
    train = None
    if not args.no_train_step and args.config == "vidvrd" and world == 1:
        sys.path.insert(0, os.path.join(REPO, "scripts"))
        from train_step import synthetic_batch
        batch.clear()
        torch.cuda.empty_cache()
        ops.set_precision(main_mode)
        tmodel = synth.load_synthetic_weights(MaskVRD(cfg, device=dev)).to(dev).train()
        tdata = synthetic_batch(cfg, c_in, dev, seed=0)
        def fwd_bwd(n):
            times = []
            for it in range(n):
                tmodel.zero_grad(set_to_none=True)
                with torch.no_grad():      # an optimiser step moved the weights (values unchanged here): the timed step rebuilds its derived operands like a real one
                    torch._foreach_mul_([p for p in tmodel.parameters() if p.requires_grad], 1.0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loss = tmodel(tdata)["total_loss"]
                loss.backward()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            return 1e3 * sorted(times[2:])[(n - 2) // 2], loss
        eager_ms, loss = fwd_bwd(5)
        # launches of one eager forward + backward, by who issues them (torch.profiler; device events)
        launches = None
        try:
            from torch.profiler import profile, ProfilerActivity
            tmodel.zero_grad(set_to_none=True)
            with torch.no_grad():
                torch._foreach_mul_([p for p in tmodel.parameters() if p.requires_grad], 1.0)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as tp:
                tmodel(tdata)["total_loss"].backward()
                torch.cuda.synchronize()
            dev_ev = [e for e in tp.events() if e.device_type == torch.autograd.DeviceType.CUDA]
            mine = sum("anonymous namespace" in e.name and "at::native" not in e.name for e in dev_ev)
            launches = {"device_events": len(dev_ev), "library_kernels": mine, "aten_and_copies": len(dev_ev) - mine}
        except Exception as exc:          # profiling is a report, never a reason to lose the line
            launches = {"error": repr(exc)[:200]}
        n_grad = sum(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in tmodel.parameters() if p.requires_grad)
        n_par = sum(p.requires_grad for p in tmodel.parameters())
        tmodel.enable_training_graphs()                      # the network's forward / backward as two HIP-graph replays
        graph_ms, _ = fwd_bwd(6)                             # (the first of these records them)
        n_grad_g = sum(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in tmodel.parameters() if p.requires_grad)
        train = {"pairs": len(tdata["so_features_list"]), "t_pad": cfg["max_seq_len"], "ms_forward_backward": eager_ms,
                 "ms_forward_backward_hip_graphs": graph_ms,
                 "params_with_finite_grad": f"{n_grad}/{n_par}", "params_with_finite_grad_hip_graphs": f"{n_grad_g}/{n_par}",
                 "total_loss": float(loss.detach()), "launches_forward_backward": launches,
                 "note": "model.train(): forward_training + total_loss.backward() on the HIP backward kernels, stochastic depth on, every "
                         "weight touched in place before each timed step (as after an optimiser update: derived operands are rebuilt); "
                         "median after 2 warm-up steps; hip_graphs: MaskVRD.enable_training_graphs() (vrdone_amd/train_graph.py), "
                         "batching eager, criterion = vrd_criterion_* (costs, assignment, losses, gradients: 4 launches for all layers)"}
        del tmodel, tdata
        # ... and at the largest shipped training shape: configs/vidor.yaml, 48 pairs x 512 frames x 8 heads (eager)
        try:
            torch.cuda.empty_cache()
            vcfg = configs.model_config("vidor")
            tmodel = synth.load_synthetic_weights(MaskVRD(vcfg, device=dev)).to(dev).train()
            tdata = synthetic_batch(vcfg, configs.input_channels(vcfg), dev, n_pairs=48, seed=0)
            v_ms, v_loss = fwd_bwd(5)
            tmodel.enable_training_graphs()
            vg_ms, _ = fwd_bwd(6)                            # (the first of these records the graphs)
            train["vidor_48x512"] = {"pairs": 48, "t_pad": vcfg["max_seq_len"], "ms_forward_backward": v_ms,
                                     "ms_forward_backward_hip_graphs": vg_ms, "total_loss": float(v_loss.detach()),
                                     "note": "eager, same step as above (every weight touched first); global attention forward / backward as the "
                                             "flash-style split-precision kernels (vrd_attention_rows / vrd_attention_bwd, head_dim 64), weight "
                                             "gradients through partial tiles + a chunk-ordered reduction (vrd_gemm_wgrad_x3)"}
            del tmodel, tdata
        except Exception as exc:
            train["vidor_48x512"] = {"error": repr(exc)[:200]}
        torch.cuda.empty_cache()
        model.eval()

    if rank == 0:
        fpp = FLOPS_PER_PAIR.get((args.config, t_pad))
        line = {
            "metric": f"subject-object pairs/sec forward ({args.pairs} pairs x {args.frames} frames x {cfg.get('embd_dim', 512)}-d)",
            "value": args.pairs * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE[main_mode],
            "f16_range_exceeded": False,      # (read from the device after every timed region; a set flag fails the run above)
            "data": "synthetic",
            "config": {"workload": f"{args.config}.yaml MaskVRD._mask_vrd, {args.pairs} pairs x {args.frames} frames "
                                   f"(T_pad {t_pad}) x C_in {c_in}, embd 512, eval, last-layer heads",
                       "pairs_per_gpu": hi - lo, "pair_chunk": model.pair_chunk, "gemm_precision": main_mode,
                       "padding": f"{t_pad - args.frames} of {t_pad} rows per pair are padding; GEMM tiles, attention key tiles and "
                                  "depthwise-conv strips made of padding only are not computed (outputs identical to computing "
                                  "them, tests/test_gpu_model.py; VRDONE_SKIP_PADDING=0 switches the GEMM part off); tight padding "
                                  f"(MaskVRD.tight_len) computes these pairs at T' = {model.tight_len(args.frames, t_pad)}",
                       "parallelism": f"pair-sharded x{world}" + (" + all-gather of predictions" if use_dist else ""),
                       "world_size": dist.get_world_size() if use_dist else 1,
                       "backend": dist.get_backend() if use_dist else None},
        }
        if LAB_ABLATION:
            line["lab_ablation"] = True
        if fpp:
            line["algorithmic_tflops"] = fpp * args.pairs * args.steps / elapsed / 1e12
        if prof:
            line["roofline"] = roofline(main_mode, prof)
            if world > 1:
                line["roofline"]["profiled_families"] = "GEMM families only (shares below are among those)"
            else:
                line["roofline"]["profiled_families"] = ("HIP events inside the timed region around the GEMM families' launches; the other "
                                                         "families' times below: one more, untimed step with events around every launch")
            tot = sum(v["ms"] for v in prof.values())
            line["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in prof.items() if v["launches"]}
            line["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}
            # algorithmic HBM bytes of each family / its time (how close each family runs to the ~6.3 TB/s roof)
            line["kernel_algorithmic_gbps"] = {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
                                               for k, v in prof.items() if v["launches"] and v["ms"] > 0}
        if alts:
            line["alt_precision"] = []
            for alt_mode, (a_elapsed, a_prof) in alts.items():
                entry = {"gemm_precision": alt_mode, "dtype": DTYPE[alt_mode], "value": args.pairs * args.steps / a_elapsed,
                         "ms_per_step": 1e3 * a_elapsed / args.steps}
                if a_prof:
                    entry["roofline"] = roofline(alt_mode, a_prof)
                    entry["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in a_prof.items() if v["launches"]}
                line["alt_precision"].append(entry)
                if alt_mode == "f32":
                    # the same workload in the reference's own arithmetic (exact f32 products on the f32 MFMA), for a reader who
                    # does not accept the split-precision emulation of the headline: top level, next to `roofline`
                    line["reference_arithmetic"] = {k: entry[k] for k in ("gemm_precision", "dtype", "value", "ms_per_step")}
                    line["reference_arithmetic"]["unit"] = "pairs/s"
                    if a_prof:
                        r = entry["roofline"]
                        line["reference_arithmetic"]["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
                                                                    if k in r}
        if ragged is not None:
            line["ragged_variant"] = ragged
        if projection is not None:
            line["shard_projection"] = projection
        if ft is not None:
            line["forward_test"] = ft
        if train is not None:
            line["train_step"] = train
        if world == 1 and not args.no_cpu_baseline:
            sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            line["cpu_baseline"] = cpu_baseline(cfg, sd_cpu, c_in, args.frames, t_pad, args.cpu_pairs)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
