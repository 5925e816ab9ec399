"""The bench line's contract on a small workload (the driver parses this line; a field that goes missing would only show at
round end): one JSON line on stdout with the metric fields, the `roofline` and `cpu_baseline` objects, and the N > 1 path
started without a launcher (two ranks sharing the GPU over gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags, env=None):
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], capture_output=True, text=True, timeout=600, env=e, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract_small():
    b = run_bench("--pairs", "64", "--steps", "2", "--warmup", "1", "--cpu-pairs", "2", "--no-ragged", "--no-forward-test",
                  "--no-train-step", "--no-shard-projection")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 2 and b["warmup"] == 1 and b["higher_is_better"] is True
    assert b["unit"] == "pairs/s" and b["value"] > 0 and b["vs_baseline"] is None and b["data"] == "synthetic"
    assert "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 64 / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = b["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert "aux" in c["sample"]                                   # says that it computes the heads the GPU leg skips
    assert "64 pairs x 256 frames" in b["metric"]                 # the shape in `metric` is the shape that ran
    # the exact-f32 figure of the same workload, at the top level (the driver's parser keeps top-level keys)
    ra = b["reference_arithmetic"]
    assert ra["gemm_precision"] == "f32" and ra["value"] > 0 and ra["unit"] == "pairs/s"
    assert abs(ra["value"] - 64 / (ra["ms_per_step"] * 1e-3)) <= 1e-6 * ra["value"] and 0 < ra["roofline"]["frac"] < 1


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment: bench.py starts torch.distributed.run itself; here the two
    ranks share the one GPU over gloo (BENCH_REHEARSAL=1).  Whole-job value = all pairs / max-over-ranks time."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--pairs", "64", "--steps", "2", "--warmup", "1",
                          "--no-alt", "--no-ragged", "--no-forward-test", "--no-train-step", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["config"]["world_size"] == 2 and b["config"]["pairs_per_gpu"] == 32
    assert b["scaling"] == "strong" and abs(b["value"] - 64 / (b["ms_per_step"] * 1e-3)) <= 1e-6 * b["value"]


def test_rccl_path_on_one_rank():
    """The code an N > 1 job runs through RCCL -- init_process_group("nccl", device_id=...), the all-gather of the predictions
    on device tensors, dist.barrier(), the float64 all_reduce(MAX) of the step time -- executed on this one-GPU box by a
    process group of ONE rank (BENCH_FORCE_DIST=1), in a fresh child process; then MaskVRD.shard_pairs() under such a group
    (its candidate all-gather forced the same way) against the unsharded forward_test."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_REHEARSAL")}
    env.update(BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    b = run_bench("--gpus", "1", "--pairs", "64", "--steps", "2", "--warmup", "1", "--no-alt", "--no-ragged", "--no-forward-test",
                  "--no-train-step", "--no-cpu-baseline", "--no-shard-projection", env=env)
    assert b["n_gpus"] == 1 and b["config"]["backend"] == "nccl" and b["config"]["world_size"] == 1
    assert "all-gather" in b["config"]["parallelism"] and b["value"] > 0
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.pop("BENCH_FORCE_DIST")
    env.update(VRDONE_FORCE_COLLECTIVE="1", OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "scripts", "sharded_eval_check.py")],
                         env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["world_size"] == 1 and line["equal_on_rank"] == [True] and line["triplets"] > 0
