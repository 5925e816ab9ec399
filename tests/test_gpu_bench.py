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
    b = run_bench("--pairs", "64", "--steps", "2", "--warmup", "1", "--cpu-pairs", "2", "--no-alt", "--no-ragged", "--no-forward-test",
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
