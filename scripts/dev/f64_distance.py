#!/usr/bin/env python3
"""GPU: distance of every precision mode to the float64 reference goldens, as multiples of the reference's own float32
error (the numbers behind tests/test_gpu_model.py::test_reference_grade_against_float64).
    python scripts/dev/f64_distance.py > gpurun_out/f64_distance.txt"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import torch  # noqa: E402

torch.set_grad_enabled(False)
import test_gpu_model as T  # noqa: E402
from vrdone_amd import ops  # noqa: E402

for mode in ("f32", "f16x3", "bf16x3"):
    ops.set_precision(mode)
    for name, t in T.F64_CASES:
        res = T.f64_distance(name, t)
        print(f"{mode:7s} {name:12s} T{t}: " + "  ".join(
            f"{k[5:]} max x{v[0]:.2f} rms x{v[1]:.2f} (|e| {v[2]:.2e}, ref32 {v[3]:.2e})" for k, v in res.items()), flush=True)
