"""forward_test's network as HIP graphs for the videos real data has (tens to a few hundred pairs of <= 128 frames).

Such a call is bound by the host: a 48-pair video is ~210 kernel launches of this library for 4 ms of kernel time in 6.4 ms
of wall time (profiles/r06_forward_test_small.json).  The reference evaluates one video per call (eval.py:72-80) in slices of
<= max_so_pair pairs (models/maskvrd.py:208-227); here a video of fewer than MaskVRD.ROWS_MIN_PAIRS pairs runs bucket by bucket
-- all pairs of one padded length T as one batch -- and a bucket's whole device side (vrd_pack_pairs from the dataloader's
per-pair matrices, backbone, FPN, predictor, vrd_postprocess) is recorded once per (T, padded pair count) and replayed.

What makes a bucket recordable where a whole video is not: its launch sequence depends on (T, number of pairs) only.  The
pair count is rounded up to one of `SIZES` with stand-in pairs (one frame of zeros): a pair's result does not depend on what
else is in its batch (tests/test_gpu_model.py: batch composition), so the real pairs' bits are the eager path's, and a
handful of recordings per padded length serves every video.  The inputs of a recording are two small device tables (pointers
to the pairs' (L, C_in) matrices, their lengths) that are overwritten before each replay; derived weight operands are built
INSIDE the recording (ops._capturing), so a replay always sees the weights as they are, as with the training graphs
(train_graph.py).  A recording is tied to the parameter tensors' storage; at most MAX_RECORDINGS are kept per model, further
shapes run eagerly, as does anything that fails to record.

OFF by default (VRDONE_EVAL_GRAPHS=1 switches it on): the replayed call returns exactly the eager call's result
(tests/test_gpu_model.py::test_forward_test_graph_replay_equals_eager, and the forward_test goldens with the switch on), but on
this stack (ROCm 7.2, PyTorch 2.10) replaying a bucket's graph takes about twice the eager call: 16 pairs x 32 frames 8.1 ms
against 4.3, 48 x 64: 8.8 / 4.3, 128 x 96: 11.0 / 6.0, 256 x 128: 15.5 / 10.6 (scripts/dev/eval_graph_probe.py,
profiles/r06_lab_eval_graph_probe.txt; ~16 us per graph node where the eager launch path spends ~10 us of host time per launch
and overlaps it with the GPU) -- a 48-pair video 11.3 ms per call against 6.4.  The training step gains from its graphs
(32 -> 20 ms) because what they remove there is autograd and Python, not launches.
"""
import os
import weakref

import torch

from . import ops

ENABLED = os.environ.get("VRDONE_EVAL_GRAPHS", "0") != "0"
MAX_PAIRS = 256                     # larger buckets keep the GPU busy on their own
MAX_FRAMES = 256
MAX_RECORDINGS = 24
WARMUP_ITERS = 2
SIZES = (8, 16, 24, 32, 48, 64, 96, 128, 192, 256)

_GRAPHS = weakref.WeakKeyDictionary()          # model -> {"recordings": {key: _Recording}, "failed": set of keys}


def _storage_key(model):
    return tuple((id(p), p.data_ptr()) for p in model.parameters())


storage_key = _storage_key


def pad_size(n):
    for s in SIZES:
        if s >= n:
            return s
    return None


class _Recording:
    """One bucket shape: (T, n_pad pairs) -> vrd_postprocess's four outputs, as a graph over two static input tables."""

    def __init__(self, model, T, n_pad, k, c_in):
        dev = model.device
        self.n_pad = n_pad
        self.storage = _storage_key(model)
        self.stand_in = torch.zeros(1, c_in, device=dev, dtype=torch.float32)      # a one-frame pair of zeros
        self.table = torch.full((n_pad,), self.stand_in.data_ptr(), dtype=torch.int64, device=dev)
        self.lens = torch.ones(n_pad, dtype=torch.int32, device=dev)

        def network():
            return model._bucket_candidates(self.table, self.lens, T, k)

        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):             # lazy initialisation (LDS opt-ins, allocator) outside the graph
            for _ in range(WARMUP_ITERS):
                network()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with ops.presplit_scope() as scope:
            self.split_plans = scope.plans        # job tables and operand buffers the captured launches point into
            with torch.cuda.graph(self.graph):
                self.outs = network()

    def __call__(self, table, lens):
        n = table.numel()
        self.table[:n].copy_(table)
        self.lens[:n].copy_(lens)
        if n < self.n_pad:                        # (always: an entry left from an earlier video may point into freed memory)
            self.table[n:].fill_(self.stand_in.data_ptr())
            self.lens[n:].fill_(1)
        self.graph.replay()
        return tuple(o[:n] for o in self.outs)    # views of the recording's buffers: consumed before its next replay


def bucket_candidates(model, table, lens, T, k, c_in, storage=None):
    """`model._bucket_candidates(table, lens, T, k)` through a recorded graph, or None when this bucket runs eagerly (replay
    switched off, bucket too large, recording budget used up, the shape failed to record before).
    storage: storage_key(model), if the caller has it (one walk over the parameters per video instead of one per bucket)."""
    n = int(table.numel())
    if not ENABLED or n > MAX_PAIRS or T > MAX_FRAMES or torch.cuda.is_current_stream_capturing():
        return None
    n_pad = pad_size(n)
    state = _GRAPHS.setdefault(model, {"recordings": {}, "failed": set()})
    graphs = state["recordings"]
    key = (int(T), n_pad, int(k), int(c_in), ops.get_precision(), ops.pair_mode())
    if key in state["failed"]:
        return None
    rec = graphs.get(key)
    if rec is not None and rec.storage != (storage if storage is not None else _storage_key(model)):
        graphs.clear()                            # the parameters were replaced or moved: every recording is stale
        rec = None
    if rec is None:
        if len(graphs) >= MAX_RECORDINGS:
            return None
        try:
            rec = _Recording(model, T, n_pad, k, c_in)
        except Exception as e:                    # noqa: BLE001 -- whatever keeps a shape from recording: it runs eagerly from now on
            import warnings
            warnings.warn(f"vrdone_amd: forward_test bucket (T = {T}, {n_pad} pairs) could not be recorded as a HIP graph ({e!r}); "
                          "it runs eagerly")
            state["failed"].add(key)
            torch.cuda.synchronize()
            return None
        graphs[key] = rec
    return rec(table, lens)


def recordings(model):
    state = _GRAPHS.get(model)
    return {} if state is None else state["recordings"]


def forget(model):
    _GRAPHS.pop(model, None)
