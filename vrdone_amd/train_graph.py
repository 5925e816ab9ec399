"""Training steps as two HIP graphs (forward, backward) instead of ~2,000 kernel launches.

A training batch is small (24 pairs x 96 frames under vidvrd.yaml): its forward + backward are ~1,200 launches of this
library plus the tensor ops between them, 26 ms of kernel time in 44-54 ms of wall time -- the step is bound by launch
overhead, not by the kernels.  `MaskVRD.enable_training_graphs()` records the network part of the step -- `_mask_vrd`,
everything from the padded batch to the predictions and back to the parameter gradients -- once per batch shape and
replays it afterwards; batching, Hungarian matching and the losses in between stay eager (the matcher's result decides
which tensor ops follow).  What makes the library's ops recordable: they launch on torch's current stream, allocate
through torch's allocator, never synchronise, and -- `ops._capturing()` -- build the derived weight operands (split /
packed / depthwise parameter images) inside the recording instead of taking them from the per-weight cache, so that every
replay derives them from the weights as they are then.

The recording follows torch.cuda.make_graphed_callables (warm-up on a side stream, forward graph, backward graph through
torch.autograd.grad into static gradient buffers, one autograd.Function that replays them) with one difference: it runs
on fresh leaf tensors that ALIAS the parameters' storage, swapped into the modules for the duration of the recording.  The
autograd engine queues a leaf's gradient on the stream of the leaf's AccumulateGrad node; a parameter that took part in an
earlier eager step whose loss is still referenced (every ordinary training loop, every batch shape after the first) has
that node on the default stream, and an event wait on the default stream in the middle of a capture aborts the process.
The aliases have no autograd history, the graphs read and write the same memory, and the replaying Function hands the
gradients to the real parameters.

DistributedDataParallel (how the reference's train.py:103-108 always wraps the model) works around it unchanged: the
recording differentiates the aliases, so no reducer hook runs inside a capture, and the replaying Function's backward hands
every gradient to the real parameters at once -- their AccumulateGrad hooks then fire outside the graphs and the reducer
all-reduces its buckets as in an eager step (scripts/ddp_step_check.py --graphs: graph-replayed DDP steps on two ranks
give the eager DDP step's averaged gradients, profiles/r03_ddp_graph_step_check.json).

Limits: one backward per forward, and gradients are not accumulated across steps -- they come back in the recording's own
buffers, which `.grad` then aliases (as with make_graphed_callables): call `zero_grad(set_to_none=True)` (torch's default)
between steps; the per-kernel profiler (`_hip.prof_*`) sees nothing inside a replay; at most `MAX_SHAPES` batch shapes are
recorded per model, any further shape runs eagerly.
"""
import weakref

import torch
from torch import nn
from torch.autograd.function import once_differentiable
from torch.utils import _pytree as pytree

from . import ops

MAX_SHAPES = 4
WARMUP_ITERS = 3

_GRAPHS = weakref.WeakKeyDictionary()          # model -> {on, recordings: {(batch shape, deep supervision, precision mode): _Recording}}


def _storage_key(model):
    """What a recording is tied to: the parameter objects that take gradients and the memory they live in (a model moved
    with .to(), cast, or given new parameter tensors needs new graphs; in-place updates and load_state_dict do not)."""
    return tuple((id(p), p.data_ptr()) for p in model.parameters() if p.requires_grad)


class _Recording:
    """Forward and backward graph of `model._mask_vrd` for one batch shape."""

    def __init__(self, model, x, m):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.storage = _storage_key(model)
        aliases = [nn.Parameter(p.detach()) for p in self.params]             # same storage, no autograd history
        by_id = {id(p): a for p, a in zip(self.params, aliases)}
        slots = [(mod, name, p) for mod in model.modules() for name, p in mod._parameters.items() if id(p) in by_id]
        self.static_x, self.static_m = x.clone(), m.clone()

        def network():
            return model._mask_vrd(self.static_x, self.static_m, with_aux=model.deep_supervision)

        def grad_outputs(flat):
            return [torch.empty_like(o) if o.requires_grad else None for o in flat]

        def backward(flat, gouts):
            return torch.autograd.grad([o for o in flat if o.requires_grad], aliases,
                                       [g for g in gouts if g is not None], allow_unused=True)
        try:
            for mod, name, p in slots:
                mod._parameters[name] = by_id[id(p)]
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                     # lazy initialisation (LDS opt-ins, allocator) outside the graphs
                for _ in range(WARMUP_ITERS):
                    flat = pytree.tree_leaves(network())
                    grads = backward(flat, grad_outputs(flat))
                    del flat, grads
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            self.fwd, self.bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with ops.presplit_scope() as scope:  # the backward graph reads the operands the forward graph's one split launch writes
                self.split_plans = scope.plans   # job tables and operand buffers the captured launches point into
                with torch.cuda.graph(self.fwd, pool=pool):
                    out = network()
                self.static_out, self.spec = pytree.tree_flatten(out)
                self.static_gout = grad_outputs(self.static_out)
                with torch.cuda.graph(self.bwd, pool=pool):
                    self.static_gin = backward(self.static_out, self.static_gout)
        finally:
            for mod, name, p in slots:
                mod._parameters[name] = p

    def __call__(self, x, m):
        flat = _Replay.apply(self, x, m, *self.params)
        return pytree.tree_unflatten(list(flat), self.spec)


class _Replay(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rec, x, m, *params):
        rec.static_x.copy_(x)
        rec.static_m.copy_(m)
        rec.fwd.replay()
        ctx.rec = rec
        outs = tuple(o.detach() for o in rec.static_out)
        ctx.mark_non_differentiable(*[o for o, s in zip(outs, rec.static_out) if not s.requires_grad])
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        rec = ctx.rec
        for buf, g in zip(rec.static_gout, grads):
            if buf is not None:
                if g is None:
                    buf.zero_()                      # an output the loss did not use
                else:
                    buf.copy_(g)
        rec.bwd.replay()
        # The gradients handed back are views of the recording's static buffers, which AccumulateGrad may adopt as `.grad`.
        # A step that KEEPS its .grad (zero_grad(set_to_none=False), gradient accumulation) would have it overwritten in place
        # by the next replay and then added to itself: such a parameter gets a copy instead.
        out = []
        for p, g in zip(rec.params, rec.static_gin):
            if g is None:
                out.append(None)
            elif p.grad is not None and p.grad.untyped_storage().data_ptr() == g.untyped_storage().data_ptr():
                raise RuntimeError("training graphs: a parameter's .grad is the recording's own gradient buffer from the previous "
                                   "step and would be overwritten -- clear gradients with zero_grad(set_to_none=True), or clone "
                                   ".grad before the next step (gradient accumulation)")
            else:
                out.append(g.detach().clone() if p.grad is not None else g.detach())
        return (None, None, None) + tuple(out)


def enable(model, on=True):
    """Switch graph replay of the training network on or off for `model`.  Switching off keeps what was recorded (switching
    on again replays it); `forget(model)` frees it."""
    state = _GRAPHS.setdefault(model, {"on": False, "recordings": {}})
    state["on"] = bool(on)


def forget(model):
    _GRAPHS.pop(model, None)


def enabled(model):
    state = _GRAPHS.get(model)
    return state is not None and state["on"]


def recordings(model):
    state = _GRAPHS.get(model)
    return {} if state is None else state["recordings"]


def mask_vrd(model, x, m):
    """`model._mask_vrd(x, m, with_aux=model.deep_supervision)` through the recorded graphs (recorded on the first use of a
    batch shape; that call replays them too and returns real predictions)."""
    graphs = _GRAPHS[model]["recordings"]
    # everything that shapes the captured launch sequence besides the batch shape: precision mode, launch-wave size, the
    # stochastic-depth probabilities and pinned keep vectors of every AffineDropPath, which parameters train
    from .models.blocks import AffineDropPath
    drops = tuple((mod.drop_prob, None if mod.keep is None else mod.keep.data_ptr()) for mod in model.modules()
                  if isinstance(mod, AffineDropPath))
    trainable = tuple(p.requires_grad for p in model.parameters())
    key = (tuple(x.shape), tuple(m.shape), bool(model.deep_supervision), ops.get_precision(), int(model.pair_chunk), drops,
           trainable)
    rec = graphs.get(key)
    if rec is not None and rec.storage != _storage_key(model):
        graphs.clear()                         # the parameters were replaced or moved: every recording is stale
        rec = None
    if rec is None:
        if len(graphs) >= MAX_SHAPES:
            return model._mask_vrd(x, m, with_aux=model.deep_supervision)
        assert x.is_cuda and not x.requires_grad
        rec = graphs[key] = _Recording(model, x, m)
    return rec(x, m)
