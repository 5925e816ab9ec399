"""Tensor-level wrappers over the C ABI (include/vrdone_hip.h).

Activations are fp32 HIP tensors in channels-last form, shape ``(B, T, C)`` (or any
``(..., C)``) whose last stride is 1 and whose rows are uniformly strided, so a column slab of a
wider buffer (``buf[..., 512:1024]``) is a legal operand: that is how concatenations are built
without copies.  Masks are ``torch.bool``/``uint8`` tensors of shape ``(B, T)``.

PyTorch is used for device memory and the current stream only; every op below runs one
hand-written HIP kernel.  Nothing here works on CPU tensors.
"""
import ctypes as C
import os

import torch

from . import _hip
from ._hip import ACT_GELU, ACT_NONE, ACT_RELU, lib  # noqa: F401


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """torch's current stream of the current device as a hipStream_t.  (Through the raw-handle call: torch.cuda.current_stream()
    builds a Stream object per call, ~5 us of the ~15 us an op of this module spends on the host -- a forward_test call on a
    real-sized video is ~230 launches and bound by exactly that.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t):
    """(data_ptr, n_rows, n_cols, leading dimension) of a channels-last operand."""
    if not t.is_cuda:
        raise RuntimeError("vrdone_amd ops need HIP tensors (no CPU path exists for the hot path)")
    if t.dtype != torch.float32:
        raise TypeError(f"expected float32, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError("channels-last operand must have unit stride along channels")
    cols = t.shape[-1]
    rows = t.numel() // cols
    if t.dim() == 1 or rows == 1:
        return t.data_ptr(), rows, cols, cols
    ld = t.stride(-2)
    for d in range(t.dim() - 3, -1, -1):       # outer dims must collapse onto the row index
        if t.shape[d] != 1 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
            raise ValueError(f"rows are not uniformly strided: shape {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), rows, cols, ld


def _ptr(t):
    return None if t is None else t.data_ptr()


def _param_ptr(t, like, what="parameter"):
    """Device pointer of a parameter tensor handed to a kernel as a flat f32 array (None stays None): it must be a
    contiguous float32 tensor on the activations' device -- a model left on the CPU, or after .half() / .double(),
    raises here instead of faulting on the GPU."""
    if t is None:
        return None
    if not t.is_cuda or t.device != like.device:
        raise RuntimeError(f"{what} lives on {t.device}, the activations on {like.device}: move the model with .to(device)")
    if t.dtype != torch.float32:
        raise TypeError(f"{what} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what} must be contiguous")
    return t.data_ptr()


class Pair:
    """A channels-last tensor stored in the GEMM-operand "pair" format of the split-precision modes: the 4*C bytes of a
    C-channel row (C % 32 == 0) hold, per block of 32 channels, [32 x 16-bit hi | 32 x 16-bit lo] instead of 32 floats --
    bf16 planes of x (fmt PAIR_BF16, the bf16x3 mode) or f16 planes of x * 2^F16_ACT_EXP (PAIR_F16, the f16x3 mode).  `t` is
    the f32-typed buffer (same shape / strides as the f32 tensor would have); `width` records the width of the producer's
    slab (the format itself does not depend on it).  Only conv_gemm() and attention() consume a Pair."""
    __slots__ = ("t", "width", "fmt")

    def __init__(self, t, width, fmt=None):
        self.t, self.width, self.fmt = t, width, pair_fmt() if fmt is None else fmt
        assert self.fmt in (_hip.PAIR_BF16, _hip.PAIR_F16)

    @property
    def shape(self):
        return self.t.shape

    def __getitem__(self, idx):      # batch slicing (rows) keeps the format
        return Pair(self.t[idx], self.width, self.fmt)

    def float(self):
        """Decode to f32 (hi + lo); for tests."""
        f16 = self.fmt == _hip.PAIR_F16
        raw = self.t.contiguous().view(torch.float16 if f16 else torch.bfloat16)                 # (..., 2*C)
        blocks = raw.reshape(*raw.shape[:-1], -1, 2, 32).float()
        val = (blocks[..., 0, :] + blocks[..., 1, :]).reshape(*self.t.shape)
        return val * 2.0 ** -_hip.F16_ACT_EXP if f16 else val


def flash_pair_ok(n_head, channels, Tq):
    """True when global attention of this shape runs the split-precision flash kernel on pair-row q/k/v."""
    return pair_mode() and channels // n_head in (64, 128) and Tq >= 32


def pair_mode():
    """True when producers should emit pair rows for GEMM-only consumers: a split-precision mode and autograd not recording
    (a differentiable forward keeps every activation as plain f32 rows, see vrdone_amd/autograd.py)."""
    return _precision in ("bf16x3", "f16x3") and not torch.is_grad_enabled()


def pair_fmt():
    """enum vrd_pair_format of the current precision mode's pair rows and split weights (0 in the f32 mode)."""
    return {"bf16x3": _hip.PAIR_BF16, "f16x3": _hip.PAIR_F16}.get(_precision, _hip.PAIR_NONE)


def split_backward():
    """True when the backward GEMMs (input and weight gradients) run as split-precision products: bf16 planes in the bf16x3 mode;
    in the f16x3 mode f16 planes of the gradient times a per-tensor power of two (backward_fmt, grad_scale) -- reference-grade
    products like the forward pass's (the reference differentiates in float32, train.py:182-186)."""
    return _precision in ("bf16x3", "f16x3")


_F16_BACKWARD = os.environ.get("VRDONE_F16_BACKWARD", "1") != "0"       # A/B switch: 0 = the f16x3 mode's backward on bf16 planes


def backward_fmt():
    """element format of the backward GEMMs' split operands"""
    return _hip.PAIR_F16 if (_precision == "f16x3" and _F16_BACKWARD) else _hip.PAIR_BF16


_grad_scales = {}


def grad_scale(g, slot=0):
    """{2^e, 2^-e} (device floats [0], [1] of a buffer of _hip.ABSMAX_SCALE_FLOATS) with max |g| 2^e in [2^13, 2^14) for the (rows, C) gradient `g`
    (vrd_absmax_scale, one launch), or None when g's rows are not float4-aligned (the caller then keeps the bf16 planes).  The
    buffer is one per (device, stream): it is only read by the launches that directly follow on the same stream; launches
    recorded into a graph get one of their own per call (from the graph's pool)."""
    pg, rows, cols, ldg = _rows(g)
    if cols % 4 or ldg % 4 or pg % 16:
        return None
    if torch.cuda.is_current_stream_capturing():
        # (a slice of a zeroed block of the recording: one fill node per block instead of one per call -- ~140 per recorded step)
        from .autograd import _zeros
        buf = _zeros(_hip.ABSMAX_SCALE_FLOATS, device=g.device)
    else:
        key = (g.device, torch.cuda.current_stream(g.device).cuda_stream, slot)      # (slot: a consumer that needs two at once)
        buf = _grad_scales.get(key)
        if buf is None:
            buf = _grad_scales[key] = torch.zeros(_hip.ABSMAX_SCALE_FLOATS, device=g.device, dtype=torch.float32)
    _hip.check(lib.vrd_absmax_scale(pg, ldg, rows, cols, buf.data_ptr(), _stream()), "vrd_absmax_scale")
    return buf


def _fmt(pair):
    """an op's `pair=` flag -> the out_pair argument of its C entry point"""
    return pair_fmt() if pair else _hip.PAIR_NONE


def recording(*tensors):
    """True when autograd is recording and one of the tensors needs a gradient: the op then runs as the
    torch.autograd.Function(s) of vrdone_amd/autograd.py (HIP kernels forward and backward)."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def join(buf, parts, dim=-1):
    """The concatenation of `parts` along `dim`: `buf`, which the producers already filled through `out=` slabs, or
    -- under autograd, where ops return fresh tensors and ignore `out=` -- torch.cat(parts)."""
    if torch.is_grad_enabled() and any(torch.is_tensor(q) and q.requires_grad for q in parts):
        return torch.cat(parts, dim=dim)
    return buf


def _unwrap(x):
    return (x.t, x.width) if isinstance(x, Pair) else (x, 0)


def _mask_ptr(m, rows):
    if m is None:
        return None
    if m.dtype not in (torch.bool, torch.uint8) or not m.is_contiguous() or m.numel() != rows:
        raise ValueError(f"mask must be a contiguous bool/uint8 tensor with one byte per row ({m.numel()} vs {rows})")
    return m.data_ptr()


# Derived weight layouts are cached ON the parameter object (so they die with it and can never be
# confused with another tensor that later reuses the same address), keyed on (data_ptr, version).
_capture_keep = []


def _capturing():
    """True while the current stream records a HIP graph (vrdone_amd/train_graph.py): derived operands are then built
    inside the graph on every replay -- a cache hit would leave their kernels out of the recording, and every replay
    would see the operand of the weights (or mask) as they were at capture time."""
    on = torch.cuda.is_current_stream_capturing()
    if not on and _capture_keep:
        _capture_keep.clear()
    return on


def _keep_for_capture(val):
    """Operands built during a capture stay referenced until it is over: call sites hand their addresses to a launch and
    drop the tensor, which the cache normally keeps alive -- freed inside a capture, the pool would give the memory to the
    next allocation before the launch that reads it."""
    _capture_keep.append(val)
    return val


def _cached(w, slot, build):
    key = (w.data_ptr(), w._version)
    hit = getattr(w, slot, None)
    if _capturing():
        # inside a recording only what presplit_weights built inside the SAME recording scope counts (its launch is part of
        # the graph; train_graph opens one scope around a recording's forward and backward graphs)
        if hit is not None and hit[0] == key and len(hit) > 2 and hit[2] is not None and hit[2] is _presplit_scope:
            return hit[1]
        return _keep_for_capture(build())
    if hit is not None and hit[0] == key:
        return hit[1]
    val = build()
    setattr(w, slot, (key, val))
    return val


def packed_conv_weight(w):
    """Conv1d weight (N, Cin, k) -> (N, k*Cin) tap-major for k = 3; k = 1 weights are used in place."""
    if w.shape[-1] == 1:
        return w
    return _cached(w, "_vrd_packed", lambda: w.detach().permute(0, 2, 1).contiguous())


# GEMM precision mode.
#   "bf16x3": every f32 product a*w is formed as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the bf16
#            MFMA with f32 accumulation (x = x_hi + x_lo split in bf16): ~17 significand bits per product.
#            Measured end to end: logits within 7e-5 of the reference (stated tolerance 1e-3).
#   "f16x3": the same three products on the f16 MFMA, operands scaled by exact powers of two (activations 2^4, weights per
#            tensor): ~22 significand bits per product.  The reference-grade mode: end to end it is as far from a float64
#            run of the reference as the reference's own float32 run is (x1.0-1.3, tests/golden/mask_vrd_f64.npz), at the
#            speed of bf16x3.  Forward products only: the backward GEMMs of this mode are bf16x3's (gradients have no
#            fixed scale to split an f16 pair at).  The default.
#            Activations beyond +-4094 overflow the f16 planes.  Every kernel that writes such planes reports it in a flag word
#            on the device (f16_range_flag()); MaskVRD.forward_test, forward_training and forward_loss read the word with their
#            results and repeat the call in the f32 mode; a direct caller of _mask_vrd checks f16_range_exceeded() itself.
#   "f32":   exact f32 MFMA products (bit-level fmaf chains); logits within 9e-6; ~2.8x slower end to end (bench.py).
# Select with set_precision() or the VRDONE_PRECISION environment variable.  Everything outside the
# conv GEMMs and the global attention (LayerNorm, depthwise convs, softmax, banded attention) is f32 in all modes.
_PRECISIONS = ("f32", "bf16x3", "f16x3")
_precision = os.environ.get("VRDONE_PRECISION", "f16x3")
if _precision not in _PRECISIONS:
    raise ValueError(f"VRDONE_PRECISION must be one of {_PRECISIONS}, got {_precision!r}")


def set_precision(mode):
    global _precision
    if mode not in _PRECISIONS:
        raise ValueError(f"precision must be one of {_PRECISIONS}, got {mode!r}")
    _precision = mode


def get_precision():
    return _precision


class use_precision:
    """`with ops.use_precision("f32"):` -- the mode inside the block, the previous one after it."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = get_precision()
        set_precision(self.mode)
        return self

    def __exit__(self, *exc):
        set_precision(self.prev)
        return False


# ---- the f16x3 mode's operand range (include/vrdone_hip.h, vrd_f16_range_flag)
_range_flags = {}


class _DeviceWord:
    """one int32 of library-owned device memory, presented to torch through __cuda_array_interface__"""

    def __init__(self, ptr):
        self.__cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def f16_range_flag(device=None):
    """The device's f16 operand-range flag as a 1-element int32 tensor (a view of the library's word, not a copy): non-zero
    once any producer of f16 pair rows met a value beyond +-4094 since the word was last cleared; the bits name the
    producing kernel families (_hip.RANGE_TAGS).  Read it WITH a call's results (`.clone()` on the same stream, or one
    `.item()` where the results are synchronised anyway) and clear it with `.zero_()`."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    flag = _range_flags.get(dev.index)
    if flag is None:
        import ctypes as C
        with torch.cuda.device(dev):
            ptr = C.c_void_p()
            _hip.check(lib.vrd_f16_range_flag(C.byref(ptr)), "vrd_f16_range_flag")
            flag = _range_flags[dev.index] = torch.as_tensor(_DeviceWord(ptr.value), device=dev)
    return flag


def f16_range_exceeded(device=None, clear=True):
    """True when the flag is set (one 4-byte read: a host synchronisation); clears it by default.  `describe_range(bits)` names
    the producers."""
    flag = f16_range_flag(device)
    bits = int(flag.item())
    if bits and clear:
        flag.zero_()
    return bits


def describe_range(bits):
    return ", ".join(name for bit, name in _hip.RANGE_TAGS.items() if bits & bit) or "none"


class SplitWeight:
    """A weight's split-precision operand: `t` the (R, K/32, 2, 32) 16-bit planes, `fmt` their element format and (PAIR_F16)
    `scale` the 4 device floats vrd_split_weight wrote: [0] = 2^-(e_w + F16_ACT_EXP), the GEMM's accumulator factor."""
    __slots__ = ("t", "fmt", "scale")

    def __init__(self, t, fmt, scale=None):
        self.t, self.fmt, self.scale = t, fmt, scale

    def data_ptr(self):
        return self.t.data_ptr()

    def set_args(self, a):
        """W_split, split_fmt and w_scale of a GemmArgs"""
        a.W_split, a.split_fmt = self.t.data_ptr(), self.fmt
        a.w_scale = self.scale.data_ptr() if self.scale is not None else None


def _split_weight(w, offset, R, Q, taps, sr, st, sq, fmt=None):
    assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and (taps * Q) % 32 == 0
    fmt = pair_fmt() if fmt is None else fmt
    f16 = fmt == _hip.PAIR_F16
    out = torch.empty(R, taps * Q // 32, 2, 32, device=w.device, dtype=torch.float16 if f16 else torch.bfloat16)
    scale = torch.empty(4, device=w.device, dtype=torch.float32) if f16 else None
    _hip.check(lib.vrd_split_weight(w.data_ptr() + 4 * offset, R, Q, taps, sr, st, sq, out.data_ptr(), fmt, _ptr(scale), _stream()),
               "vrd_split_weight")
    return SplitWeight(out, fmt, scale)


def _split_slot(base, fmt=None):
    """the cache attribute of a weight's split operand: one per element format"""
    return base + ("_f16" if (pair_fmt() if fmt is None else fmt) == _hip.PAIR_F16 else "")


def split_conv_weight(w, fmt=None):
    """SplitWeight of the tap-major packed weight (K = Cin*k, K % 32 == 0): (N, K/32, 2, 32) 16-bit, per block of 32 K
    positions [32 x hi | 32 x lo] -- the pair-row block format of vrd_common.h, so one 128-byte line holds what a K step
    needs from a weight row -- in the current mode's element format.  Built once per weight, weight version and format."""
    N, Cin, k = w.shape
    return _cached(w, _split_slot("_vrd_split", fmt), lambda: _split_weight(w.detach(), 0, N, Cin, k, Cin * k, 1, k, fmt))


def split_conv_weight_dgrad(w, fmt=None):
    """The same operand for the input-gradient GEMM of the conv: the (Cin, k*N) matrix [c][tap*N + n] = w[n][c][k-1-tap]
    (transposed, taps flipped), straight from the parameter, in the backward GEMMs' element format (backward_fmt; `fmt`: the
    format the op's forward ran under, autograd.Linear)."""
    N, Cin, k = w.shape
    fmt = backward_fmt() if fmt is None else fmt
    return _cached(w, _split_slot("_vrd_split_t", fmt), lambda: _split_weight(w.detach(), k - 1, Cin, N, k, k, -1, Cin * k, fmt))


# ---- all split operands of a training step in one launch
_presplit_scope = None     # token of the recording scope that is open (train_graph), None outside
_presplit_on = os.environ.get("VRDONE_PRESPLIT", "1") != "0"        # A/B switch


class presplit_scope:
    """`with ops.presplit_scope():` around the graph captures of one recording: operands that presplit_weights builds inside
    are accepted by the captures of the same scope (the backward graph reads what the forward graph's launch wrote).
    `plans` collects every plan a capture inside the scope used: the captured launches hold only raw device pointers into
    its job table, chunk tables and operand buffers, so the recording keeps this list for as long as its graphs live
    (presplit_weights drops a model's plans of other modes / replaced parameters from the model's own dict)."""

    def __enter__(self):
        global _presplit_scope
        self.plans = []
        self.prev, _presplit_scope = _presplit_scope, self
        return self

    def __exit__(self, *exc):
        global _presplit_scope
        _presplit_scope = self.prev
        return False


def presplit_weights(weights, plans):
    """The split-precision operands of `weights` (Conv1d parameters (N, Cin, k)) -- the forward operand where Cin*k % 32 == 0
    and the input-gradient operand where N*k % 32 == 0 -- built by ONE vrd_split_weights launch into persistent buffers and
    left in the weights' operand caches (the attributes split_conv_weight / split_conv_weight_dgrad look at) for the current
    version of each weight.  A training step otherwise
    re-splits every weight with a launch of its own, twice (after every optimiser update): 242 launches on the 24-pair batch.
    plans: a dict the caller owns (the model's): tuple of weight addresses -> (job table on the device, chunk tables, the
    persistent operand buffers).  Plans of another mode or of replaced parameters are dropped from it here; a graph
    recording that captured a plan's launch keeps its own reference (presplit_scope.plans), so the buffers its replays
    read and write outlive the dict entry.
    The job table is built (and uploaded) on the first call for a set of weights -- outside any graph capture: inside one,
    without a table, nothing is done and the per-weight launches run as before."""
    if _precision not in ("bf16x3", "f16x3") or not weights or not _presplit_on:
        return
    fmt = pair_fmt()
    # (the plan is keyed on the weights' addresses, shapes and the element format: another model whose parameters land on
    # the same addresses with other shapes, or a change of mode, builds its own)
    key = (fmt, backward_fmt()) + tuple((w.data_ptr(), tuple(w.shape)) for w in weights)
    for old in [k for k in plans if k != key]:        # operands of another mode / of parameters that were replaced
        del plans[old]
    plan = plans.get(key)
    if plan is None:
        if _capturing():
            return
        import ctypes as C
        jobs, outs, chunk_job, chunk_index = [], [], [], []
        for w in weights:
            assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 3
            N, Cin, k = w.shape
            forms = []
            if (Cin * k) % 32 == 0:
                forms.append((_split_slot("_vrd_split"), 0, N, Cin, k, Cin * k, 1, k))           # = split_conv_weight
            if (N * k) % 32 == 0:       # = split_conv_weight_dgrad: the backward GEMMs' format
                forms.append((_split_slot("_vrd_split_t", backward_fmt()), k - 1, Cin, N, k, k, -1, Cin * k))
            for slot, offset, R, Q, taps, sr, st, sq in forms:
                jf = backward_fmt() if slot.startswith("_vrd_split_t") else fmt
                jf16 = jf == _hip.PAIR_F16
                out = torch.empty(R, taps * Q // 32, 2, 32, device=w.device, dtype=torch.float16 if jf16 else torch.bfloat16)
                scale = torch.empty(4, device=w.device, dtype=torch.float32) if jf16 else None
                jobs.append(_hip.SplitJob(src=w.data_ptr() + 4 * offset, out=out.data_ptr(), R=R, Q=Q, taps=taps, fmt=jf, sr=sr, st=st,
                                          sq=sq, scale=_ptr(scale)))
                n_tiles = -(-R // 32) * (taps * Q // 32)          # 32 x 32 tiles: row block x K block
                chunk_job.extend([len(jobs) - 1] * n_tiles)
                chunk_index.extend(range(n_tiles))
                outs.append((w.data_ptr(), slot, SplitWeight(out, jf, scale)))
        if not jobs:
            return
        raw = bytes((_hip.SplitJob * len(jobs))(*jobs))
        dev = weights[0].device
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        plan = (table, torch.tensor(chunk_job, dtype=torch.int32, device=dev), torch.tensor(chunk_index, dtype=torch.int32, device=dev), outs,
                len(jobs))
        plans[key] = plan
    table, cj, ci, outs, n_jobs = plan
    if _capturing() and _presplit_scope is not None and not any(q is plan for q in _presplit_scope.plans):
        _presplit_scope.plans.append(plan)
    by_ptr = {w.data_ptr(): w for w in weights}
    if not _capturing():        # nothing moved since the last call (an evaluation of the same weights, a second forward): keep the operands
        def current(ptr, slot):
            hit = getattr(by_ptr[ptr], slot, None)
            return hit is not None and hit[0] == (ptr, by_ptr[ptr]._version) and hit[1] is not None
        if all(current(ptr, slot) for ptr, slot, _ in outs):
            return
    _hip.check(lib.vrd_split_weights(table.data_ptr(), n_jobs, cj.data_ptr(), ci.data_ptr(), cj.numel(), _stream()), "vrd_split_weights")
    for ptr, slot, out in outs:
        w = by_ptr[ptr]
        setattr(w, slot, ((ptr, w._version), out, _presplit_scope if _capturing() else None))


def bct_to_btc(x, c0, count, out, pair=False, frames=None, index=None):
    """channels [c0, c0+count) of x (B, C, T) -> out (B, T, count-wide slab).
    frames: only the first `frames` of the T frames; index (n,) int32 device: out sequence i = x[index[i]] (n sequences)."""
    Bx, Ct, Tx = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32 and x.is_cuda
    T = Tx if frames is None else frames
    B = Bx if index is None else index.numel()
    assert index is None or (index.dtype == torch.int32 and index.is_cuda and index.is_contiguous())
    p, rows, cols, ld = _rows(out)
    assert rows == B * T and cols == count and T <= Tx
    fmt = _fmt(pair)
    _hip.check(lib.vrd_bct_to_btc(x.data_ptr(), B, Ct, T, c0, count, p, ld, fmt, Tx, _ptr(index), _stream()), "vrd_bct_to_btc")
    return Pair(out, count, fmt) if fmt else out


def pair_table(feats):
    """Device pointer table + lengths of the dataloader's per-pair feature list (one upload for the whole video,
    done before any kernel is queued: a host->device copy waits for everything already queued on the GPU).
    Every feats[i] must be the (C_in, L) view of a contiguous (L, C_in) float32 matrix on the device."""
    dev = feats[0].device
    C_in = feats[0].shape[0]
    for f in feats:
        if not (f.is_cuda and f.dtype == torch.float32 and f.shape[0] == C_in and f.stride(0) == 1 and
                (f.shape[1] == 1 or f.stride(1) == C_in)):
            return None
    return (torch.tensor([f.data_ptr() for f in feats], dtype=torch.int64, device=dev),
            torch.tensor([f.shape[1] for f in feats], dtype=torch.int32, device=dev))


def pack_pairs(table, lens, T, V, Cc, S, E, pair_wide):
    """Batch the pairs described by `table` (B device pointers) / `lens` (B) straight into the backbone's
    channels-last operand buffers (see vrd_pack_pairs).
    Returns (vis (2B,T,V) tensor|Pair, clip or None, so_box (B,T,S), ent (2B,T,E), mask (B,T) bool)."""
    B = table.shape[0]
    dev = table.device
    assert table.dtype == torch.int64 and lens.dtype == torch.int32 and table.is_contiguous() and lens.is_contiguous()
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)      # noqa: E731
    vis, so_box, ent = new(2 * B, T, V), new(B, T, S), new(2 * B, T, E)
    clip = new(2 * B, T, Cc) if Cc else None
    a = _hip.PackArgs()
    a.src, a.lens = table.data_ptr(), lens.data_ptr()
    a.P, a.C_in, a.T, a.V, a.Cc, a.S, a.E = B, 2 * V + 2 * Cc + S + 2 * E, T, V, Cc, S, E
    a.vis, a.clip, a.so_box, a.ent = vis.data_ptr(), _ptr(clip), so_box.data_ptr(), ent.data_ptr()
    a.pair_wide = _fmt(pair_wide)
    _hip.check(lib.vrd_pack_pairs(C.byref(a), _stream()), "vrd_pack_pairs")
    mask = torch.arange(T, device=dev)[None, :] < lens[:, None]
    if a.pair_wide:
        vis = Pair(vis, V, a.pair_wide)
        clip = Pair(clip, Cc, a.pair_wide) if Cc else None
    return vis, clip, so_box, ent, mask


def gather_pairs(source, sel, T, S, E, pair_wide):
    """Batch the pairs `sel` (device int64 indices into a proposals.PairSource) straight from the per-tracklet rows into
    the backbone's operand buffers, computing the box features on the way (vrd_gather_pairs).
    Returns (vis, clip or None, so_box, ent, mask) like pack_pairs."""
    return gather_rows(source, source.s_row[sel].contiguous(), source.o_row[sel].contiguous(), source.lens_dev[sel].contiguous(),
                       T, S, E, pair_wide)


def gather_rows(source, s_row, o_row, lens, T, S, E, pair_wide, boxes_only=False):
    """gather_pairs for explicit tables: sequence p = lens[p] frames starting at rows s_row[p] (subject half of the
    outputs) and o_row[p] (object half) of the source's per-tracklet arrays, stepping by the source's stride.
    boxes_only: the wide visual / clip rows are not gathered (returned as None)."""
    assert S == 5 and E == 8, "the reference's box features are 5 (pair) + 8 (entity) channels (utils/misc.py:158-217)"
    B = lens.shape[0]
    dev = source.vis.device
    V, Cc = source.n_visual, source.n_clip
    assert s_row.dtype == o_row.dtype == torch.int64 and lens.dtype == torch.int32 and s_row.shape == o_row.shape == (B,)
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)      # noqa: E731
    so_box, ent = new(B, T, S), new(2 * B, T, E)
    vis = None if boxes_only else new(2 * B, T, V)
    clip = new(2 * B, T, Cc) if Cc and not boxes_only else None
    a = _hip.GatherArgs()
    a.vis, a.clip, a.boxes = source.vis.data_ptr(), _ptr(source.clip), source.boxes.data_ptr()
    a.s_row, a.o_row, a.lens = s_row.data_ptr(), o_row.data_ptr(), lens.data_ptr()
    a.P, a.T, a.V, a.Cc, a.stride = B, T, V, Cc, source.stride
    a.w, a.h = source.wh
    a.out_vis, a.out_clip, a.out_so_box, a.out_ent = _ptr(vis), _ptr(clip), so_box.data_ptr(), ent.data_ptr()
    a.pair_wide = _fmt(pair_wide)
    _hip.check(lib.vrd_gather_pairs(C.byref(a), _stream()), "vrd_gather_pairs")
    mask = torch.arange(T, device=dev)[None, :] < lens[:, None]
    if a.pair_wide and not boxes_only:
        vis = Pair(vis, V, a.pair_wide)
        clip = Pair(clip, Cc, a.pair_wide) if Cc else None
    return vis, clip, so_box, ent, mask


def assemble_pairs(streams, snippets, stream_row, lens, T, piece, reach):
    """(2P, T, D) entity-stage rows of P pairs from rows computed once per tracklet (`streams`, any shape (..., D);
    stream_row (2P,) int64 = row of frame 0 of each subject then object) and the window-edge pieces `snippets`
    (4P, L, D) of `piece` frames; see vrd_assemble_args."""
    P = lens.shape[0]
    D = streams.shape[-1]
    L = snippets.shape[1]
    assert snippets.shape == (4 * P, L, D) and stream_row.shape == (2 * P,) and stream_row.dtype == torch.int64
    assert lens.dtype == torch.int32 and streams.is_contiguous() and snippets.is_contiguous()
    out = torch.empty(2 * P, T, D, device=streams.device, dtype=torch.float32)
    a = _hip.AssembleArgs()
    a.streams, a.snippets, a.stream_row, a.lens = streams.data_ptr(), snippets.data_ptr(), stream_row.data_ptr(), lens.data_ptr()
    a.P, a.T, a.D, a.L, a.piece, a.reach = P, T, D, L, piece, reach
    a.out = out.data_ptr()
    _hip.check(lib.vrd_assemble_pairs(C.byref(a), _stream()), "vrd_assemble_pairs")
    return out


def btc_to_bct(x):
    """(B, T, C) channels-last -> new (B, C, T) tensor."""
    if recording(x):
        from . import autograd
        return autograd.FromChannelsLast.apply(x)
    B, T, Cc = x.shape
    p, rows, cols, ld = _rows(x)
    out = torch.empty(B, Cc, T, device=x.device, dtype=torch.float32)
    _hip.check(lib.vrd_btc_to_bct(p, ld, B, Cc, T, out.data_ptr(), _stream()), "vrd_btc_to_bct")
    return out


def to_channels_last(x):
    """(B, C, T) -> (B, T, C)."""
    if recording(x):
        from . import autograd
        return autograd.ToChannelsLast.apply(x)
    B, Cc, T = x.shape
    out = torch.empty(B, T, Cc, device=x.device, dtype=torch.float32)
    return bct_to_btc(x.contiguous(), 0, Cc, out)


_skip_padding = os.environ.get("VRDONE_SKIP_PADDING", "1") != "0"
SKIP_MIN_ROWS = 65536        # below this the 256 x 256 GEMM kernel (the one that takes the block list) is not used


def row_blocks(mask):
    """Padding map of a (B, T) validity mask for vrd_gemm: (order, n_active per segment, segment length) -- two device
    int32 tensors made by vrd_row_blocks -- or None when the rows do not split into aligned 32-row blocks.  Cached on
    the mask tensor object (keyed on its address and version counter), so it dies with it."""
    hit = getattr(mask, "_vrd_row_blocks", None)
    key = (mask.data_ptr(), mask._version)
    capturing = _capturing()
    if hit is not None and hit[0] == key and not capturing:
        return hit[1]
    rows = mask.numel()
    val = None
    if rows % 32 == 0 and mask.is_contiguous() and mask.data_ptr() % 16 == 0 and mask.dtype in (torch.bool, torch.uint8):
        # about eight segments (one per XCD of the GEMM's tile order), each a whole number of 256-row tiles
        nblk = rows // 32
        seg_len = 8 * (((nblk + 7) // 8 + 7) // 8)
        order = torch.empty(nblk, device=mask.device, dtype=torch.int32)
        count = torch.empty((nblk + seg_len - 1) // seg_len, device=mask.device, dtype=torch.int32)
        _hip.check(lib.vrd_row_blocks(mask.data_ptr(), rows, seg_len, order.data_ptr(), count.data_ptr(), _stream()),
                   "vrd_row_blocks")
        val = (order, count, seg_len)
    if capturing:
        return _keep_for_capture(val)
    mask._vrd_row_blocks = (key, val)
    return val


def conv_gemm(x, weight, bias=None, *, act=ACT_NONE, row_mask=None, scale=None, res=None, res_masked=False,
              res2=None, out=None, out_pair=False, skip_rows=None, row_scale=None, _launch=None, _dgrad=False, _split_fmt=None,
              _a_scale=None, _bfmt=None):
    """Dense Conv1d (k = 1 or 3, stride 1, zero padding k//2) with the fused epilogue of
    vrd_gemm.  x: (B, T, Cin) tensor or Pair; weight: the Conv1d parameter (N, Cin, k).
    out_pair: write the result as pair rows of width N (returns a Pair).
    skip_rows: validity mask of the rows; aligned 32-row blocks without a valid row skip the contraction (exact with
    row_mask, which is then the default; without row_mask those rows hold bias-only filler, so pass it only where
    no valid row ever reads a padded one: projections feeding masked attention, an MLP's hidden layer).
    row_scale (rows,): per-row factor on the branch term (stochastic depth, blocks.py:1107-1120); autograd path only.
    _split_fmt: element format of the split products instead of the mode's (the unfused backward GEMMs pass PAIR_BF16; 0 = exact
    f32 products whatever the mode, None = the mode's format).
    _dgrad / _a_scale: the conv's input-gradient GEMM on the parameter's transposed operand; x is then a gradient, and in the
    f16 format its rows are split at the power-of-two factor of grad_scale(x) (vrd_gemm_args.a_scale).
    Under autograd (`recording`) the op runs as autograd.conv_gemm and returns a fresh tensor (`out` is ignored)."""
    # row_scale (stochastic depth, sampled whenever the model trains) lives in the autograd form's epilogue: it goes there even
    # when nothing of this call needs a gradient (a frozen sub-module under requires_grad_(False), reference blocks.py:1107-1120
    # drops paths regardless)
    if _launch is None and not isinstance(x, Pair) and (recording(x, weight, bias, scale, res, res2) or row_scale is not None):
        from . import autograd
        assert not out_pair
        return autograd.conv_gemm(x, weight, bias, act=act, row_mask=row_mask, scale=scale, row_scale=row_scale, res=res,
                                  res_masked=res_masked, res2=res2)
    assert row_scale is None, "row_scale (drop-path sampling) exists for plain f32 rows only"
    N, Cin, k = weight.shape
    if _dgrad:          # the conv's input gradient: weight (Cin_of_x... = N_w, Cin_w, k) acts as the (Cin_w, N_w, k) flipped conv
        N, Cin = Cin, N
    x_fmt = x.fmt if isinstance(x, Pair) else 0
    x, a_width = _unwrap(x)
    pa, rows, cols, lda = _rows(x)
    assert cols == Cin, f"input has {cols} channels, weight expects {Cin}"
    T = x.shape[-2]
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    pc, rows_c, cols_c, ldc = _rows(out)
    assert rows_c == rows and cols_c == N
    a = _hip.GemmArgs()
    if _dgrad:
        # split-precision only (autograd.Linear checks): the operand is built from the parameter in one launch; the f32
        # weight pointer is not read by the split-precision kernels and points at the same buffer
        assert (Cin * k) % 32 == 0 and not a_width
        wt = split_conv_weight_dgrad(weight, _bfmt)
        a.W = wt.data_ptr()
        wt.set_args(a)
        if wt.fmt == _hip.PAIR_F16:      # x is a gradient: its f16 planes need its own power-of-two factor
            assert _a_scale is not None
            a.a_scale = _a_scale.data_ptr()
    else:
        a.W = _param_ptr(packed_conv_weight(weight), x, "conv weight")
    a.A, a.lda, a.bias = pa, lda, _param_ptr(bias, x, "conv bias")
    a.C, a.ldc = pc, ldc
    a.M, a.N, a.Cin, a.taps, a.T = rows, N, Cin, k, T
    a.act = act
    a.row_mask = _mask_ptr(row_mask, rows)
    a.scale = _param_ptr(scale, x, "drop-path scale")
    if a_width:
        assert pair_fmt() and x_fmt == pair_fmt() and Cin % 32 == 0, "pair input needs the split-precision mode it was made in and Cin % 32 == 0"
    w_fmt = pair_fmt() if _split_fmt is None else _split_fmt
    if _dgrad:
        w_fmt = a.split_fmt
    assert not (a_width and w_fmt != x_fmt) and not (out_pair and w_fmt != pair_fmt())
    if not _dgrad and w_fmt and (Cin * k) % 32 == 0:
        split_conv_weight(weight, w_fmt).set_args(a)
    a.a_pair_width = a_width
    a.c_pair = _fmt(out_pair)
    if skip_rows is None:
        skip_rows = row_mask
    # (the kernels that take the block list: the 256 x 256 split kernel -- pair-row input -- and the exact-f32 kernel)
    f32_kernel = not a_width and not _dgrad and not (w_fmt and (Cin * k) % 32 == 0)
    if _skip_padding and skip_rows is not None and (a_width or f32_kernel) and rows >= SKIP_MIN_ROWS and skip_rows.numel() == rows:
        blocks = row_blocks(skip_rows)
        if blocks is not None:
            a.row_blocks, a.row_blocks_active = blocks[0].data_ptr(), blocks[1].data_ptr()
            a.row_block_seg_len = blocks[2]
    if res is not None:
        pr, rr, rc, ldr = _rows(res)
        assert rr == rows and rc == N
        a.res, a.ldres, a.res_masked = pr, ldr, 1 if res_masked else 0
    if res2 is not None:
        pr, rr, rc, ldr = _rows(res2)
        assert rr == rows and rc == N
        a.res2, a.ldres2 = pr, ldr
    if _launch is not None:            # conv_gemm_batch collects the argument structs instead of launching
        _launch.append(a)
    else:
        _hip.check(lib.vrd_gemm(C.byref(a), _stream()), "vrd_gemm")
    return Pair(out, N, a.c_pair) if a.c_pair else out


def conv_gemm_batch(calls):
    """Several conv_gemm calls -- [(args, kwargs), ...], at most 4 -- handed to the library together (vrd_gemm_batch):
    GEMMs that differ only in input, weight, bias and output, like the q / k / v projections of an attention block, run
    as one launch.  Returns the list of results."""
    assert 1 <= len(calls) <= 4
    if torch.is_grad_enabled():         # differentiable path: one op per projection
        return [conv_gemm(*args, **kwargs) for args, kwargs in calls]
    collected, results = [], []
    for args, kwargs in calls:
        results.append(conv_gemm(*args, _launch=collected, **kwargs))
    arr = (_hip.GemmArgs * len(collected))(*collected)
    _hip.check(lib.vrd_gemm_batch(arr, len(collected), _stream()), "vrd_gemm_batch")
    return results


def layernorm(x, gamma, beta, *, relu=False, post_add=None, out=None, pair=False):
    """Channel LayerNorm.  post_add: (period, C) rows added after the affine, row r gets
    post_add[r % period].  pair: write pair rows (returns a Pair)."""
    if recording(x, gamma, beta, post_add):
        from . import autograd
        assert not pair
        return autograd.layernorm(x, gamma, beta, relu=relu, post_add=post_add)
    px, rows, cols, ldx = _rows(x)
    if out is None:
        out = torch.empty(*x.shape, device=x.device, dtype=torch.float32)
    py, rows_y, cols_y, ldy = _rows(out)
    assert rows_y == rows and cols_y == cols
    pa, lda, period = None, 0, 0
    if post_add is not None:
        pa, period, ca, lda = _rows(post_add)
        assert ca == cols
    assert gamma.numel() == cols and beta.numel() == cols
    _hip.check(lib.vrd_layernorm(px, ldx, py, ldy, rows, cols, _param_ptr(gamma, x, "LayerNorm weight"),
                                 _param_ptr(beta, x, "LayerNorm bias"), 1 if relu else 0,
                                 pa, lda, period, _fmt(pair), _stream()), "vrd_layernorm")
    return Pair(out, cols) if _fmt(pair) else out


CONV_LN_MAXK = 32
_conv_ln_on = os.environ.get("VRDONE_CONV_LN", "1") != "0"


def conv_ln_ok(x, weight, *tensors):
    """The fused few-channel conv -> LayerNorm row kernel takes this call: inference, f32 rows, taps * Cin <= 32, 256 / 512 outputs."""
    N, Cin, k = weight.shape
    return (_conv_ln_on and not isinstance(x, Pair) and k in (1, 3) and Cin * k <= CONV_LN_MAXK and (3 + k) * Cin <= 64 and N in (256, 512) and
            not recording(x, weight, *tensors))


def conv_ln(x, weight, bias, *, row_mask=None, gamma=None, beta=None, relu=False, out=None, pair=False):
    """Dense Conv1d with few input channels (k = 1 or 3, zero padding k // 2) * row_mask -> [LayerNorm(gamma, beta)] -> [ReLU] as one
    row kernel (vrd_conv_ln): the box-feature embeddings.  x: (B, T, Cin) f32 rows; out: (B, T, N) buffer or column slab of one;
    pair: write pair rows (returns a Pair)."""
    N, Cin, k = weight.shape
    px, rows, cols, ldx = _rows(x)
    assert cols == Cin and weight.is_contiguous()
    if out is None:
        out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    py, rows_y, cols_y, ldy = _rows(out)
    assert rows_y == rows and cols_y == N
    a = _hip.ConvLnArgs()
    a.x, a.ldx, a.rows = px, ldx, rows
    a.Cin, a.taps, a.T, a.N = Cin, k, x.shape[-2], N
    a.w, a.bias = _param_ptr(weight, x, "conv weight"), _param_ptr(bias, x, "conv bias")
    a.row_mask = _mask_ptr(row_mask, rows)
    a.gamma, a.beta = _param_ptr(gamma, x, "LayerNorm weight"), _param_ptr(beta, x, "LayerNorm bias")
    a.relu = 1 if relu else 0
    a.y, a.ldy, a.out_pair = py, ldy, _fmt(pair)
    _hip.check(lib.vrd_conv_ln(C.byref(a), _stream()), "vrd_conv_ln")
    return Pair(out, N) if a.out_pair else out


_CONST_ROWS = {}


def _const_row(n, fill, device):
    """A constant vector (the stand-in of an absent bias / LayerNorm affine in a parameter image), made once per size: in a
    training step the images are rebuilt for every weight, which was ~170 fill launches per step."""
    key = (n, float(fill), str(device))
    t = _CONST_ROWS.get(key)
    if t is None:
        t = _CONST_ROWS[key] = torch.full((n,), fill, device=device, dtype=torch.float32)
    return t


def _dwconv_block(w, bias, gamma, beta):
    """The on-chip parameter image of one dwconv_ln set (vrd_dwconv_ln_args.packed): taps tap-major | bias | gamma |
    beta.  Cached on the weight, keyed on the (address, version) of all four tensors."""
    parts = (w, bias, gamma, beta)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in parts)
    hit = getattr(w, "_vrd_dw_block", None)
    capturing = _capturing()
    if hit is not None and hit[0] == key and not capturing:
        return hit[1]
    Cout, g, k = w.shape
    one = lambda t, fill: _const_row(Cout, fill, w.device) if t is None else t.detach().float().reshape(-1)  # noqa: E731
    val = torch.cat([w.detach().float().permute(1, 2, 0).reshape(-1), one(bias, 0.0), one(gamma, 1.0), one(beta, 0.0)]).contiguous()
    if capturing:
        return _keep_for_capture(val)
    w._vrd_dw_block = (key, val)
    return val


def dwconv_ln(x, sets, *, mask_out=None, stride=1, x_up=None, pre_ln=None, segs=None):
    """Fused depthwise conv * mask -> LayerNorm for up to three weight sets sharing x.
    sets: list of dicts(weight=(C, g, k) Conv1d weight, bias=None, gamma=None, beta=None, relu=False, out=None).
    pre_ln = (gamma, beta): the input rows are LayerNorm'ed as they are read (the block's ln1).
    segs (inference path): x is a ragged row space (1, R, C) -- [(first row, sequences, frames)] groups of sequences back to
    back (vrd_row_segs); outputs, mask_out and x_up follow the same grouping at their own frame counts.
    Returns the list of outputs, each (B, T/stride, C)."""
    if recording(x, x_up, *(t for st in sets for t in (st["weight"], st.get("bias"), st.get("gamma"), st.get("beta"))),
                 *(pre_ln or ())):
        from . import autograd
        return autograd.dwconv_ln(x, sets, mask_out=mask_out, stride=stride, x_up=x_up, pre_ln=pre_ln)
    B, Tin, Cx = x.shape
    w0 = sets[0]["weight"]
    Cout, g, k = w0.shape
    px, rows, cols, ldx = _rows(x)
    assert cols == Cout * g
    Tout = Tin // stride
    a = _hip.DwconvLnArgs()
    seg_table = None
    if segs is not None:
        assert B == 1 and sum(n * T for _, n, T in segs) == Tin
        seg_table = _hip.RowSegs.of(segs)
        a.segs = C.pointer(seg_table)
    a.x, a.ldx = px, ldx
    if x_up is not None:
        pu, ru, cu, ldu = _rows(x_up)
        assert cu == cols and ru * 2 == rows
        a.x_up, a.ldx_up = pu, ldu
    a.B, a.Tin, a.C, a.ksize, a.stride, a.group_in = B, Tin, Cout, k, stride, g
    if pre_ln is not None:
        assert g == 1 and x_up is None
        a.pre_gamma, a.pre_beta = _param_ptr(pre_ln[0], x, "LayerNorm weight"), _param_ptr(pre_ln[1], x, "LayerNorm bias")
    a.mask_out = _mask_ptr(mask_out, B * Tout)
    a.n_out = len(sets)
    outs = []
    for i, s in enumerate(sets):
        assert tuple(s["weight"].shape) == (Cout, g, k) and s["weight"].is_contiguous()
        o = s.get("out")
        if o is None:
            o = torch.empty(B, Tout, Cout, device=x.device, dtype=torch.float32)
        po, ro, co, ldo = _rows(o)
        assert ro == B * Tout and co == Cout
        a.w[i], a.bias[i] = _param_ptr(s["weight"], x, "depthwise weight"), _param_ptr(s.get("bias"), x, "depthwise bias")
        a.gamma[i], a.beta[i] = (_param_ptr(s.get("gamma"), x, "LayerNorm weight"),
                                 _param_ptr(s.get("beta"), x, "LayerNorm bias"))
        a.packed[i] = _dwconv_block(s["weight"], s.get("bias"), s.get("gamma"), s.get("beta")).data_ptr()
        a.relu[i] = 1 if s.get("relu") else 0
        a.y[i], a.ldy[i] = po, ldo
        a.out_pair[i] = _fmt(s.get("pair"))
        outs.append(Pair(o, Cout) if _fmt(s.get("pair")) else o)
    _hip.check(lib.vrd_dwconv_ln(C.byref(a), _stream()), "vrd_dwconv_ln")
    return outs


def _rel_pe_ptr(rel_pe, like, n_head, half_win):
    if rel_pe is None:
        return None
    assert rel_pe.numel() == n_head * (2 * half_win + 1) and rel_pe.dtype == torch.float32 and rel_pe.is_contiguous()
    assert rel_pe.device == like.device
    return rel_pe.data_ptr()


def local_attention(q, k, v, mask, n_head, half_win, pair=False, rel_pe=None, out=None, segs=None):
    """rel_pe: None or the module's (1, 1, n_head, window) relative position bias (reference blocks.py:739-743).
    out: (B, T, C) contiguous rows to write instead of a fresh tensor (inference path).
    segs (inference path): q / k / v / mask are a ragged row space (1, R, ...): [(first row, sequences, frames)] (vrd_row_segs)."""
    if recording(q, k, v, rel_pe):
        from . import autograd
        return autograd.LocalAttention.apply(q, k, v, mask, n_head, half_win, rel_pe)
    B, T, Cc = q.shape
    rel = _rel_pe_ptr(rel_pe, q, n_head, half_win)
    pq, rows, cols, ld = _rows(q)
    pk, _, _, ldk = _rows(k)
    pv, _, _, ldv = _rows(v)
    assert ld == ldk == ldv
    if out is None:
        out = torch.empty(B, T, Cc, device=q.device, dtype=torch.float32)
    assert out.shape == (B, T, Cc) and out.is_contiguous()
    if segs is not None:
        assert B == 1 and sum(n * t for _, n, t in segs) == T
        table = _hip.RowSegs.of(segs)
        _hip.check(lib.vrd_local_attn_segs(pq, pk, pv, ld, _mask_ptr(mask, rows), rel, C.byref(table), Cc, n_head, half_win,
                                           out.data_ptr(), Cc, _fmt(pair), _stream()), "vrd_local_attn_segs")
        return Pair(out, Cc) if _fmt(pair) else out
    _hip.check(lib.vrd_local_attn(pq, pk, pv, ld, _mask_ptr(mask, rows), rel, B, T, Cc, n_head, half_win,
                                  out.data_ptr(), Cc, _fmt(pair), _stream()), "vrd_local_attn")
    return Pair(out, Cc) if _fmt(pair) else out


def attention(q, k, v, kv_mask, n_head, algo=0, pair=False, q_mask=None, out=None):
    """Global masked attention; q: (B, Tq, C), k/v: (B, Tk, C); kv_mask (B, Tk) or None.
    out: (B, Tq, C) contiguous rows to write instead of a fresh tensor (inference path).
    pair: pair-row output when the flash kernel runs (otherwise a plain tensor is returned).
    q_mask (B, Tq): rows the caller masks afterwards; the split-precision kernel leaves out query tiles without a valid
    row (they read 0)."""
    if not isinstance(q, Pair) and recording(q, k, v):
        from . import autograd
        return autograd.Attention.apply(q, k, v, kv_mask, n_head)
    if isinstance(q, Pair):
        assert isinstance(k, Pair) and isinstance(v, Pair) and q.width == k.width == v.width == q.shape[-1]
        assert q.fmt == k.fmt == v.fmt
        fmt = q.fmt
        q, k, v = q.t, k.t, v.t
        B, Tq, Cc = q.shape
        Tk = k.shape[1]
        pq, _, _, ldq = _rows(q)
        pk, rows_k, _, ldk = _rows(k)
        pv, _, _, ldv = _rows(v)
        assert ldk == ldv
        if out is None:
            out = torch.empty(B, Tq, Cc, device=q.device, dtype=torch.float32)
        assert out.shape == (B, Tq, Cc) and out.is_contiguous()
        _hip.check(lib.vrd_attention_pair(pq, ldq, pk, pv, ldk, _mask_ptr(kv_mask, rows_k), _mask_ptr(q_mask, B * Tq), B, Tq, Tk, n_head,
                                          Cc // n_head, out.data_ptr(), Cc, fmt if pair else 0, fmt, _stream()),
                   "vrd_attention_pair")
        return Pair(out, Cc, fmt) if pair else out
    B, Tq, Cc = q.shape
    Tk = k.shape[1]
    pq, _, _, ldq = _rows(q)
    pk, rows_k, _, ldk = _rows(k)
    pv, _, _, ldv = _rows(v)
    assert ldk == ldv
    if out is None:
        out = torch.empty(B, Tq, Cc, device=q.device, dtype=torch.float32)
    assert out.shape == (B, Tq, Cc) and out.is_contiguous()
    hd = Cc // n_head
    flash = algo == 2 or (algo == 0 and hd in (64, 128))      # mirrors vrd_attention's auto choice
    pair = bool(pair and flash and pair_fmt())
    _hip.check(lib.vrd_attention(pq, ldq, pk, pv, ldk, _mask_ptr(kv_mask, rows_k), B, Tq, Tk, n_head, hd,
                                 out.data_ptr(), Cc, algo, _fmt(pair), _stream()), "vrd_attention")
    return Pair(out, Cc) if pair else out


def maxpool_mask(x, mask_in, out=None):
    """MaxPool1d(3, 2, 1)(x) * mask[::2]; returns (pooled (B, T/2, C), mask_out (B, T/2) bool).
    out: (pooled, mask_out) contiguous buffers to write instead of fresh tensors (inference path)."""
    if recording(x):
        from . import autograd
        return autograd.MaxPoolMask.apply(x, mask_in)
    B, T, Cc = x.shape
    px, rows, cols, ldx = _rows(x)
    if out is None:
        y = torch.empty(B, T // 2, Cc, device=x.device, dtype=torch.float32)
        m_out = torch.empty(B, T // 2, device=x.device, dtype=torch.bool)
    else:
        y, m_out = out
        assert y.shape == (B, T // 2, Cc) and y.is_contiguous() and m_out.shape == (B, T // 2) and m_out.is_contiguous()
        assert m_out.dtype == torch.bool
    _hip.check(lib.vrd_maxpool_mask(px, ldx, B, T, Cc, _mask_ptr(mask_in, rows), y.data_ptr(), Cc, m_out.data_ptr(),
                                    _stream()), "vrd_maxpool_mask")
    return y, m_out


def mask_head(emb, feat, out_mask, fill=-10.0):
    """emb (B, Q, Dp), feat (B, T, Dp), out_mask (B, T) -> (B, Q, T)."""
    if recording(emb, feat):
        from . import autograd
        return autograd.MaskHead.apply(emb, feat, out_mask, fill)
    B, Q, Dp = emb.shape
    T = feat.shape[1]
    pe, _, _, lde = _rows(emb)
    pf, rows_f, _, ldf = _rows(feat)
    seg = torch.empty(B, Q, T, device=emb.device, dtype=torch.float32)
    _hip.check(lib.vrd_mask_head(pe, lde, pf, ldf, _mask_ptr(out_mask, rows_f), B, Q, T, Dp, fill, seg.data_ptr(),
                                 _stream()), "vrd_mask_head")
    return seg


def assign_tables(sizes, dev):
    """(first row, relation count) of every pair as device int32 tensors: what vrd_assign walks."""
    first, at = [], 0
    for n in sizes:
        first.append(at)
        at += n
    return torch.tensor(first, dtype=torch.int32, device=dev), torch.tensor(sizes, dtype=torch.int32, device=dev)


def assign(cost, sizes, tables=None):
    """Minimum-cost assignment of every pair's relations to its queries on the device (vrd_assign): cost (sum N, Q) f32,
    rows grouped pair by pair, sizes = [N_p] (each <= Q <= 16); tables = assign_tables(sizes, device) when the caller
    assigns several cost matrices of the same batch.  Returns the query of every relation, (sum N,) int32."""
    assert cost.is_cuda and cost.dtype == torch.float32 and cost.dim() == 2 and cost.stride(1) == 1
    G, Q = cost.shape
    assert sum(sizes) == G and max(sizes) <= Q <= 16
    first_d, count_d = tables if tables is not None else assign_tables(sizes, cost.device)
    out = torch.full((G,), -1, dtype=torch.int32, device=cost.device)
    _hip.check(lib.vrd_assign(cost.data_ptr(), cost.stride(0), first_d.data_ptr(), count_d.data_ptr(), len(sizes), Q,
                              out.data_ptr(), _stream()), "vrd_assign")
    return out


def postprocess(logits, masks, valid_len, topk):
    """logits (P, Q, K1), masks (P, Q, T), valid_len (P,) int32 ->
    (top_score (P,Q,k) f32, top_cat (P,Q,k) i32, seg_first (P,Q) i32, seg_last (P,Q) i32)."""
    P, Q, K1 = logits.shape
    T = masks.shape[-1]
    assert logits.is_contiguous() and masks.is_contiguous() and valid_len.dtype == torch.int32
    dev = logits.device
    ts = torch.empty(P, Q, topk, device=dev, dtype=torch.float32)
    tc = torch.empty(P, Q, topk, device=dev, dtype=torch.int32)
    sf = torch.empty(P, Q, device=dev, dtype=torch.int32)
    sl = torch.empty(P, Q, device=dev, dtype=torch.int32)
    _hip.check(lib.vrd_postprocess(logits.data_ptr(), masks.data_ptr(), valid_len.data_ptr(), P, Q, K1, T, topk,
                                   ts.data_ptr(), tc.data_ptr(), sf.data_ptr(), sl.data_ptr(), _stream()),
               "vrd_postprocess")
    return ts, tc, sf, sl
