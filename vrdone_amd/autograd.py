"""Differentiable forms of the ops in vrdone_amd.ops: one torch.autograd.Function per kernel family, forward AND
backward on hand-written HIP kernels (csrc/vrd_backward.hip for the gradients).  The reference gets its gradients from
autograd over ATen ops (train.py:186, models/maskvrd.py:168-198); this is what makes `loss.backward()` work on the HIP
path.

ops.* dispatch here when autograd is recording (torch.is_grad_enabled()) and an input requires a gradient.  In that
mode every activation is a plain f32 channels-last tensor (no pair rows, no writes into concatenation slabs, no padding
skip), and a fused eval op is run as its differentiable parts, each of which saves what its backward needs:

    conv_gemm(x, W, b, act, row_mask, scale, row_scale, res, res2)
        = ScaleResidual( Activation( Linear(x, W, b, row_mask) ), scale, row_scale, row_mask, res, res2 )
    dwconv_ln(x, sets, mask_out, stride, x_up, pre_ln)
        = [ LayerNorm_o( DepthwiseConv_o( LayerNorm_pre(x) (+ up2(x_up)) ) ) ]

PyTorch is used for memory, views / concatenations of results and autograd's own bookkeeping (summing the gradients
of a tensor used twice); the arithmetic of every forward and backward op is a kernel of libvrdone_hip.so.
"""
import ctypes as C

import os

import torch
from torch.autograd import Function

from . import _hip, ops
from ._hip import PAIR_BF16, PAIR_F16, lib
from .ops import ACT_NONE, ACT_RELU, _mask_ptr, _ptr, _rows, _stream

check = _hip.check


def _new_like_rows(t, cols=None):
    return torch.empty(*t.shape[:-1], t.shape[-1] if cols is None else cols, device=t.device, dtype=torch.float32)


def _dense(t):
    """Gradients arrive with arbitrary strides (expanded, sliced): kernels want uniformly strided rows."""
    return t if (t.stride(-1) == 1 and t.is_contiguous()) else t.contiguous()


_partial_scratch = {}
_PARTIAL_SUMS = os.environ.get("VRD_PARTIAL_SUMS", "1") != "0"      # A/B switch: 0 = the gradient kernels end in float atomics


def _partials(device):
    """The buffer the gradient kernels park their workgroups' partial sums in before a second launch adds them up (users:
    vrd_gemm_wgrad_x3 -- the row chunks' partial tiles, include/vrdone_hip.h: 4 * CUs * 16,384 + N * K floats suffice --,
    vrd_layernorm_bwd, vrd_colsum and vrd_dwconv_wgrad; sized for weights of up to 4 M elements, beyond that the wgrad kernel
    falls back to atomics).  It is only ever live between two adjacent launches of ONE stream, so there is one per (device,
    stream): two backward passes on different streams of a device (two replicas in a process, autograd on a side stream) each
    get their own.  Launches recorded into a graph allocate theirs per call from the graph's private pool (a buffer cached
    from one recording's pool must not be written by another recording's replays)."""
    dev = torch.device(device)
    if not _PARTIAL_SUMS:
        key = (dev, None)
        if key not in _partial_scratch:
            _partial_scratch[key] = torch.empty(0, device=dev, dtype=torch.float32)
        return _partial_scratch[key]
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    n = 4 * cus * 16384 + (4 << 20)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, device=dev, dtype=torch.float32)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _partial_scratch.get(key)
    if buf is None:
        buf = _partial_scratch[key] = torch.empty(n, device=dev, dtype=torch.float32)
    return buf


# ---------------------------------------------------------------------------------------------- raw launchers
def colsum(a, out, *, b=None, b_cstride=1, b_coffset=0, b_rstride=1, shift=0, T=1, row_mask=None, row_scale=None):
    """out[c] += sum_r a[r,c] * b[...] * mask[r] * row_scale[r]  (vrd_colsum)."""
    pa, rows, cols, lda = _rows(a)
    pb, ldb = (None, 0)
    if b is not None:
        pb, _, _, ldb = _rows(b)
    assert out.numel() == cols and out.is_contiguous() and out.dtype == torch.float32
    part = _partials(a.device)
    check(lib.vrd_colsum(pa, lda, pb, ldb, b_cstride, b_coffset, b_rstride, shift, T, _mask_ptr(row_mask, rows),
                         _ptr(row_scale), rows, cols, out.data_ptr(), part.data_ptr(), part.numel(), _stream()), "vrd_colsum")
    return out


def rowcol_scale(v, *, col_scale=None, row_scale=None, row_mask=None, res=None, res_masked=False, res2=None):
    pv, rows, cols, ldv = _rows(v)
    out = _new_like_rows(v)
    po, _, _, ldo = _rows(out)
    pr, ldr = (None, 0) if res is None else (_rows(res)[0], _rows(res)[3])
    pr2, ldr2 = (None, 0) if res2 is None else (_rows(res2)[0], _rows(res2)[3])
    if row_scale is not None:
        assert row_scale.numel() == rows and row_scale.is_contiguous() and row_scale.dtype == torch.float32
    check(lib.vrd_rowcol_scale(pv, ldv, rows, cols, _ptr(col_scale), _ptr(row_scale), _mask_ptr(row_mask, rows), pr, ldr,
                               1 if res_masked else 0, pr2, ldr2, po, ldo, _stream()), "vrd_rowcol_scale")
    return out


def activation(x, act, dy=None):
    px, rows, cols, ldx = _rows(x)
    out = _new_like_rows(x)
    po, _, _, ldo = _rows(out)
    pd, ldd = (None, 0) if dy is None else (_rows(dy)[0], _rows(dy)[3])
    check(lib.vrd_activation(px, ldx, pd, ldd, rows, cols, act, po, ldo, _stream()), "vrd_activation")
    return out


def bmm(A, a_strides, B, b_strides, Cmat, c_strides, Z0, Z1, M, N, K, alpha=1.0, accumulate=False):
    """Strided batched matmul (vrd_bmm); *_strides = (z0, z1, row, col) in floats."""
    a = _hip.BmmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cmat.data_ptr()
    a.a_z0, a.a_z1, a.a_row, a.a_col = a_strides
    a.b_z0, a.b_z1, a.b_row, a.b_col = b_strides
    a.c_z0, a.c_z1, a.c_row, a.c_col = c_strides
    a.Z0, a.Z1, a.M, a.N, a.K, a.alpha, a.accumulate = Z0, Z1, M, N, K, alpha, 1 if accumulate else 0
    check(lib.vrd_bmm(C.byref(a), _stream()), "vrd_bmm")
    return Cmat


FUSED_ATTN_BWD = os.environ.get("VRDONE_FUSED_ATTN_BWD", "1") != "0"      # A/B switch: 0 = the five-product form everywhere


# ------------------------------------------------------------------------------------------------------ Functions
class _ZeroArena:
    """Zero-initialised accumulators for the gradient kernels (weight / bias / LayerNorm / scale gradients are summed with
    atomics into their output), carved out of 16 MiB blocks: one fill per block instead of one `torch.zeros` per gradient
    (~500 of the ~930 fills of a training step on the 24-pair batch).  A block is never handed out twice; it lives as long
    as a gradient that points into it.  A block opened during a graph capture belongs to the graph's pool and is dropped
    when the capture state changes."""
    BLOCK = 4 << 20            # floats

    def __init__(self):
        self.block, self.at, self.capturing = None, 0, False

    def take(self, n, device):
        capturing = torch.cuda.is_current_stream_capturing()
        padded = -(-n // 64) * 64                      # keep every slice 256-byte aligned
        if (self.block is None or self.block.device != device or capturing != self.capturing or
                self.at + padded > self.block.numel()):
            self.block = torch.zeros(max(self.BLOCK, padded), device=device, dtype=torch.float32)
            self.at, self.capturing = 0, capturing
        out = self.block[self.at:self.at + n]
        self.at += padded
        return out


_arena = _ZeroArena()


_USE_ARENA = os.environ.get("VRD_ZERO_ARENA", "1") != "0"


def _zeros(*shape, device):
    if not _USE_ARENA:
        return torch.zeros(*shape, device=device, dtype=torch.float32)
    n = 1
    for d in shape:
        n *= d
    return _arena.take(n, torch.device(device)).view(*shape)


class Linear(Function):
    """y = Conv1d(x; W (N, Cin, k), b) * row_mask, k in {1, 3}, on channels-last rows (vrd_gemm without epilogue terms)."""

    @staticmethod
    def forward(ctx, x, weight, bias, row_mask):
        y = ops.conv_gemm(x, weight, bias, row_mask=row_mask)
        ctx.save_for_backward(x, weight)
        ctx.row_mask, ctx.has_bias = row_mask, bias is not None
        # the backward runs in the arithmetic of ITS forward, whatever the mode is by then (MaskVRD.forward_training repeats a
        # step whose activations leave the f16 range in the f32 mode: the graph it returns is differentiated outside that block)
        ctx.split_bwd, ctx.bfmt = ops.split_backward(), ops.backward_fmt()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        mask = ctx.row_mask
        dy = _dense(dy)
        N, Cin, k = weight.shape
        pg, rows, _, ldg = _rows(dy)
        T = x.shape[-2]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # split-precision mode: the transposed (k = 3: tap-flipped) operand comes straight from the parameter in one
            # launch (ops.split_conv_weight_dgrad); as tensor expressions -- transpose, then the hi / lo split of a tensor
            # that is new every step -- it was seven launches per weight and step
            # (the parameter object itself while autograd is not recording -- the usual case in a backward pass: its operand
            # caches, which ops.presplit_weights fills for the whole step, hang on the object; a detached view has none)
            wd = weight if not torch.is_grad_enabled() else weight.detach()
            fused = (ctx.split_bwd and (N * k) % 32 == 0 and N % 4 == 0 and dy.stride(-2) % 4 == 0 and
                     dy.data_ptr() % 16 == 0 and weight.is_contiguous())
            # the unfused fallback: bf16 planes in either split mode (gradients have no fixed scale), and EXACT f32 products
            # (_split_fmt 0, not None = "the mode at hand") behind an f32 forward -- the backward of a step that was repeated in
            # f32 runs after that block has been left, in whatever mode is current by then
            gfmt = PAIR_BF16 if ctx.split_bwd else 0
            # f16x3 mode: the gradient's power-of-two factor for its f16 planes (one absmax launch, shared with the weight gradient)
            gs = ops.grad_scale(dy) if (fused and ctx.bfmt == PAIR_F16) else None
            if fused and ctx.bfmt == PAIR_F16 and gs is None:
                fused = False
            if k == 1:       # masking the rows of dy = masking the rows of dx
                if fused:
                    dx = ops.conv_gemm(dy, wd, None, row_mask=mask, _dgrad=True, _a_scale=gs, _bfmt=ctx.bfmt)
                else:
                    dx = ops.conv_gemm(dy, weight.detach().permute(1, 0, 2).contiguous(), None, row_mask=mask, _split_fmt=gfmt)
            else:            # dx[r] = sum_tap (dy * mask)[r - (tap - 1)] W[:, :, tap]: a k=3 conv with flipped, transposed taps
                g = rowcol_scale(dy, row_mask=mask) if mask is not None else dy
                if fused:
                    dx = ops.conv_gemm(g, wd, None, _dgrad=True, _a_scale=gs, _bfmt=ctx.bfmt)
                else:
                    dx = ops.conv_gemm(g, weight.detach().flip(2).permute(1, 0, 2).contiguous(), None, _split_fmt=gfmt)
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            packed = _zeros(N, k * Cin, device=dy.device)
            px, _, _, ldx = _rows(x)
            if ctx.split_bwd:
                # split-precision products like the forward GEMMs of the mode (bf16 planes; f16x3 mode: f16 planes, the gradient at
                # its own power-of-two factor); the bias gradient (exact f32 column sums) in the same pass
                if want_db:
                    db = _zeros(N, device=dy.device)
                part = _partials(dy.device)
                if ctx.bfmt == PAIR_F16:
                    if not (ctx.needs_input_grad[0] and gs is not None):
                        gs = ops.grad_scale(dy)
                else:
                    gs = None
                check(lib.vrd_gemm_wgrad_x3(pg, ldg, px, ldx, _mask_ptr(mask, rows), rows, N, Cin, k, T, packed.data_ptr(),
                                            db.data_ptr() if want_db else None, part.data_ptr(), part.numel(),
                                            gs.data_ptr() if gs is not None else None, _stream()),
                      "vrd_gemm_wgrad_x3")
                want_db = False
            else:            # exact f32 products
                check(lib.vrd_gemm_wgrad(pg, ldg, px, ldx, _mask_ptr(mask, rows), rows, N, Cin, k, T, packed.data_ptr(), _stream()),
                      "vrd_gemm_wgrad")
            # tap-major -> the Conv1d layout (N, Cin, k), with the parameter's own strides (DDP's bucket views expect them)
            dw = packed.view(N, Cin, 1) if k == 1 else packed.view(N, k, Cin).permute(0, 2, 1).contiguous()
        if want_db:
            db = colsum(dy, _zeros(N, device=dy.device), row_mask=mask)
        return dx, dw, db, None


class Activation(Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.save_for_backward(x)
        ctx.act = act
        return activation(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return activation(x, ctx.act, dy=_dense(dy)), None


class ScaleResidual(Function):
    """out = v * scale[c] * row_scale[r] * mask[r] + res * (mask if res_masked) + res2."""

    @staticmethod
    def forward(ctx, v, scale, row_scale, row_mask, res, res_masked, res2):
        ctx.save_for_backward(v, scale if scale is not None else v.new_empty(0))
        ctx.has_scale = scale is not None
        ctx.row_scale, ctx.row_mask, ctx.res_masked = row_scale, row_mask, res_masked
        return rowcol_scale(v, col_scale=scale, row_scale=row_scale, row_mask=row_mask, res=res, res_masked=res_masked, res2=res2)

    @staticmethod
    def backward(ctx, dy):
        v, scale = ctx.saved_tensors
        scale = scale if ctx.has_scale else None
        dy = _dense(dy)
        dv = ds = dres = dres2 = None
        if ctx.needs_input_grad[0]:
            dv = rowcol_scale(dy, col_scale=scale, row_scale=ctx.row_scale, row_mask=ctx.row_mask)
        if scale is not None and ctx.needs_input_grad[1]:
            ds = colsum(dy, _zeros(v.shape[-1], device=dy.device), b=v, T=1,
                        row_mask=ctx.row_mask, row_scale=ctx.row_scale).view_as(scale)
        if ctx.needs_input_grad[4]:
            dres = rowcol_scale(dy, row_mask=ctx.row_mask) if (ctx.res_masked and ctx.row_mask is not None) else dy
        if ctx.needs_input_grad[6]:
            dres2 = dy
        return dv, ds, None, None, dres, None, dres2


class LayerNormFn(Function):
    """Channel LayerNorm (+ReLU) (+post_add rows, period = post_add.shape[0])."""

    @staticmethod
    def forward(ctx, x, gamma, beta, relu, post_add):
        y = ops.layernorm(x, gamma, beta, relu=relu, post_add=post_add)
        ctx.save_for_backward(x, gamma, beta)
        ctx.relu, ctx.period = relu, None if post_add is None else post_add.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        dy = _dense(dy)
        px, rows, cols, ldx = _rows(x)
        pd, _, _, ldd = _rows(dy)
        dx = _new_like_rows(x)
        pdx, _, _, lddx = _rows(dx)
        dg = _zeros(cols, device=x.device)
        db = _zeros(cols, device=x.device)
        part = _partials(dy.device)
        check(lib.vrd_layernorm_bwd(px, ldx, pd, ldd, rows, cols, gamma.data_ptr(), beta.data_ptr(), 1 if ctx.relu else 0,
                                    pdx, lddx, dg.data_ptr(), db.data_ptr(), part.data_ptr(), part.numel(), _stream()),
              "vrd_layernorm_bwd")
        dpost = None
        if ctx.period is not None and ctx.needs_input_grad[4]:
            # y[r] += post_add[r % period]: sum dy over the rows of each residue = column sums of the (rows/period, period*C) view
            dpost = colsum(dy.reshape(rows // ctx.period, ctx.period * cols),
                           _zeros(ctx.period * cols, device=x.device)).view(ctx.period, cols)
        return dx, dg.view_as(gamma), db.view_as(beta), None, dpost


class DepthwiseConv(Function):
    """D_o = mask_out * (bias_o + depthwise_conv(x (+ up2(x_up)); w_o)) for 1..3 weight sets (vrd_dwconv_ln without
    LayerNorm).  args after the fixed ones: w_0, b_0, w_1, b_1, ... (b_i may be None)."""

    @staticmethod
    def forward(ctx, x, x_up, mask_out, stride, *wb):
        ws, bs = wb[0::2], wb[1::2]
        outs = ops.dwconv_ln(x, [dict(weight=w, bias=b) for w, b in zip(ws, bs)], mask_out=mask_out, stride=stride, x_up=x_up)
        ctx.save_for_backward(x, *( [x_up] if x_up is not None else [] ), *ws)
        ctx.has_up, ctx.mask_out, ctx.stride, ctx.has_bias = x_up is not None, mask_out, stride, [b is not None for b in bs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dDs):
        saved = ctx.saved_tensors
        x = saved[0]
        x_up = saved[1] if ctx.has_up else None
        ws = saved[2 if ctx.has_up else 1:]
        n = len(ws)
        B, Tin, Cx = x.shape
        Cout, gin, k = ws[0].shape
        s = ctx.stride
        Tout = Tin // s
        dev = x.device
        dDs = [_dense(d) if d is not None else torch.zeros(B, Tout, Cout, device=dev) for d in dDs]
        a = _hip.DwconvBwdArgs()
        for i in range(n):
            p, _, _, ld = _rows(dDs[i])
            a.dD[i], a.lddd[i], a.w[i] = p, ld, ws[i].data_ptr()
        a.n_out, a.B, a.Tin, a.C, a.ksize, a.stride, a.group_in = n, B, Tin, Cout, k, s, gin
        a.mask_out = _mask_ptr(ctx.mask_out, B * Tout)
        dx = torch.empty(B, Tin, Cx, device=dev, dtype=torch.float32)
        a.dx, a.lddx = dx.data_ptr(), Cx
        dx_up = None
        if ctx.has_up:
            dx_up = torch.empty(B, Tin // 2, Cx, device=dev, dtype=torch.float32)
            a.dx_up, a.lddx_up = dx_up.data_ptr(), Cx
        check(lib.vrd_dwconv_bwd(C.byref(a), _stream()), "vrd_dwconv_bwd")
        # the conv's input, for the weight gradients: x + nearest-x2-upsampled x_up (the copy is torch's, the add a kernel)
        xin = x if x_up is None else rowcol_scale(x, res2=x_up.repeat_interleave(2, dim=1))
        grads = []
        pxin, _, _, ldxin = _rows(xin)
        for i in range(n):
            # all taps, both inputs of a group and the bias in one pass over dD (vrd_dwconv_wgrad)
            gw = _zeros(Cout, gin, k, device=dev)          # the parameter's own layout and strides (DDP's bucket views expect them)
            gb = _zeros(Cout, device=dev) if ctx.has_bias[i] else None
            pd, rows_out, _, ldd = _rows(dDs[i])
            part = _partials(dev)
            check(lib.vrd_dwconv_wgrad(pd, ldd, pxin, ldxin, k, s, gin, Tout, _mask_ptr(ctx.mask_out, rows_out), rows_out, Cout,
                                       gw.data_ptr(), gb.data_ptr() if gb is not None else None, part.data_ptr(), part.numel(),
                                       _stream()), "vrd_dwconv_wgrad")
            grads.append(gw)
            grads.append(gb)
        return (dx, dx_up, None, None, *grads)


class LocalAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, n_head, half_win, rel_pe=None):
        rel = None if rel_pe is None else rel_pe.detach()
        out = ops.local_attention(q, k, v, mask, n_head, half_win, rel_pe=rel)
        ctx.save_for_backward(q, k, v, rel)
        ctx.mask, ctx.n_head, ctx.half_win = mask, n_head, half_win
        return out

    @staticmethod
    def backward(ctx, dO):
        q, k, v, rel = ctx.saved_tensors
        dO = _dense(dO)
        B, T, Cc = q.shape
        q, k, v = (t if t.stride(-2) == Cc and t.is_contiguous() else t.contiguous() for t in (q, k, v))
        W = 2 * ctx.half_win + 1
        dq, dk, dv = (torch.empty(B, T, Cc, device=q.device, dtype=torch.float32) for _ in range(3))
        scratch = torch.empty(2 * B * T * ctx.n_head * W, device=q.device, dtype=torch.float32)
        check(lib.vrd_local_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), Cc, dO.data_ptr(), Cc, _mask_ptr(ctx.mask, B * T),
                                     ops._rel_pe_ptr(rel, q, ctx.n_head, ctx.half_win), B, T, Cc, ctx.n_head, ctx.half_win,
                                     dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), Cc, scratch.data_ptr(), _stream()),
              "vrd_local_attn_bwd")
        d_rel = None
        if rel is not None and ctx.needs_input_grad[6]:      # the bias is added to every row's scores: d rel_pe = sum of dS
            d_rel = scratch[B * T * ctx.n_head * W:].view(B * T, ctx.n_head, W).sum(0).view(rel.shape)
        return dq, dk, dv, None, None, None, d_rel


class Attention(Function):
    """Global masked attention (f32 kernels of vrd_attention)."""

    @staticmethod
    def forward(ctx, q, k, v, kv_mask, n_head):
        B, Tq, Cc = q.shape
        Tk = k.shape[1]
        lse = None
        if (FUSED_ATTN_BWD and ops.split_backward() and Cc // n_head == 64 and Tq >= 32 and Tk >= 32 and
                q.is_contiguous() and k.is_contiguous() and v.is_contiguous()):
            # the split-precision flash forward on f32 rows (operands split while they are staged), which keeps every query's
            # log-sum-exp for the backward
            out = torch.empty_like(q)
            lse = torch.empty(B, n_head, Tq, device=q.device, dtype=torch.float32)
            check(lib.vrd_attention_rows(q.data_ptr(), Cc, k.data_ptr(), v.data_ptr(), Cc, _mask_ptr(kv_mask, B * Tk), B, Tq, Tk, n_head,
                                         Cc // n_head, ops.pair_fmt(), out.data_ptr(), Cc, lse.data_ptr(), _stream()), "vrd_attention_rows")
        else:
            out = ops.attention(q, k, v, kv_mask, n_head)
        ctx.save_for_backward(q, k, v, out)
        ctx.kv_mask, ctx.n_head, ctx.lse = kv_mask, n_head, lse
        ctx.split_bwd, ctx.bfmt = ops.split_backward(), ops.backward_fmt()          # (as in Linear: the backward follows its forward's mode)
        return out

    @staticmethod
    def backward(ctx, dO):
        q, k, v, out = (t.contiguous() for t in ctx.saved_tensors)
        dO = dO.contiguous()
        B, Tq, Cc = q.shape
        Tk, H = k.shape[1], ctx.n_head
        hd = Cc // H
        dev = q.device
        if FUSED_ATTN_BWD and ctx.split_bwd and hd == 64 and Tq >= 32 and Tk >= 32:
            # flash style (vrd_attention_bwd): the scores are recomputed tile by tile in the split of the other backward GEMMs
            # (bf16 planes; f16x3 mode: f16 planes, dO and dS at power-of-two factors from the absolute maxima of dO and v);
            # no (B, H, Tq, Tk) matrix exists
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            scratch = torch.empty(2, B, H, Tq, device=dev, dtype=torch.float32)
            so = sv = None
            if ctx.bfmt == PAIR_F16:
                so = ops.grad_scale(dO, slot=0)
                sv = ops.grad_scale(v, slot=1) if so is not None else None
                if sv is None:
                    so = None
            check(lib.vrd_attention_bwd(q.data_ptr(), Cc, k.data_ptr(), v.data_ptr(), Cc, out.data_ptr(), dO.data_ptr(), Cc,
                                        _mask_ptr(ctx.kv_mask, B * Tk), B, Tq, Tk, H, hd, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                        _ptr(ctx.lse), scratch.data_ptr(), _ptr(so), _ptr(sv), _stream()), "vrd_attention_bwd")
            return dq, dk, dv, None, None
        P = torch.empty(B, H, Tq, Tk, device=dev, dtype=torch.float32)
        dS = torch.empty(B, H, Tq, Tk, device=dev, dtype=torch.float32)
        scale = hd ** -0.5
        sc = (H * Tq * Tk, Tq * Tk)                   # z strides of P / dS
        row_q, row_k = (Tq * Cc, hd), (Tk * Cc, hd)   # z strides of (B, T, H*hd) operands
        if Tq >= 32 and Tk >= 32:
            # long sequences: scores and dP as matrix products on the matrix cores (vrd_bmm), then softmax / dS row by row
            bmm(q, (*row_q, Cc, 1), k, (*row_k, 1, Cc), P, (*sc, Tk, 1), B, H, Tq, Tk, hd, alpha=scale)       # S = scale Q K^T
            bmm(dO, (*row_q, Cc, 1), v, (*row_k, 1, Cc), dS, (*sc, Tk, 1), B, H, Tq, Tk, hd)                  # dP = dO V^T
            check(lib.vrd_attn_bwd_softmax(P.data_ptr(), dS.data_ptr(), _mask_ptr(ctx.kv_mask, B * Tk), B, Tq, Tk, H, _stream()),
                  "vrd_attn_bwd_softmax")
        else:
            check(lib.vrd_attn_bwd_probs(q.data_ptr(), Cc, k.data_ptr(), v.data_ptr(), Cc, dO.data_ptr(), Cc,
                                         _mask_ptr(ctx.kv_mask, B * Tk), B, Tq, Tk, H, hd, P.data_ptr(), dS.data_ptr(), _stream()),
                  "vrd_attn_bwd_probs")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        # dq[b, i, h, :] = scale * sum_j dS[b,h,i,j] k[b,j,h,:]
        bmm(dS, (*sc, Tk, 1), k, (*row_k, Cc, 1), dq, (*row_q, Cc, 1), B, H, Tq, hd, Tk, alpha=scale)
        # dk[b, j, h, :] = scale * sum_i dS[b,h,i,j] q[b,i,h,:]
        bmm(dS, (*sc, 1, Tk), q, (*row_q, Cc, 1), dk, (*row_k, Cc, 1), B, H, Tk, hd, Tq, alpha=scale)
        # dv[b, j, h, :] = sum_i P[b,h,i,j] dO[b,i,h,:]
        bmm(P, (*sc, 1, Tk), dO, (*row_q, Cc, 1), dv, (*row_k, Cc, 1), B, H, Tk, hd, Tq)
        return dq, dk, dv, None, None


class MaxPoolMask(Function):
    @staticmethod
    def forward(ctx, x, mask_in):
        y, m_out = ops.maxpool_mask(x, mask_in)
        ctx.save_for_backward(x)
        ctx.mask_in = mask_in
        ctx.mark_non_differentiable(m_out)
        return y, m_out

    @staticmethod
    def backward(ctx, dy, _dm):
        (x,) = ctx.saved_tensors
        dy = _dense(dy)
        B, T, Cc = x.shape
        px, _, _, ldx = _rows(x)
        pd, _, _, ldd = _rows(dy)
        dx = torch.empty(B, T, Cc, device=x.device, dtype=torch.float32)
        check(lib.vrd_maxpool_bwd(px, ldx, pd, ldd, B, T, Cc, _mask_ptr(ctx.mask_in, B * T), dx.data_ptr(), Cc, _stream()),
              "vrd_maxpool_bwd")
        return dx, None


class MaskHead(Function):
    @staticmethod
    def forward(ctx, emb, feat, out_mask, fill):
        seg = ops.mask_head(emb, feat, out_mask, fill)
        ctx.save_for_backward(emb, feat)
        ctx.out_mask = out_mask
        return seg

    @staticmethod
    def backward(ctx, dseg):
        emb, feat = (t.contiguous() for t in ctx.saved_tensors)
        B, Q, Dp = emb.shape
        T = feat.shape[1]
        g = dseg.masked_fill(~ctx.out_mask[:, None, :].bool(), 0.0).contiguous()    # filled frames carry no gradient
        demb, dfeat = torch.empty_like(emb), torch.empty_like(feat)
        # demb[b, q, :] = sum_t g[b,q,t] feat[b,t,:];  dfeat[b, t, :] = sum_q g[b,q,t] emb[b,q,:]
        bmm(g, (Q * T, 0, T, 1), feat, (T * Dp, 0, Dp, 1), demb, (Q * Dp, 0, Dp, 1), B, 1, Q, Dp, T)
        bmm(g, (Q * T, 0, 1, T), emb, (Q * Dp, 0, Dp, 1), dfeat, (T * Dp, 0, Dp, 1), B, 1, T, Dp, Q)
        return demb, dfeat, None, None


class ToChannelsLast(Function):
    """(B, C, T) -> (B, T, C); its backward is the opposite layout change."""

    @staticmethod
    def forward(ctx, x):
        return ops.to_channels_last(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.btc_to_bct(_dense(dy))


class FromChannelsLast(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.btc_to_bct(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.to_channels_last(dy.contiguous())


# ---------------------------------------------------------------------------------- differentiable ops (ops.* call these)
def conv_gemm(x, weight, bias=None, *, act=ACT_NONE, row_mask=None, scale=None, row_scale=None, res=None, res_masked=False,
              res2=None):
    v = Linear.apply(x, weight, bias, row_mask)
    if act != ACT_NONE:
        v = Activation.apply(v, act)          # act(0) = 0 for ReLU and GELU: masking before the activation = after it
    if scale is not None or row_scale is not None or res is not None or res2 is not None:
        v = ScaleResidual.apply(v, scale, row_scale, row_mask, res, res_masked, res2)
    return v


def layernorm(x, gamma, beta, *, relu=False, post_add=None):
    return LayerNormFn.apply(x, gamma, beta, relu, post_add)


def dwconv_ln(x, sets, *, mask_out=None, stride=1, x_up=None, pre_ln=None):
    if pre_ln is not None:
        x = LayerNormFn.apply(x, pre_ln[0], pre_ln[1], False, None)
    wb = []
    for s in sets:
        wb += [s["weight"], s.get("bias")]
    outs = DepthwiseConv.apply(x, x_up, mask_out, stride, *wb)
    res = []
    for s, d in zip(sets, outs):
        if s.get("gamma") is not None:
            d = LayerNormFn.apply(d, s["gamma"], s["beta"], bool(s.get("relu")), None)
        elif s.get("relu"):
            d = Activation.apply(d, ACT_RELU)
        res.append(d)
    return res
