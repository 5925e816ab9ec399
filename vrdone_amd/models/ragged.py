"""A ragged batch in ONE row space (inference only).

`MaskVRD._mask_vrd` computes the pairs of a batch at the shortest padded lengths that give the reference's outputs
(`tight padding`, models/maskvrd.py), pairs of equal padded length as one bucket.  Run bucket by bucket, every launch of
the network shrinks with its bucket, and below ~260 k rows a bucket's GEMMs fall off the 256 x 256 kernel; so the
buckets stayed few and coarse.  Here all buckets lie back to back in one row buffer per activation -- bucket i = rows
[off_i, off_i + n_i * T_i) as n_i sequences of T_i frames -- and

  * everything that works row by row -- LayerNorm, every k = 1 conv GEMM with its epilogue (bias, GELU, channel scale,
    residuals, row mask), which is the bulk of the FLOPs -- runs ONCE over all rows, whatever the bucket structure: a row's
    result does not depend on which rows share its launch;
  * the dense k = 3 convs of the embedding stage run once over all rows too: a k = 3 conv reads its two neighbour rows, and
    in a flat row space frame 0 of a sequence would read the last frame of the sequence before it, where the reference's
    Conv1d(padding=1) reads zero.  That last frame is zeroed in the conv's input.  It is read by itself, by the frame before
    it and by the next sequence's first frame: the first two are padded frames here (tight padding keeps a coarsest-level
    stride of padded frames behind every pair it shortens; pairs within one frame of their padded length are bucketed apart,
    behind the others, and their buckets' k = 3 convs run one by one) whose outputs the row mask zeroes, which makes the flat
    conv equal to the per-sequence one on every valid row;
  * only the kernels that need the (sequences, frames) structure loop over the buckets' views: depthwise conv + LayerNorm
    (stride 1 / 2, FPN upsample-add), the banded and the global attention, the pyramid's max-pool, the boundary transpose and
    the mask head.

Subject and object rows of the shared-weight stages are stacked as [all subject buckets | all object buckets], so that
either half is a contiguous row range with the same bucket layout.

The composition below follows the modules' own `cl` paths line by line (backbones.py `entity_stage` / `pair_stage`,
blocks.py `TransformerBlock.cl`, local_transformer.py `MaskedConvTransformerDecoderLayer.cl`, fpns.py, predictor.py) and
uses their parameters; reference: models/maskvrd.py:363-414 pads every pair to one length and computes every padded frame.
"""
import torch

from .blocks import _ConvAttention, _ops


_MAX_SEGS = 32          # _hip.MAX_SEGS: groups of sequences one launch takes


class Layout:
    """Where the buckets lie in the row space of one pyramid level: segs = [(first row, sequences, frames)].
    Buckets whose sequences all end in two padded frames (`flat`) come first: over their rows [0, rows_flat) the dense k = 3
    convs run as one flat launch (see the module text); the others' k = 3 convs run bucket by bucket."""

    def __init__(self, buckets):
        """buckets: [(sequences, frames, flat)]"""
        self.segs, self.flat, off = [], [], 0
        self._tails = {}
        for n, T, flat in buckets:
            self.segs.append((off, n, T))
            self.flat.append(bool(flat))
            off += n * T
        assert self.flat == sorted(self.flat, reverse=True), "buckets that take the flat k = 3 convs come first"
        self.rows = off
        self.rows_flat = sum(n * T for (_, n, T), f in zip(self.segs, self.flat) if f)

    def twice(self):
        """segs of the stacked [subject | object] rows"""
        return self.segs + [(self.rows + off, n, T) for off, n, T in self.segs]

    def tail_rows(self, halves, device):
        """last row of every sequence of the flat buckets (of both halves of a stacked row space), built ON the device: a table
        uploaded from the host in the middle of a call would wait for every launch queued before it"""
        key = (halves, str(device))
        if key not in self._tails:
            idx = [torch.arange(h * self.rows + off + T - 1, h * self.rows + off + n * T, T, dtype=torch.int64, device=device)
                   for h in range(halves) for (off, n, T), f in zip(self.segs, self.flat) if f]
            self._tails[key] = (torch.cat(idx) if len(idx) > 1 else idx[0]) if idx else None
        return self._tails[key]


def _part(x, off, n, T):
    """rows [off, off + n T) of the flat (1, R, C) operand x (tensor, Pair or None) as (n, T, C)"""
    if x is None:
        return None
    ops = _ops()
    if isinstance(x, ops.Pair):
        return ops.Pair(_part(x.t, off, n, T), x.width, x.fmt)
    return x[0, off:off + n * T].unflatten(0, (n, T))


def _mpart(m, off, n, T):
    return None if m is None else m[0, off:off + n * T].view(n, T)


def _merged(segs):
    """buckets of one frame count (the predictor's queries) are one launch"""
    if len(segs) > 1 and all(T == segs[0][2] for _, _, T in segs):
        return [(segs[0][0], sum(n for _, n, _ in segs), segs[0][2])]
    return segs


def _raw(x):
    return x.t if isinstance(x, _ops().Pair) else x


def _zero_rows(x, rows):
    """the listed rows of a flat operand (f32 rows or pair rows: all-zero bits are the value zero in both) := 0"""
    _raw(x)[0].index_fill_(0, rows, 0.0)


def _dwconv_rows(x, sets, mask_out, segs, *, stride=1, x_up=None, pre_ln=None):
    """ops.dwconv_ln over all buckets into joint output buffers; x (1, R, C) f32 rows, x_up the rows of the coarser level.
    One launch with the buckets as the kernel's row groups (vrd_row_segs)."""
    ops = _ops()
    Cout = sets[0]["weight"].shape[0]
    rows_out = sum(n * (T // stride) for _, n, T in segs)
    bufs = [torch.empty(1, rows_out, Cout, device=x.device, dtype=torch.float32) for _ in sets]
    segs = _merged(segs)
    if len(segs) == 1:
        off, n, T = segs[0]
        oo, To = off // stride, T // stride
        ops.dwconv_ln(_part(x, off, n, T), [dict(st, out=_part(b, oo, n, To)) for st, b in zip(sets, bufs)],
                      mask_out=_mpart(mask_out, oo, n, To), stride=stride, x_up=_part(x_up, off // 2, n, T // 2), pre_ln=pre_ln)
    else:
        for g0 in range(0, len(segs), _MAX_SEGS):
            part = segs[g0:g0 + _MAX_SEGS]
            r0, r1 = part[0][0], part[-1][0] + part[-1][1] * part[-1][2]
            rel = [(off - r0, n, T) for off, n, T in part]
            ops.dwconv_ln(x[:, r0:r1], [dict(st, out=b[:, r0 // stride:r1 // stride]) for st, b in zip(sets, bufs)],
                          mask_out=None if mask_out is None else mask_out[:, r0 // stride:r1 // stride], stride=stride,
                          x_up=None if x_up is None else x_up[:, r0 // 2:r1 // 2], pre_ln=pre_ln, segs=rel)
    return [ops.Pair(b, Cout) if ops._fmt(st.get("pair")) else b for st, b in zip(sets, bufs)]


def _attention_rows(q, k, v, kv_mask, q_mask, n_head, qsegs, ksegs, *, half_win=None, rel_pe=None, pair=False, plain=False):
    """global (half_win None) or banded attention bucket by bucket; plain: MaskedMHA's call form (f32 rows in and out)."""
    ops = _ops()
    Cc = q.shape[-1]
    out = torch.empty(1, sum(n * T for _, n, T in qsegs), Cc, device=_raw(q).device, dtype=torch.float32)
    r = None
    # buckets of one frame count on both sides (the predictor's query self-attention: every segment is n x Q rows) are one launch
    if len(qsegs) > 1 and len(_merged(qsegs)) == 1 and len(_merged(ksegs)) == 1:
        qsegs, ksegs = _merged(qsegs), _merged(ksegs)
    if half_win is not None and 1 < len(qsegs) <= _MAX_SEGS:      # banded attention: the buckets as the kernel's row groups
        assert list(qsegs) == list(ksegs)
        r = ops.local_attention(q, k, v, kv_mask, n_head, half_win, pair=pair, rel_pe=rel_pe, out=out, segs=list(qsegs))
        return ops.Pair(out, Cc, r.fmt) if isinstance(r, ops.Pair) else out
    # global attention walks the buckets; a bucket's launch is n x heads x query blocks workgroups of one per CU -- 4.02 rounds
    # of the chip for 257 pairs x 4 heads take five --, so the buckets' launches alternate between ATTN_LANES streams and fill
    # each other's last rounds
    dev = _raw(q).device
    lanes = _lanes(dev) if half_win is None and len(qsegs) > 2 else None
    if lanes:
        main = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)
        for lane in lanes:
            lane.wait_event(ready)
    for i, ((qo, n, Tq), (ko, nk, Tk)) in enumerate(zip(qsegs, ksegs)):
        assert n == nk
        o = _part(out, qo, n, Tq)
        if lanes and i % (len(lanes) + 1):
            with torch.cuda.stream(lanes[i % (len(lanes) + 1) - 1]):
                r = _attention_one(q, k, v, kv_mask, q_mask, n_head, (qo, n, Tq), (ko, nk, Tk), o, pair, plain)
            continue
        if half_win is not None:
            assert Tq == Tk
            r = ops.local_attention(_part(q, qo, n, Tq), _part(k, ko, n, Tk), _part(v, ko, n, Tk), _mpart(kv_mask, ko, n, Tk),
                                    n_head, half_win, pair=pair, rel_pe=rel_pe, out=o)
        elif plain:
            r = ops.attention(_part(q, qo, n, Tq), _part(k, ko, n, Tk), _part(v, ko, n, Tk), _mpart(kv_mask, ko, n, Tk), n_head, out=o)
        else:
            r = ops.attention(_part(q, qo, n, Tq), _part(k, ko, n, Tk), _part(v, ko, n, Tk), _mpart(kv_mask, ko, n, Tk), n_head,
                              pair=pair, q_mask=_mpart(q_mask, qo, n, Tq), out=o)
    if lanes:
        for lane in lanes:
            main.wait_stream(lane)
    return ops.Pair(out, Cc, r.fmt) if isinstance(r, ops.Pair) else out


def _attention_one(q, k, v, kv_mask, q_mask, n_head, qseg, kseg, o, pair, plain):
    ops = _ops()
    (qo, n, Tq), (ko, _, Tk) = qseg, kseg
    if plain:
        return ops.attention(_part(q, qo, n, Tq), _part(k, ko, n, Tk), _part(v, ko, n, Tk), _mpart(kv_mask, ko, n, Tk), n_head, out=o)
    return ops.attention(_part(q, qo, n, Tq), _part(k, ko, n, Tk), _part(v, ko, n, Tk), _mpart(kv_mask, ko, n, Tk), n_head,
                         pair=pair, q_mask=_mpart(q_mask, qo, n, Tq), out=o)


ATTN_LANES = int(__import__("os").environ.get("VRDONE_ROWS_ATTN_STREAMS", "2"))      # streams the buckets' global attention alternates between
_side_streams = {}


def _lanes(dev):
    """the side streams of `dev` (ATTN_LANES - 1 of them), or None: one stream, or a capture in progress"""
    if ATTN_LANES <= 1 or torch.cuda.is_current_stream_capturing():
        return None
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), ATTN_LANES)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(ATTN_LANES - 1)]
    return _side_streams[key]


def _attn_rows(mod, q_in, k_in, v_in, q_mask, kv_mask, qsegs, ksegs, *, stride=1, pre_ln=None, pre_ln_on="", **epilogue):
    """An attention module's cl / cl_qkv on flat rows (blocks.py LocalMaskedMHCA.cl / MaskedMHCA.cl with stride, q_mask =
    kv_mask = the strided mask; local_transformer.py MaskedMHCA_QKV.cl_qkv; blocks.py MaskedMHA.cl_qkv)."""
    ops = _ops()
    if stride > 1:
        qsegs_o = ksegs_o = [(off // stride, n, T // stride) for off, n, T in qsegs]
    else:
        qsegs_o, ksegs_o = qsegs, ksegs
    if isinstance(mod, _ConvAttention):
        outs = {}
        for x, m, use_ln, _, names in mod._groups(q_in, k_in, v_in, q_mask, kv_mask, pre_ln, pre_ln_on):
            res = _dwconv_rows(x, [mod._branch_set(nm) for nm in names], m, qsegs if x is q_in else ksegs, stride=stride,
                               pre_ln=pre_ln if use_ln else None)
            outs.update(zip(names, res))
        q, k, v = outs["query"], outs["key"], outs["value"]
        half_win = getattr(mod, "_half_win", None)
        if half_win is None and hasattr(mod, "window_overlap"):
            half_win = mod.window_overlap                                              # blocks.py LocalMaskedMHCA
        # (the flash kernel's pair-row q / k / v: one choice for all buckets)
        qkv_pair = half_win is None and ops.flash_pair_ok(mod.n_head, mod.n_embd, min(T for _, _, T in qsegs_o))
        q, k, v = mod._project(q, k, v, out_pair=qkv_pair, q_mask=q_mask, kv_mask=kv_mask)
        att = _attention_rows(q, k, v, kv_mask, q_mask, mod.n_head, qsegs_o, ksegs_o, half_win=half_win,
                              rel_pe=getattr(mod, "rel_pe", None), pair=ops.pair_mode())
    else:                                                                              # blocks.py MaskedMHA.cl_qkv
        q = ops.conv_gemm(q_in, mod.query.weight, mod.query.bias)
        k = ops.conv_gemm(k_in, mod.key.weight, mod.key.bias)
        v = ops.conv_gemm(v_in, mod.value.weight, mod.value.bias)
        att = _attention_rows(q, k, v, kv_mask, None, mod.n_head, qsegs_o, ksegs_o, plain=True)
    return ops.conv_gemm(att, mod.proj.weight, mod.proj.bias, row_mask=q_mask, **epilogue)


def _block(blk, x, mask, segs, out=None):
    """blocks.py TransformerBlock.cl (eval: no stochastic depth) -> (y, mask_out, segs_out)"""
    ops = _ops()
    s = blk.attn.n_kv_stride
    if s > 1:
        assert s == 2
        rows_o = sum(n * (T // 2) for _, n, T in segs)
        skip = torch.empty(1, rows_o, x.shape[-1], device=x.device, dtype=torch.float32)
        m_out = torch.empty(1, rows_o, device=x.device, dtype=torch.bool)
        for off, n, T in segs:
            ops.maxpool_mask(_part(x, off, n, T), _mpart(mask, off, n, T), out=(_part(skip, off // 2, n, T // 2), _mpart(m_out, off // 2, n, T // 2)))
        segs_o = [(off // 2, n, T // 2) for off, n, T in segs]
    else:
        skip, m_out, segs_o = x, mask, segs
    y = _attn_rows(blk.attn, x, x, x, m_out, m_out, segs, segs, stride=s, pre_ln=(blk.ln1.weight, blk.ln1.bias), pre_ln_on="qkv",
                   scale=blk._scale(blk.drop_path_attn), res=skip, res_masked=True)
    h = blk.ln2.cl(y, pair=ops.pair_mode())
    h = ops.conv_gemm(h, blk.mlp[0].weight, blk.mlp[0].bias, act=ops.ACT_GELU, out_pair=ops.pair_mode(), skip_rows=m_out)
    y = ops.conv_gemm(h, blk.mlp[3].weight, blk.mlp[3].bias, row_mask=m_out, scale=blk._scale(blk.drop_path_mlp), res=y, out=out)
    return y, m_out, segs_o


def _decoder_layer(layer, tgt, memory, tgt_mask, memory_mask, qsegs, ksegs, *, query_pos=None, stream_add=None, out=None):
    """local_transformer.py MaskedConvTransformerDecoderLayer.cl (eval) on flat rows"""
    ops = _ops()
    from .local_transformer import MaskedMHCA_QKV
    fuse1 = query_pos is None and isinstance(layer.self_attn, MaskedMHCA_QKV)
    fuse2 = query_pos is None and isinstance(layer.multihead_attn, MaskedMHCA_QKV)
    last = not layer.with_ffn
    kw1 = dict(scale=layer._scale(layer.drop_path_attn1), res=tgt, res_masked=True)
    if fuse1:
        tgt = _attn_rows(layer.self_attn, tgt, tgt, tgt, tgt_mask, tgt_mask, qsegs, qsegs, pre_ln=(layer.ln1.weight, layer.ln1.bias),
                         pre_ln_on="qk", **kw1)
    else:
        t2 = layer.ln1.cl(tgt, post_add=query_pos)
        tgt = _attn_rows(layer.self_attn, t2, t2, tgt, tgt_mask, tgt_mask, qsegs, qsegs, **kw1)
    kw = dict(scale=layer._scale(layer.drop_path_attn2), res=tgt, res_masked=True, res2=stream_add if last else None,
              out=out if last else None)
    if fuse2:
        tgt = _attn_rows(layer.multihead_attn, tgt, memory, memory, tgt_mask, memory_mask, qsegs, ksegs,
                         pre_ln=(layer.ln2.weight, layer.ln2.bias), pre_ln_on="q", **kw)
    else:
        t2 = layer.ln2.cl(tgt, post_add=query_pos)
        tgt = _attn_rows(layer.multihead_attn, t2, memory, memory, tgt_mask, memory_mask, qsegs, ksegs, **kw)
    if layer.with_ffn:
        assert stream_add is None
        t2 = layer.ln3.cl(tgt)
        h = ops.conv_gemm(t2, layer.mlp[0].weight, layer.mlp[0].bias, act=ops.ACT_GELU)
        tgt = ops.conv_gemm(h, layer.mlp[3].weight, layer.mlp[3].bias, row_mask=tgt_mask, scale=layer._scale(layer.drop_path_mlp),
                            res=tgt, out=out)
    return tgt


def _rows_of(x, sl):
    ops = _ops()
    return ops.Pair(x.t[:, sl], x.width, x.fmt) if isinstance(x, ops.Pair) else x[:, sl]


def _conv3_rows(h, conv, row_mask, lay, halves, tails, out=None, out_pair=False, norm=None):
    """A dense k = 3 conv * mask over the row space: one flat launch per half over the rows of the buckets whose sequences end
    in two padded frames (their last rows zeroed in the input first), bucket by bucket for the others.
    norm: the LayerNorm (+ ReLU) behind a few-channel conv, fused into the same launches (ops.conv_ln); with few input channels
    and no norm the conv alone runs as that row kernel."""
    ops = _ops()
    assert conv.kernel_size[0] == 3
    N = conv.weight.shape[0]
    if out is None:
        out = torch.empty(1, halves * lay.rows, N, device=_raw(h).device, dtype=torch.float32)
    if tails is not None:
        _zero_rows(h, tails)
    small = ops.conv_ln_ok(h, conv.weight, conv.bias, *((norm.weight, norm.bias) if norm is not None else ()))
    assert small or norm is None

    def one(x, m, o):
        if small:
            ops.conv_ln(x, conv.weight, conv.bias, row_mask=m, out=o, pair=out_pair, relu=norm is not None,
                        gamma=None if norm is None else norm.weight.reshape(-1), beta=None if norm is None else norm.bias.reshape(-1))
        else:
            ops.conv_gemm(x, conv.weight, conv.bias, row_mask=m, out=o, out_pair=out_pair)
    for hf in range(halves):
        base = hf * lay.rows
        if lay.rows_flat:
            sl = slice(base, base + lay.rows_flat)
            one(_rows_of(h, sl), row_mask[:, sl], out[:, sl])
        for (off, n, T), flat in zip(lay.segs, lay.flat):
            if not flat:
                one(_part(h, base + off, n, T), _mpart(row_mask, base + off, n, T), _part(out, base + off, n, T))
    return ops.Pair(out, N) if out_pair else out


def _embed(h, convs, norms, mask2, out, lay, tails2):
    """backbones.py _embed on the stacked rows: k = 3 conv * mask -> LN -> ReLU"""
    ops = _ops()
    last = len(convs) - 1
    if last == 0 and ops.conv_ln_ok(h, convs[0].conv.weight, convs[0].conv.bias, norms[0].weight, norms[0].bias):
        # few input channels (the box features): conv, mask, LayerNorm and ReLU as one row kernel, straight into the consumer's slab
        return _conv3_rows(h, convs[0].conv, mask2, lay, 2, tails2, out=out, out_pair=bool(ops.pair_mode()), norm=norms[0])
    for i, (conv, norm) in enumerate(zip(convs, norms)):
        h = _conv3_rows(h, conv.conv, mask2, lay, 2, tails2)
        h = norm.cl(h, relu=True, out=out if i == last else None, pair=ops.pair_mode())
    return h


def unpack_rows(bb, x, plan, lay):
    """backbones.py _unpack for the buckets of `plan` = [(T_i, pair indices (int32, device), n_i, ...)] over the caller's batch
    x (B, C_in, T) -> vis, clip, so_box, ent in the row space (vis / clip / ent stacked [subject | object])."""
    ops = _ops()
    R = lay.rows
    V, Cc, S, E = bb.n_visual, bb.n_clip, bb.n_bbox_so, bb.n_bbox_entity
    pair = ops.pair_mode()
    new = lambda rows, width: torch.empty(1, rows, width, device=x.device, dtype=torch.float32)      # noqa: E731

    def stacked(c0, width, as_pair):
        h = new(2 * R, width)
        for (off, n, T), bucket in zip(lay.segs, plan):
            t2, idx = bucket[0], bucket[1]
            ops.bct_to_btc(x, c0, width, _part(h, off, n, T), pair=as_pair, frames=t2, index=idx)
            ops.bct_to_btc(x, c0 + width, width, _part(h, R + off, n, T), pair=as_pair, frames=t2, index=idx)
        return ops.Pair(h, width) if as_pair else h

    o0 = 2 * V + 2 * Cc
    so_box = new(R, S)
    for (off, n, T), bucket in zip(lay.segs, plan):
        ops.bct_to_btc(x, o0, S, _part(so_box, off, n, T), frames=bucket[0], index=bucket[1])
    return stacked(0, V, pair), (stacked(2 * V, Cc, pair) if Cc else None), so_box, stacked(o0 + S, E, False)


def entity_rows(bb, vis, clip, ent, mask2, lay):
    """backbones.py entity_stage on the stacked rows (1, 2R, .) -> so (1, 2R, D)"""
    ops = _ops()
    assert not bb.use_abs_pe, "absolute position rows are laid out per padded length: such models run bucket by bucket"
    dev = mask2.device
    R = lay.rows
    D = bb.s_fuse_norm.num_channels
    pair = ops.pair_mode()
    new = lambda rows, width: torch.empty(1, rows, width, device=dev, dtype=torch.float32)      # noqa: E731
    cat = (lambda t: ops.Pair(t, D)) if pair else (lambda t: t)                                   # noqa: E731
    tails2 = lay.tail_rows(2, dev)
    fuse_in = new(2 * R, 2 * D)
    if bb.n_clip:
        vc = new(2 * R, 2 * D)
        _embed(vis, bb.visual_embd, bb.visual_embd_norm, mask2, vc[..., :D], lay, tails2)
        _embed(clip, bb.clip_embd, bb.clip_embd_norm, mask2, vc[..., D:], lay, tails2)
        bb.visual_clip_fuse.cl(cat(vc), row_mask=mask2, out=fuse_in[..., :D], out_pair=pair)
    else:
        _embed(vis, bb.visual_embd, bb.visual_embd_norm, mask2, fuse_in[..., :D], lay, tails2)
    _embed(ent, [bb.bbox_entity_embd], [bb.bbox_entity_norm], mask2, fuse_in[..., D:], lay, tails2)
    so = bb.visual_bbox_fuse.cl(cat(fuse_in), row_mask=mask2)
    so, _, _ = _block(bb.stem[0], so, mask2, lay.twice())
    return so


def pair_rows(bb, so, so_box, mask, lay):
    """backbones.py pair_stage: so (1, 2R, D) entity-stage rows [subject | object], so_box (1, R, S), mask (1, R)
    -> feats, masks, segs per pyramid level"""
    ops = _ops()
    dev = mask.device
    R = lay.rows
    D = bb.s_fuse_norm.num_channels
    pair = ops.pair_mode()
    segs, segs2 = lay.segs, lay.twice()
    new = lambda rows, width: torch.empty(1, rows, width, device=dev, dtype=torch.float32)      # noqa: E731
    cat = (lambda t: ops.Pair(t, D)) if pair else (lambda t: t)                                   # noqa: E731
    mask2 = torch.cat([mask, mask], dim=1)
    for i, (s_attn, o_attn) in enumerate(zip(bb.s_attn, bb.o_attn)):
        if i:
            so, _, _ = _block(bb.stem[i], so, mask2, segs2)
        s, o = so[:, :R], so[:, R:]
        nxt = new(2 * R, D)
        _decoder_layer(s_attn, s, o, mask, mask, segs, segs, stream_add=s, out=nxt[:, :R])        # s + s_attn(s, o)
        _decoder_layer(o_attn, o, s, mask, mask, segs, segs, stream_add=o, out=nxt[:, R:])        # uses the pre-update s
        so = nxt
    so_in = new(R, 2 * D)
    bb.s_fuse_norm.cl(so[:, :R], out=so_in[..., :D], pair=pair)
    bb.o_fuse_norm.cl(so[:, R:], out=so_in[..., D:], pair=pair)
    pair_box = new(R, 2 * D)
    bb.so_fuse.cl(cat(so_in), row_mask=mask, out=pair_box[..., :D], out_pair=pair)
    _conv3_rows(so_box, bb.bbox_so_embd.conv, mask, lay, 1, lay.tail_rows(1, dev), out=pair_box[..., D:], out_pair=pair)
    e = bb.so_visual_bbox_fuse.cl(cat(pair_box), row_mask=mask)

    feats, masks, lays = [e], [mask], [segs]
    for blk in bb.branch:
        e, mask, segs = _block(blk, e, mask, segs)
        feats.append(e)
        masks.append(mask)
        lays.append(segs)
    return feats, masks, lays


def backbone_rows(bb, x, plan, lay, mask):
    """backbones.py cl for the buckets of `plan` over the caller's batch x (B, C_in, T); mask: flat (1, R) validity of the rows"""
    vis, clip, so_box, ent = unpack_rows(bb, x, plan, lay)
    so = entity_rows(bb, vis, clip, ent, torch.cat([mask, mask], dim=1), lay)
    return pair_rows(bb, so, so_box, mask, lay)


def neck_rows(neck, feats, masks, lays):
    """fpns.py FPN1D_Fuse.cl"""
    ops = _ops()
    y = None
    for l in range(len(neck.lateral_convs) - 1, -1, -1):
        x = neck.input_norms[l].cl(feats[l], pair=ops.pair_mode() and neck.lateral_convs[l] is not None)
        fpn = dict(weight=neck.fpn_convs[l].conv.weight, gamma=neck.fpn_norms[l].weight, beta=neck.fpn_norms[l].bias)
        if neck.lateral_convs[l] is None:
            y, = _dwconv_rows(x, [fpn], masks[l], lays[l])
        else:
            c = ops.conv_gemm(x, neck.lateral_convs[l].conv.weight, None, row_mask=masks[l])
            c = neck.lateral_norms[l].cl(c)
            y, = _dwconv_rows(c, [fpn], masks[l], lays[l], x_up=y)
    mf = neck.mask_features.conv
    out, = _dwconv_rows(y, [dict(weight=mf.weight, bias=mf.bias)], masks[0], lays[0])
    return out


def predictor_rows(pred, x, mask_features, mask, output_mask, ksegs, segs0, with_aux, fill=-10.0):
    """predictor.py MaskedTransformerPredictor.cl: the queries of all pairs are one (B, Q, C) tensor (pairs in bucket
    order); the decoder's cross attention and the mask head walk the buckets.  -> [(logits (B, Q, K+1), [seg_i (n_i, Q, T_i)])]
    for the last decoder layer, preceded by the auxiliary layers' when asked for."""
    ops = _ops()
    if with_aux is None:
        with_aux = pred.aux_loss
    src = pred.input_norm.cl(x, pair=ops.pair_mode() and pred.input_proj is not None)
    if pred.input_proj is not None:
        src = ops.conv_gemm(src, pred.input_proj.weight, pred.input_proj.bias, row_mask=mask)
    qe = pred.query_embed.weight
    Q, Cq = qe.shape
    B = sum(n for _, n, _ in ksegs)
    qsegs, p = [], 0
    for _, n, _ in ksegs:
        qsegs.append((p * Q, n, Q))
        p += n
    dec = pred.transformer.decoder
    tgt = torch.zeros(1, B * Q, Cq, device=_raw(src).device, dtype=torch.float32)
    hs = []
    all_layers = with_aux and pred.aux_loss
    for i, layer in enumerate(dec.layers):
        tgt = _decoder_layer(layer, tgt, src, None, mask, qsegs, ksegs, query_pos=qe)
        if all_layers or i == dec.num_layers - 1:
            hs.append(dec.norm.cl(tgt))
    heads = []
    for h in hs:
        h3 = h.view(B, Q, Cq)
        logits = ops.conv_gemm(h3, pred.class_embed.weight, pred.class_embed.bias)
        emb = pred.mask_embed.cl(h3)
        segs_out, p = [], 0
        for off, n, T in segs0:
            segs_out.append(ops.mask_head(emb[p:p + n], _part(mask_features, off, n, T), _mpart(output_mask, off, n, T), fill))
            p += n
        heads.append((logits, segs_out))
    return heads


def heads_rows(model, feats, masks, lays, with_aux):
    """neck + predictor over the row space -> [(logits (B, Q, K+1), [mask logits (n_i, Q, T_i) per bucket])] per decoder layer asked for"""
    mask_features = neck_rows(model.neck, feats, masks, lays)
    return predictor_rows(model.predictor, feats[-1], mask_features, masks[-1], masks[0], lays[-1], lays[0], with_aux)


def filler_buckets(rows, t_max):
    """[(sequences, frames)] of all-padding sequences that round a row space of `rows` rows up to a multiple of 256: the
    256 x 256 GEMM kernel takes row counts that are multiples of 64 (its epilogue has no row predicates), and with 256 that
    holds for the first three pyramid levels -- the launches large enough for that kernel.  Sequences of 32 frames, plus one of
    32 + (8, 16 or 24) when the buckets' lengths are not all multiples of 32 (reference padded lengths are multiples of the
    window stride, e.g. 48); none when such a sequence would be longer than t_max (the frames the caller's batch has)."""
    need = (-rows) % 256
    if need == 0 or rows % 8 or rows < 65536:          # (below ~65 k rows no launch reaches the 256 x 256 kernel)
        return []
    odd = need % 32
    first = 32 + odd if odd else 0
    if first > t_max or t_max < 32:
        return []
    if need < first:
        need += 256
    out = [(1, first)] if first else []
    if need - first:
        out.insert(0, ((need - first) // 32, 32))
    return out


def with_filler(buckets, make, t_max):
    """`buckets` [(frames, ..., sequences, flat)] plus, behind the last flat one, the filler buckets make(n, T) the row count asks
    for -> (buckets, indices of the fillers)"""
    fill = filler_buckets(sum(b[2] * b[0] for b in buckets), t_max)
    at = sum(1 for b in buckets if b[-1])
    return list(buckets[:at]) + [make(n, T) for n, T in fill] + list(buckets[at:]), set(range(at, at + len(fill)))


def mask_vrd_rows(model, x, masks2d, plan, with_aux, out=None):
    """MaskVRD._mask_vrd for the buckets of `plan` = [(T_i, pair indices, n_i, flat_i)] in one row space; results written into (or
    returned as) the batch-shaped output dict: pred_logits (B, Q, K+1), pred_masks (B, Q, T) at the batch's own padded length,
    -10 behind a bucket's."""
    B, T = masks2d.shape
    dev = x.device
    # (a filler bucket recomputes the first frames of some pair under an all-false mask: finite numbers nobody reads)
    plan, filler = with_filler(plan, lambda n, t: (t, plan[0][1][:1].repeat(n), n, True), T)
    lay = Layout([(n, t2, flat) for t2, _, n, flat in plan])
    idx64 = [b[1].long() for b in plan]
    mask = torch.cat([masks2d[i64, :b[0]].reshape(-1) if j not in filler else torch.zeros(b[2] * b[0], dtype=torch.bool, device=dev)
                      for j, (b, i64) in enumerate(zip(plan, idx64))]).view(1, lay.rows)
    heads = heads_rows(model, *backbone_rows(model.backbone, x, plan, lay, mask), with_aux)
    fill = -10.0                                        # the predictor's value on padded frames (predictor.py:39)
    if out is None:
        Q, K1 = heads[-1][0].shape[1:]
        new = lambda: {"pred_logits": torch.empty(B, Q, K1, device=dev), "pred_masks": torch.full((B, Q, T), fill, device=dev)}  # noqa: E731
        out = new()
        if len(heads) > 1:
            out["aux_outputs"] = [new() for _ in heads[:-1]]
    for dst, (logits, segs) in zip(out.get("aux_outputs", []) + [out], heads):
        p = 0
        for j, (b, i64, seg) in enumerate(zip(plan, idx64, segs)):
            if j not in filler:
                dst["pred_logits"][i64] = logits[p:p + b[2]]
                dst["pred_masks"][i64, :, :b[0]] = seg
            p += b[2]
    return out
