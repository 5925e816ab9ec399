"""Backbone of the relation encoder: embedding convs, entity-box fusion, stem (local-window
transformer on subject and object) interleaved with subject<->object mutual attention, s/o
fusion and the stride-2 pyramid branch.  Same constructor and parameter tree as the reference's
models/backbones.py; the forward chains HIP kernels on channels-last buffers.

Subject and object share the embedding / stem weights (reference backbones.py:172-177,
213-214), so both streams run as ONE batch of 2B sequences; concatenations are never
materialised -- producers write into column slabs of the consumer's input buffer.
"""
import torch
from torch import nn

from .blocks import ConvMLP, LayerNorm, MaskedConv1D, TransformerBlock, _from_cl, _mask2d, _ops, get_sinusoid_encoding
from .local_transformer import MaskedConvTransformerDecoderLayer


class MaskConvTransformerBackbone(nn.Module):
    def __init__(self, n_visual, n_bbox_entity, n_bbox_so, n_embd, n_head, n_embd_ks, fuse_ks, n_fuse_head,
                 fuse_path_drop, fuse_qx_stride, fuse_kv_stride, max_len, arch=(2, 2, 3), mha_win_size=[-1] * 4,
                 scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, use_abs_pe=False,
                 use_rel_pe=False, use_local=True):
        super().__init__()
        assert len(arch) == 3 and len(mha_win_size) == 1 + arch[-1]
        assert with_ln, "built for embd_with_ln=True"
        assert scale_factor == 2 and n_embd_ks == 3
        self.n_visual, self.n_bbox_entity, self.n_bbox_so = n_visual, n_bbox_entity, n_bbox_so
        self.n_clip = 0
        self.arch, self.mha_win_size, self.max_len = arch, mha_win_size, max_len
        self.relu = nn.ReLU(inplace=True)
        self.scale_factor, self.use_abs_pe, self.use_rel_pe = scale_factor, use_abs_pe, use_rel_pe

        if use_abs_pe:      # reference backbones.py:70-72 (not part of the checkpoint)
            self.register_buffer("pos_embd", get_sinusoid_encoding(max_len, n_embd) / (n_embd ** 0.5), persistent=False)
        self.visual_embd, self.visual_embd_norm = self._embedding(n_visual, n_embd, n_embd_ks, arch[0])
        self.bbox_entity_embd = MaskedConv1D(n_bbox_entity, n_embd, n_embd_ks, stride=1, padding=n_embd_ks // 2)
        self.bbox_entity_norm = LayerNorm(n_embd)
        self.visual_bbox_fuse = ConvMLP(n_embd * 2, n_embd, n_embd, kernel_size=fuse_ks, num_layers=2)

        self.stem = nn.ModuleList()
        self.s_attn = nn.ModuleList()
        self.o_attn = nn.ModuleList()
        for _ in range(arch[1]):
            self.stem.append(TransformerBlock(n_embd, n_head, n_ds_strides=(1, 1), attn_pdrop=attn_pdrop,
                                              proj_pdrop=proj_pdrop, path_pdrop=path_pdrop,
                                              mha_win_size=mha_win_size[0], use_rel_pe=use_rel_pe))
            for mutual in (self.s_attn, self.o_attn):
                mutual.append(MaskedConvTransformerDecoderLayer(
                    n_embd=n_embd, n_head=n_fuse_head, path_pdrop=fuse_path_drop, n_qx_stride=fuse_qx_stride,
                    n_kv_stride=fuse_kv_stride, with_ffn=False, use_local=use_local,
                    win_size=mha_win_size[0] if use_local else None))
        self.s_fuse_norm = LayerNorm(n_embd)
        self.o_fuse_norm = LayerNorm(n_embd)
        self.so_fuse = ConvMLP(n_embd * 2, n_embd, n_embd, kernel_size=fuse_ks, num_layers=2)
        self.bbox_so_embd = MaskedConv1D(n_bbox_so, n_embd, n_embd_ks, stride=1, padding=n_embd_ks // 2)
        self.so_visual_bbox_fuse = ConvMLP(n_embd * 2, n_embd, n_embd, kernel_size=fuse_ks, num_layers=2)

        self.branch = nn.ModuleList(
            TransformerBlock(n_embd, n_head, n_ds_strides=(scale_factor, scale_factor), attn_pdrop=attn_pdrop,
                             proj_pdrop=proj_pdrop, path_pdrop=path_pdrop, mha_win_size=mha_win_size[1 + i],
                             use_rel_pe=use_rel_pe)
            for i in range(arch[2]))
        self._zero_biases()

    @staticmethod
    def _embedding(n_in, n_embd, ks, depth):
        convs, norms = nn.ModuleList(), nn.ModuleList()
        for i in range(depth):
            convs.append(MaskedConv1D(n_in if i == 0 else n_embd, n_embd, ks, stride=1, padding=ks // 2, bias=False))
            norms.append(LayerNorm(n_embd))
        return convs, norms

    def _zero_biases(self):
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv1d)) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def in_channels(self):
        return 2 * self.n_visual + 2 * self.n_clip + self.n_bbox_so + 2 * self.n_bbox_entity

    # -------------------------------------------------------------------------------------
    def _unpack(self, x, frames=None, index=None):
        """Boundary layout (B, C_in, T) -> the channels-last operand buffers of cl_parts: the subject and
        object slabs of every shared-weight stage are stacked on the batch axis (2B sequences); the wide
        visual / clip slabs feed GEMMs only, so they are pair rows in the split-precision modes.
        frames / index: only the first `frames` frames of the pairs x[index] (ops.bct_to_btc)."""
        ops = _ops()
        B, _, T = x.shape
        if index is not None:
            B = index.numel()
        if frames is not None:
            T = frames
        V, Cc, S, E = self.n_visual, self.n_clip, self.n_bbox_so, self.n_bbox_entity
        pair = ops.pair_mode()

        def stacked(c0, width, as_pair):
            h = torch.empty(2, B, T, width, device=x.device, dtype=torch.float32)
            ops.bct_to_btc(x, c0, width, h[0], pair=as_pair, frames=frames, index=index)
            ops.bct_to_btc(x, c0 + width, width, h[1], pair=as_pair, frames=frames, index=index)
            h = h.view(2 * B, T, width)
            return ops.Pair(h, width) if as_pair else h

        o0 = 2 * V + 2 * Cc
        so_box = torch.empty(B, T, S, device=x.device, dtype=torch.float32)
        ops.bct_to_btc(x, o0, S, so_box, frames=frames, index=index)
        return (stacked(0, V, pair), stacked(2 * V, Cc, pair) if Cc else None, so_box, stacked(o0 + S, E, False))

    @staticmethod
    def _embed(h, convs, norms, mask2, out, post_add=None):
        """k=3 conv * mask -> LN -> ReLU stack on 2B stacked sequences; the last LN writes into `out`, a column
        slab of the consumer GEMM's input buffer (pair rows in bf16x3 mode)."""
        ops = _ops()
        last = len(convs) - 1
        if last == 0 and post_add is None and ops.conv_ln_ok(h, convs[0].conv.weight, convs[0].conv.bias, norms[0].weight, norms[0].bias):
            # few input channels (the box features): conv, mask, LayerNorm and ReLU as one row kernel
            c, nm = convs[0].conv, norms[0]
            return ops.conv_ln(h, c.weight, c.bias, row_mask=mask2, gamma=nm.weight.reshape(-1), beta=nm.bias.reshape(-1), relu=True,
                               out=out, pair=ops.pair_mode())
        for i, (conv, norm) in enumerate(zip(convs, norms)):
            h = ops.conv_gemm(h, conv.conv.weight, conv.conv.bias, row_mask=mask2)
            h = norm.cl(h, relu=True, out=out if i == last else None, pair=ops.pair_mode(), post_add=post_add if i == last else None)
        return h

    def cl(self, x, mask):
        """x: (B, C_in, T) contiguous fp32 (the boundary layout), mask: (B, T) bool.
        Returns channels-last feats [(B, T/2^l, D)] and masks [(B, T/2^l)]."""
        assert x.shape[1] == self.in_channels(), f"expected {self.in_channels()} input channels, got {x.shape[1]}"
        return self.cl_parts(*self._unpack(x.contiguous()), mask)

    def cl_parts(self, vis, clip, so_box, ent, mask):
        """vis (2B, T, V), clip (2B, T, Cc) or None, so_box (B, T, S), ent (2B, T, E): channels-last operand
        buffers (from _unpack or ops.pack_pairs); mask (B, T) bool."""
        return self.pair_stage(self.entity_stage(vis, clip, ent, torch.cat([mask, mask], dim=0)), so_box, mask)

    def entity_reach(self):
        """How many frames either side of a frame the entity stage reads (embedding convs, then the first stem block's
        depthwise conv and attention window), or None when that stage is not local (global attention)."""
        win = self.mha_win_size[0]
        if self.use_abs_pe:          # the position rows count frames of the PAIR's window
            return None
        if win <= 1 or any(c.kernel_size[0] != 1 for c in self.visual_bbox_fuse.layers):
            return None
        return self.arch[0] + 1 + win // 2        # k = 3 embedding convs, k = 3 depthwise conv, half a window

    def _position_rows(self, mask2):
        """Absolute position encoding as one row per (sequence, frame): the table's first T frames (linearly re-interpolated
        to T frames once T reaches max_len, eval only) on valid frames, zeros on padded ones (reference backbones.py:180-196)."""
        n, T = mask2.shape
        pe = self.pos_embd
        if self.training:
            assert T <= self.max_len, "Reached max length."
        elif T >= self.max_len:
            pe = torch.nn.functional.interpolate(pe, T, mode='linear', align_corners=False)
        rows = pe[0, :, :T].t()                                              # (T, D)
        return (rows[None] * mask2[..., None].to(rows.dtype)).reshape(n * T, -1).contiguous()

    def entity_stage(self, vis, clip, ent, mask2):
        """Everything that sees ONE entity's frames only -- embeddings, visual/box fusion and the first stem block, all
        with weights shared between subject and object (reference backbones.py:172-214): n sequences in, (n, T, D) out.
        Every op in it is local in time (see entity_reach), which is what lets forward_test run it once per tracklet
        instead of once per pair."""
        ops = _ops()
        n, T = mask2.shape
        Cc = self.n_clip
        D = self.s_fuse_norm.num_channels
        new = lambda *shape: torch.empty(*shape, device=mask2.device, dtype=torch.float32)   # noqa: E731

        # concatenation buffers hold two D-wide slabs; in bf16x3 mode both slabs are pair rows (width D)
        pair = ops.pair_mode()
        cat = (lambda t: ops.Pair(t, D)) if pair else (lambda t: t)          # noqa: E731

        # [visual (+clip) | entity box] -> visual_bbox_fuse
        # (ops.join: the slab-filled buffer, or -- under autograd, where ops return fresh tensors -- the concatenation)
        fuse_in = new(n, T, 2 * D)
        pe = self._position_rows(mask2) if self.use_abs_pe else None        # (n * T, D): pe[t] on valid frames, 0 on padded ones
        if Cc:
            vc = new(n, T, 2 * D)
            a = self._embed(vis, self.visual_embd, self.visual_embd_norm, mask2, vc[..., :D])
            b = self._embed(clip, self.clip_embd, self.clip_embd_norm, mask2, vc[..., D:])
            # (CLIP variant: the position rows are added behind the visual / CLIP fusion, backbones.py:362-384)
            a = self.visual_clip_fuse.cl(cat(ops.join(vc, (a, b))), row_mask=mask2, out=fuse_in[..., :D], out_pair=pair, res=pe)
        else:
            a = self._embed(vis, self.visual_embd, self.visual_embd_norm, mask2, fuse_in[..., :D], post_add=pe)
        b = self._embed(ent, [self.bbox_entity_embd], [self.bbox_entity_norm], mask2, fuse_in[..., D:])
        so = self.visual_bbox_fuse.cl(cat(ops.join(fuse_in, (a, b))), row_mask=mask2)
        so, _ = self.stem[0].cl(so, mask2)
        return so

    def pair_stage(self, so, so_box, mask):
        """so: (2B, T, D) entity-stage output, subject rows then object rows; from the first subject<->object attention on."""
        ops = _ops()
        B, T = mask.shape
        D = self.s_fuse_norm.num_channels
        dev = mask.device
        mask2 = torch.cat([mask, mask], dim=0)
        new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)   # noqa: E731
        pair = ops.pair_mode()
        cat = (lambda t: ops.Pair(t, D)) if pair else (lambda t: t)          # noqa: E731

        for i, (s_attn, o_attn) in enumerate(zip(self.s_attn, self.o_attn)):
            if i:
                so, _ = self.stem[i].cl(so, mask2)
            s, o = so[:B], so[B:]
            nxt = new(2 * B, T, D)
            s2, _ = s_attn.cl(s, o, mask, mask, stream_add=s, out=nxt[:B])         # s + s_attn(s, o)
            o2, _ = o_attn.cl(o, s, mask, mask, stream_add=o, out=nxt[B:])         # uses the pre-update s
            so = ops.join(nxt, (s2, o2), dim=0)

        so_in = new(B, T, 2 * D)
        a = self.s_fuse_norm.cl(so[:B], out=so_in[..., :D], pair=pair)
        b = self.o_fuse_norm.cl(so[B:], out=so_in[..., D:], pair=pair)
        pair_box = new(B, T, 2 * D)
        a = self.so_fuse.cl(cat(ops.join(so_in, (a, b))), row_mask=mask, out=pair_box[..., :D], out_pair=pair)
        conv = self.bbox_so_embd.conv
        if ops.conv_ln_ok(so_box, conv.weight, conv.bias):
            b = ops.conv_ln(so_box, conv.weight, conv.bias, row_mask=mask, out=pair_box[..., D:], pair=pair)
        else:
            b = ops.conv_gemm(so_box, conv.weight, conv.bias, row_mask=mask, out=pair_box[..., D:], out_pair=pair)
        e = self.so_visual_bbox_fuse.cl(cat(ops.join(pair_box, (a, b))), row_mask=mask)

        feats, masks = [e], [mask]
        for blk in self.branch:
            e, mask = blk.cl(e, mask)
            feats.append(e)
            masks.append(mask)
        return feats, masks

    def forward(self, x, mask):
        feats, masks = self.cl(x, _mask2d(mask))
        return tuple(_from_cl(f) for f in feats), tuple(m[:, None, :] for m in masks)


class MaskConvTransformerBackboneWithCLIP(MaskConvTransformerBackbone):
    """Adds the CLIP-feature embedding and its fusion MLP (reference models/backbones.py:250-436)."""

    def __init__(self, n_visual, n_clip, n_bbox_entity, n_bbox_so, n_embd, n_head, n_embd_ks, fuse_ks, n_fuse_head,
                 fuse_path_drop, fuse_qx_stride, fuse_kv_stride, max_len, arch=(2, 2, 3), mha_win_size=[-1] * 4,
                 scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, use_abs_pe=False,
                 use_rel_pe=False, use_local=True):
        super().__init__(n_visual=n_visual, n_bbox_entity=n_bbox_entity, n_bbox_so=n_bbox_so, n_embd=n_embd,
                         n_head=n_head, n_embd_ks=n_embd_ks, fuse_ks=fuse_ks, n_fuse_head=n_fuse_head,
                         fuse_path_drop=fuse_path_drop, fuse_qx_stride=fuse_qx_stride, fuse_kv_stride=fuse_kv_stride,
                         max_len=max_len, arch=arch, mha_win_size=mha_win_size, scale_factor=scale_factor,
                         with_ln=with_ln, attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop, path_pdrop=path_pdrop,
                         use_abs_pe=use_abs_pe, use_rel_pe=use_rel_pe, use_local=use_local)
        self.n_clip = n_clip
        self.clip_embd, self.clip_embd_norm = self._embedding(n_clip, n_embd, n_embd_ks, arch[0])
        self.visual_clip_fuse = ConvMLP(n_embd * 2, n_embd, n_embd, kernel_size=fuse_ks, num_layers=2)
        self._zero_biases()
