"""q/k/v-separated attention modules and the decoder layer used for the subject-object
mutual attention (SOS) and for the predictor's query decoder.  Class names, constructor
arguments and parameter trees follow the reference's models/local_transformer.py; forwards run
HIP kernels (see models/blocks.py for the two call forms)."""
from typing import Optional

import torch
from torch import nn, Tensor

from .blocks import (make_rel_pe, AffineDropPath, LayerNorm, MaskedMHA, _ConvAttention, _from_cl, _mask2d, _ops,
                     _to_cl)
from .transformer import _get_clones


class MaskedMHA_QKV(MaskedMHA):
    """Separate q/k/v inputs, plain 1x1 projections; reference models/local_transformer.py:13-67."""

    def forward(self, q, k, v, _qx_mask, _kv_mask, _attn_mask=None):
        assert _attn_mask is None
        y, _ = self.cl_qkv(_to_cl(q), _to_cl(k), _to_cl(v), _mask2d(_qx_mask), _mask2d(_kv_mask))
        return _from_cl(y), _qx_mask


def _qkv_kernel(stride):
    # reference local_transformer.py:109,118: k = 1 when the stride argument is 0, 3 when it is 1
    return stride + 1 if (stride > 1 or stride == 0) else 3


class MaskedMHCA_QKV(_ConvAttention):
    """Conv attention with separate q/k/v inputs and full (global) masked attention;
    reference models/local_transformer.py:69-187."""
    _half_win = None

    def __init__(self, n_embd, n_head, n_qx_stride=0, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_qx_stride in (0, 1) and n_kv_stride in (0, 1) and attn_pdrop == 0.0 and proj_pdrop == 0.0
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        self._build(n_embd, n_head, _qkv_kernel(n_qx_stride), _qkv_kernel(n_kv_stride), 1)

    def cl_qkv(self, q_in, k_in, v_in, q_mask, kv_mask, pre_ln=None, pre_ln_on="", **epilogue):
        """pre_ln = (gamma, beta) with pre_ln_on naming the inputs ('q', 'k', 'v') that are passed un-normalised and
        get that LayerNorm inside the depthwise-conv kernel (the decoder layer's ln1 / ln2)."""
        ops = _ops()
        q, k, v = self._prep(q_in, k_in, v_in, q_mask, kv_mask, pre_ln=pre_ln, pre_ln_on=pre_ln_on)
        # global attention on the split-precision flash kernel consumes q/k/v as pair rows
        qkv_pair = self._half_win is None and ops.flash_pair_ok(self.n_head, self.n_embd, q.shape[1])
        q, k, v = self._project(q, k, v, out_pair=qkv_pair, q_mask=q_mask, kv_mask=kv_mask)
        if self._half_win is None:
            att = ops.attention(q, k, v, kv_mask, self.n_head, pair=ops.pair_mode(), q_mask=q_mask)
        else:
            assert q.shape[1] == k.shape[1]
            att = ops.local_attention(q, k, v, kv_mask, self.n_head, self._half_win, pair=ops.pair_mode(),
                                      rel_pe=getattr(self, "rel_pe", None))
        return ops.conv_gemm(att, self.proj.weight, self.proj.bias, row_mask=q_mask, **epilogue), q_mask

    def forward(self, q, k, v, _qx_mask, _kv_mask, _attn_mask=None):
        assert _attn_mask is None
        qc = _to_cl(q)
        kc = qc if k is q else _to_cl(k)
        vc = kc if v is k else (qc if v is q else _to_cl(v))
        qm = _mask2d(_qx_mask)
        km = qm if _kv_mask is _qx_mask else _mask2d(_kv_mask)
        y, _ = self.cl_qkv(qc, kc, vc, qm, km)
        return _from_cl(y), _qx_mask


class LocalMaskedMHCA_QKV(MaskedMHCA_QKV):
    """Same parameters, banded attention (configs/vidor_local.yaml `use_local: True`);
    reference models/local_transformer.py:289-623."""

    def __init__(self, n_embd, n_head, window_size, n_qx_stride=0, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0,
                 use_rel_pe=False):
        super().__init__(n_embd, n_head, n_qx_stride, n_kv_stride, attn_pdrop, proj_pdrop)
        assert window_size > 1 and window_size % 2 == 1
        self.window_size, self.window_overlap = window_size, window_size // 2
        self.use_rel_pe = use_rel_pe
        self._half_win = self.window_overlap
        self.rel_pe = make_rel_pe(n_embd, n_head, window_size) if use_rel_pe else None


class MaskedConvTransformerDecoderLayer(nn.Module):
    """self-attention -> cross-attention (-> FFN), pre-LN with channel-scaled residual branches;
    reference models/local_transformer.py:625-835."""

    def __init__(self, n_embd, n_head, n_hidden=None, act_layer=nn.GELU, attn_pdrop=0.0, proj_pdrop=0.0,
                 path_pdrop=0.0, n_qx_stride=0, n_kv_stride=1, with_ffn=True, use_local=False, win_size=None,
                 use_rel_pe=False):
        super().__init__()
        assert n_qx_stride >= 0 and n_kv_stride >= 0 and act_layer is nn.GELU
        self.with_ffn = with_ffn
        self.ln1 = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        if use_local:
            assert win_size is not None and n_qx_stride != 0 and n_kv_stride != 0, \
                "local decoder layers are built for the conv (stride 1) form only"
            self.self_attn = LocalMaskedMHCA_QKV(n_embd, n_head, window_size=win_size, n_qx_stride=n_qx_stride,
                                                 n_kv_stride=n_kv_stride, use_rel_pe=use_rel_pe)
            self.multihead_attn = LocalMaskedMHCA_QKV(n_embd, n_head, window_size=win_size, n_qx_stride=n_qx_stride,
                                                      n_kv_stride=n_kv_stride, use_rel_pe=use_rel_pe)
        else:
            if n_qx_stride == 0:
                self.self_attn = MaskedMHA_QKV(n_embd, n_head)
            else:       # reference passes n_qx_stride for both strides of the self-attention (:711-718)
                self.self_attn = MaskedMHCA_QKV(n_embd, n_head, n_qx_stride=n_qx_stride, n_kv_stride=n_qx_stride)
            if n_kv_stride == 0:
                assert n_qx_stride == 0
                self.multihead_attn = MaskedMHA_QKV(n_embd, n_head)
            else:
                self.multihead_attn = MaskedMHCA_QKV(n_embd, n_head, n_qx_stride=n_qx_stride, n_kv_stride=n_kv_stride)
        if path_pdrop > 0.0:
            self.drop_path_attn1 = AffineDropPath(n_embd, drop_prob=path_pdrop)
            self.drop_path_attn2 = AffineDropPath(n_embd, drop_prob=path_pdrop)
        else:
            self.drop_path_attn1 = nn.Identity()
            self.drop_path_attn2 = nn.Identity()
        if with_ffn:
            self.ln3 = LayerNorm(n_embd)
            n_hidden = n_hidden or 4 * n_embd
            self.mlp = nn.Sequential(nn.Conv1d(n_embd, n_hidden, 1), act_layer(), nn.Dropout(proj_pdrop, inplace=True),
                                     nn.Conv1d(n_hidden, n_embd, 1), nn.Dropout(proj_pdrop, inplace=True))
            self.drop_path_mlp = AffineDropPath(n_embd, drop_prob=path_pdrop) if path_pdrop > 0.0 else nn.Identity()

    @staticmethod
    def _scale(dp):
        return dp.scale if isinstance(dp, AffineDropPath) else None

    @staticmethod
    def _drop(dp, tgt):
        """Per-row stochastic-depth factors for a branch shaped like tgt (B, Tq, C), or None."""
        return dp.row_factors(tgt.shape[0], tgt.shape[1], tgt.device) if isinstance(dp, AffineDropPath) else None

    def cl(self, tgt, memory, tgt_mask, memory_mask, query_pos=None, stream_add=None, out=None):
        """tgt (B, Tq, C), memory (B, Tk, C); masks (B, T) or None (= all valid).  query_pos: (Tq, C)
        rows added to the normalised target.  stream_add: extra tensor added to the cross-attention
        output (the SOS update s + s_mutual of reference backbones.py:220-221), fused into the GEMM."""
        ops = _ops()
        # without query_pos (the SOS layers) ln1 / ln2 feed only the depthwise-conv branches of the attention modules
        # and are applied inside that kernel; with query_pos (predictor) they stay separate kernels
        fuse1 = query_pos is None and isinstance(self.self_attn, MaskedMHCA_QKV)
        fuse2 = query_pos is None and isinstance(self.multihead_attn, MaskedMHCA_QKV)
        last = not self.with_ffn
        if fuse1:
            tgt, _ = self.self_attn.cl_qkv(tgt, tgt, tgt, tgt_mask, tgt_mask, pre_ln=(self.ln1.weight, self.ln1.bias),
                                           pre_ln_on="qk", scale=self._scale(self.drop_path_attn1),
                                           row_scale=self._drop(self.drop_path_attn1, tgt), res=tgt, res_masked=True)
        else:
            t2 = self.ln1.cl(tgt, post_add=query_pos)
            tgt, _ = self.self_attn.cl_qkv(t2, t2, tgt, tgt_mask, tgt_mask, scale=self._scale(self.drop_path_attn1),
                                           row_scale=self._drop(self.drop_path_attn1, tgt), res=tgt, res_masked=True)
        kw = dict(scale=self._scale(self.drop_path_attn2), row_scale=self._drop(self.drop_path_attn2, tgt), res=tgt,
                  res_masked=True, res2=stream_add if last else None, out=out if last else None)
        if fuse2:
            tgt, _ = self.multihead_attn.cl_qkv(tgt, memory, memory, tgt_mask, memory_mask,
                                                pre_ln=(self.ln2.weight, self.ln2.bias), pre_ln_on="q", **kw)
        else:
            t2 = self.ln2.cl(tgt, post_add=query_pos)
            tgt, _ = self.multihead_attn.cl_qkv(t2, memory, memory, tgt_mask, memory_mask, **kw)
        if self.with_ffn:
            assert stream_add is None
            t2 = self.ln3.cl(tgt)
            h = ops.conv_gemm(t2, self.mlp[0].weight, self.mlp[0].bias, act=ops.ACT_GELU)
            tgt = ops.conv_gemm(h, self.mlp[3].weight, self.mlp[3].bias, row_mask=tgt_mask, scale=self._scale(self.drop_path_mlp),
                                row_scale=self._drop(self.drop_path_mlp, tgt), res=tgt, out=out)
        return tgt, tgt_mask

    def forward(self, tgt, memory, tgt_mask: Optional[Tensor] = None, memory_mask: Optional[Tensor] = None,
                pos: Optional[Tensor] = None, query_pos: Optional[Tensor] = None, cross_first: bool = False,
                attn_mask: Optional[Tensor] = None):
        assert pos is None and not cross_first and attn_mask is None, "only the call form used on the path is built"
        qp = None
        if query_pos is not None:      # (B, C, Tq), identical over B on the path (predictor query embedding)
            qp = query_pos[0].t().contiguous()
        tm = _mask2d(tgt_mask)
        mm = tm if memory_mask is tgt_mask else _mask2d(memory_mask)
        y, _ = self.cl(_to_cl(tgt), _to_cl(memory), tm, mm, query_pos=qp)
        return _from_cl(y), tgt_mask


class MaskedConvTransformerDecoder(nn.Module):
    """Stack of decoder layers with a shared output norm; reference models/local_transformer.py:838-905."""

    def __init__(self, n_embd, n_head, n_hidden, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.1, n_qx_stride=0,
                 n_kv_stride=1, num_layers=4, norm=None, return_intermediate=False, use_local=False, win_size=None,
                 use_rel_pe=False):
        super().__init__()
        layer = MaskedConvTransformerDecoderLayer(n_embd, n_head, n_hidden, attn_pdrop=attn_pdrop,
                                                  proj_pdrop=proj_pdrop, path_pdrop=path_pdrop,
                                                  n_qx_stride=n_qx_stride, n_kv_stride=n_kv_stride,
                                                  use_local=use_local, win_size=win_size, use_rel_pe=use_rel_pe)
        self.layers = _get_clones(layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def cl(self, tgt, memory, memory_mask, query_pos, all_layers):
        """Returns the normalised output of every layer (all_layers) or of the last one only."""
        outs = []
        for i, layer in enumerate(self.layers):
            tgt, _ = layer.cl(tgt, memory, None, memory_mask, query_pos=query_pos)
            if all_layers or i == self.num_layers - 1:
                outs.append(self.norm.cl(tgt))
        return outs


class MaskedConvTransformerDecoderOnly(nn.Module):
    """Query decoder of the predictor; reference models/local_transformer.py:908-976."""

    def __init__(self, n_embd, n_head, n_hidden, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.1, n_qx_stride=0,
                 n_kv_stride=1, num_layers=4, return_intermediate=False, use_local=False, win_size=None,
                 use_rel_pe=False):
        super().__init__()
        self.decoder = MaskedConvTransformerDecoder(n_embd, n_head, n_hidden, attn_pdrop=attn_pdrop,
                                                    proj_pdrop=proj_pdrop, path_pdrop=path_pdrop,
                                                    n_qx_stride=n_qx_stride, n_kv_stride=n_kv_stride,
                                                    num_layers=num_layers, norm=LayerNorm(n_embd),
                                                    return_intermediate=return_intermediate, use_local=use_local,
                                                    win_size=win_size, use_rel_pe=use_rel_pe)
        self.n_embd, self.n_head = n_embd, n_head
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Conv1d)) and m.bias is not None:
                nn.init.zeros_(m.bias)

    def cl(self, src, src_mask, query_embed, all_layers):
        """src (B, Tk, C), query_embed (Q, C) -> list of (B, Q, C) normalised layer outputs."""
        B = src.shape[0]
        tgt = torch.zeros(B, query_embed.shape[0], query_embed.shape[1], device=src.device, dtype=torch.float32)
        return self.decoder.cl(tgt, src, src_mask, query_embed, all_layers)
