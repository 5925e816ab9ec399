"""Operators of the relation-encoding path: same class names, constructor arguments and
parameter trees as the reference's models/blocks.py (so checkpoints, ModelEma and the
optimizer's weight-decay grouping in utils/train_utils.py:44-65 work unchanged), with the
forward of every module running HIP kernels (vrdone_amd.ops -> libvrdone_hip.so).

Two call forms per module:
  * ``forward(...)``  the reference's signature on ``(B, C, T)`` tensors / ``(B, 1, T)`` bool
    masks (a layout change is done around the channels-last core; used by module-level tests),
  * ``cl(...)``       the channels-last core on ``(B, T, C)`` tensors / ``(B, T)`` masks that
    MaskVRD chains end to end without any transposes.
Under torch.no_grad() (eval, validation) the modules run the fused inference kernels; with autograd recording
they run the differentiable composition of vrdone_amd/autograd.py (HIP forward and backward kernels, plain f32 rows),
and in training mode AffineDropPath samples its per-sample keep factors (reference blocks.py:1107-1120).
"""
import math

import torch
from torch import nn

from .transformer import _get_activation_fn  # noqa: F401  (re-exported like the reference)


def _ops():
    from .. import ops      # deferred: constructing / loading a model needs no GPU
    return ops


def _mask2d(mask):
    """(B, 1, T) bool -> contiguous (B, T) bool."""
    return mask.reshape(mask.shape[0], mask.shape[-1]).contiguous()


def _from_cl(x_cl):
    return _ops().btc_to_bct(x_cl)


def _to_cl(x):
    return _ops().to_channels_last(x)


class MaskedConv1D(nn.Module):
    """Conv1d followed by the (down-sampled) mask; reference models/blocks.py:63-113."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, padding_mode='zeros'):
        super().__init__()
        assert kernel_size % 2 == 1 and kernel_size // 2 == padding
        self.stride = stride
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                              padding_mode)
        if bias:
            nn.init.zeros_(self.conv.bias)

    def cl(self, x, mask, out=None):
        """x (B, T, Cin), mask (B, T) -> (y (B, T/stride, Cout), mask_out)."""
        ops = _ops()
        conv = self.conv
        assert x.shape[1] % self.stride == 0
        m_out = mask if self.stride == 1 else mask[:, ::self.stride].contiguous()
        if conv.groups == 1:
            assert self.stride == 1, "dense strided conv is not on the path"
            y = ops.conv_gemm(x, conv.weight, conv.bias, row_mask=m_out, out=out)
        else:
            assert conv.groups == conv.out_channels, "only depthwise / 2-in-per-group convs are on the path"
            y, = ops.dwconv_ln(x, [dict(weight=conv.weight, bias=conv.bias, out=out)], mask_out=m_out,
                               stride=self.stride)
        return y, m_out

    def forward(self, x, mask, downsample=True):
        assert downsample or self.stride == 1
        y, m = self.cl(_to_cl(x), _mask2d(mask))
        return _from_cl(y), m[:, None, :]


class LayerNorm(nn.Module):
    """LayerNorm over the channel axis of (B, C, T); reference models/blocks.py:116-158."""

    def __init__(self, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        super().__init__()
        assert affine and eps == 1e-5, "the fused kernels implement the affine, eps=1e-5 form used on the path"
        self.num_channels, self.eps, self.affine = num_channels, eps, affine
        kw = {'device': device, 'dtype': dtype}
        self.weight = nn.Parameter(torch.ones(1, num_channels, 1, **kw))
        self.bias = nn.Parameter(torch.zeros(1, num_channels, 1, **kw))

    def cl(self, x, relu=False, post_add=None, out=None, pair=False):
        """pair=True: emit GEMM-operand pair rows (only when the sole consumer is a conv GEMM)."""
        return _ops().layernorm(x, self.weight, self.bias, relu=relu, post_add=post_add, out=out, pair=pair)

    def forward(self, x):
        assert x.dim() == 3 and x.shape[1] == self.num_channels
        return _from_cl(self.cl(_to_cl(x)))


def get_sinusoid_encoding(n_position, d_hid):
    """(1, C, T) sinusoid position table (reference models/blocks.py:161-172): channel 2i = sin(t / 10000^(2i/C)), channel
    2i+1 = cos of the same angle, computed in float64 and rounded to float32."""
    import numpy as np
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    angle = pos / np.power(10000.0, 2 * (j // 2) / d_hid)
    return torch.tensor(np.where(j % 2 == 0, np.sin(angle), np.cos(angle)), dtype=torch.float32).unsqueeze(0).transpose(1, 2)


class ConvMLP(nn.Module):
    """1x1-conv MLP with GELU between layers; reference models/blocks.py:37-61."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers, kernel_size=1, act_layer='gelu', drop=0.0,
                 with_bias=True):
        super().__init__()
        assert kernel_size == 1 and act_layer == 'gelu' and drop == 0.0
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Conv1d(a, b, 1, bias=with_bias) for a, b in zip(dims[:-1], dims[1:]))
        for layer in self.layers:
            if layer.bias is not None:
                nn.init.zeros_(layer.bias)

    def cl(self, x, row_mask=None, out=None, out_pair=False, res=None):
        """x: tensor or ops.Pair.  Hidden activations only feed the next GEMM, so they travel as pair rows.
        res: rows added to the (masked) output of the last layer."""
        ops = _ops()
        last = self.num_layers - 1
        for i, layer in enumerate(self.layers):
            if i < last:
                x = ops.conv_gemm(x, layer.weight, layer.bias, act=ops.ACT_GELU, out_pair=ops.pair_mode(), skip_rows=row_mask)
            else:
                x = ops.conv_gemm(x, layer.weight, layer.bias, row_mask=row_mask, out=out, out_pair=out_pair, res=res)
        return x

    def forward(self, x):
        return _from_cl(self.cl(_to_cl(x)))


class _FactorPool:
    """Stochastic-depth factors floor(keep_prob + U[0, 1)) / keep_prob (reference drop_path, blocks.py:1107-1120), drawn POOL at
    a time and handed out in slices: a training step has ~30 AffineDropPath modules, and drawing per module was four tiny
    launches each (rand, add, floor, div) -- 120 of the step's ~560 tensor-op launches, each a node of a recorded step.
    A pool drawn while a HIP graph is being recorded belongs to that recording (its `rand` is replayed, so every replay gets
    fresh factors; the slices recorded into the graph point into it) and is dropped when the capture state changes, as
    autograd._ZeroArena's blocks are."""
    POOL = 4096

    def __init__(self):
        self.pools = {}                # (device, keep_prob) -> [factors, next index, drawn during a capture]

    def take(self, n, keep_prob, device):
        capturing = torch.cuda.is_current_stream_capturing()
        key = (device, keep_prob)
        ent = self.pools.get(key)
        if ent is None or ent[2] != capturing or ent[1] + n > ent[0].numel():
            size = max(self.POOL, n)
            ent = self.pools[key] = [torch.floor(keep_prob + torch.rand(size, device=device)) / keep_prob, 0, capturing]
        out = ent[0][ent[1]:ent[1] + n]
        ent[1] += n
        return out


_factor_pool = _FactorPool()


class AffineDropPath(nn.Module):
    """Per-channel scale (+ stochastic depth when training); reference models/blocks.py:1134-1149.
    In eval it is a channel scale, which the GEMM epilogue applies (``scale`` argument)."""

    def __init__(self, num_dim, drop_prob=0.0, init_scale_value=1e-4):
        super().__init__()
        self.scale = nn.Parameter(init_scale_value * torch.ones(1, num_dim, 1))
        self.drop_prob = drop_prob
        self.keep = None            # tests may pin the per-sample keep decisions: a 0/1 vector (>= B entries) used instead of sampling

    def row_factors(self, n_samples, rows_per_sample, device):
        """Stochastic depth per sample (reference drop_path, blocks.py:1107-1120): factor_b = floor(keep_prob + U[0,1))
        / keep_prob, one per sample, expanded to one per row of the (n_samples * rows_per_sample, C) branch; None when
        nothing is dropped (eval, drop_prob 0, or autograd not recording)."""
        if not self.training or self.drop_prob == 0.0 or not torch.is_grad_enabled():
            return None
        keep_prob = 1.0 - self.drop_prob
        if self.keep is not None:
            assert self.keep.numel() >= n_samples
            factors = self.keep[:n_samples].to(device=device, dtype=torch.float32) / keep_prob
        else:
            factors = _factor_pool.take(n_samples, keep_prob, torch.device(device))
        return factors[:, None].expand(n_samples, rows_per_sample).contiguous().view(-1)

    def forward(self, x):
        """(B, C, T) * scale, with the training-time drop (plain tensor arithmetic: the hot path applies scale and
        factors inside the output-projection / MLP GEMM's epilogue instead, see TransformerBlock.cl)."""
        y = x * self.scale
        rf = self.row_factors(x.shape[0], 1, x.device)
        return y if rf is None else y * rf.view(-1, 1, 1)


class Scale(nn.Module):
    """Learnable scalar (reference models/blocks.py:1084-1102); not on the hot path, kept because
    utils/train_utils.py:8 imports it."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(init_value, dtype=torch.float32))

    def forward(self, x):
        return x * self.scale


class _ConvAttention(nn.Module):
    """Parameter tree shared by the conv-attention modules: depthwise conv + LayerNorm per
    q/k/v branch, then 1x1 projections (creation order = reference state_dict order)."""

    def _build(self, n_embd, n_head, q_kernel, kv_kernel, stride):
        assert n_embd % n_head == 0
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        for name, ks in (("query", q_kernel), ("key", kv_kernel), ("value", kv_kernel)):
            setattr(self, f"{name}_conv", MaskedConv1D(n_embd, n_embd, ks, stride=stride, padding=ks // 2,
                                                        groups=n_embd, bias=False))
            setattr(self, f"{name}_norm", LayerNorm(n_embd))
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.attn_drop, self.proj_drop = nn.Dropout(0.0), nn.Dropout(0.0)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)

    def _branch_set(self, name):
        conv, norm = getattr(self, f"{name}_conv"), getattr(self, f"{name}_norm")
        # the branch output feeds only its 1x1 projection GEMM
        return dict(weight=conv.conv.weight, gamma=norm.weight, beta=norm.bias, pair=_ops().pair_mode())

    def _groups(self, q_in, k_in, v_in, q_mask, kv_mask, pre_ln=None, pre_ln_on="qkv"):
        """[(x, mask, use_ln, kernel size, [branch names])]: branches that read the same rows (same input tensor, mask, kernel
        size and input-LayerNorm choice) share one dwconv_ln launch."""
        specs = [("query", q_in, q_mask, "q"), ("key", k_in, kv_mask, "k"), ("value", v_in, kv_mask, "v")]
        groups = []
        for name, x, m, tag in specs:
            ks = getattr(self, f"{name}_conv").conv.kernel_size
            use_ln = pre_ln is not None and tag in pre_ln_on
            for g in groups:
                if g[0] is x and g[1] is m and g[2] == use_ln and g[3] == ks:
                    g[4].append(name)
                    break
            else:
                groups.append((x, m, use_ln, ks, [name]))
        return groups

    def _prep(self, q_in, k_in, v_in, q_mask, kv_mask, stride=1, pre_ln=None, pre_ln_on="qkv"):
        """dwconv * mask -> LN for the three branches (one launch per group of _groups).
        pre_ln (gamma, beta): inputs named in pre_ln_on ('q', 'k', 'v') are LayerNorm'ed as they are read."""
        ops = _ops()
        outs = {}
        for x, m, use_ln, _, names in self._groups(q_in, k_in, v_in, q_mask, kv_mask, pre_ln, pre_ln_on):
            res = ops.dwconv_ln(x, [self._branch_set(n) for n in names], mask_out=m, stride=stride,
                                pre_ln=pre_ln if use_ln else None)
            outs.update(zip(names, res))
        return outs["query"], outs["key"], outs["value"]

    def _project(self, q, k, v, out_pair=False, q_mask=None, kv_mask=None):
        """1x1 projections.  q_mask / kv_mask: validity of the rows; fully padded row blocks are not contracted (their
        keys / values are masked inside the attention kernels, their query rows by the output projection's row mask)."""
        ops = _ops()
        return tuple(ops.conv_gemm_batch([
            ((q, self.query.weight, self.query.bias), dict(out_pair=out_pair, skip_rows=q_mask)),
            ((k, self.key.weight, self.key.bias), dict(out_pair=out_pair, skip_rows=kv_mask)),
            ((v, self.value.weight, self.value.bias), dict(out_pair=out_pair, skip_rows=kv_mask))]))


def make_rel_pe(n_embd, n_head, window_size):
    """The learnable bias of the window scores, one value per (head, window slot): shape and initialisation of reference
    models/blocks.py:739-743 (truncated normal, std sqrt(2 / n_embd), cut at +-2)."""
    rel_pe = nn.Parameter(torch.zeros(1, 1, n_head, window_size))
    nn.init.trunc_normal_(rel_pe, std=(2.0 / n_embd) ** 0.5)
    return rel_pe


class LocalMaskedMHCA(_ConvAttention):
    """Banded-window conv attention; reference models/blocks.py:656-989."""

    def __init__(self, n_embd, n_head, window_size, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0,
                 use_rel_pe=False):
        super().__init__()
        assert window_size > 1 and window_size % 2 == 1
        assert n_qx_stride == n_kv_stride and n_kv_stride in (1, 2)
        assert attn_pdrop == 0.0 and proj_pdrop == 0.0
        self.window_size, self.window_overlap = window_size, window_size // 2
        self.use_rel_pe = use_rel_pe
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        ks = n_kv_stride + 1 if n_kv_stride > 1 else 3
        self._build(n_embd, n_head, ks, ks, n_kv_stride)
        self.rel_pe = make_rel_pe(n_embd, n_head, window_size) if use_rel_pe else None

    def cl(self, x, mask, mask_out=None, pre_ln=None, **epilogue):
        """x = LN1 output (B, T, C), or the block input with pre_ln = (ln1.weight, ln1.bias) applied inside the
        depthwise-conv kernel; epilogue kwargs go to the output-projection GEMM."""
        ops = _ops()
        s = self.n_kv_stride
        if mask_out is None:
            mask_out = mask if s == 1 else mask[:, ::s].contiguous()
        # (the reference's sliding-chunk form needs T / s to be a multiple of 2 * window_overlap, blocks.py:828, and its callers
        # pad for that; the banded kernel here does not: MaskVRD's tight padding runs pairs at shorter padded lengths)
        q, k, v = self._prep(x, x, x, mask_out, mask_out, stride=s, pre_ln=pre_ln)
        q, k, v = self._project(q, k, v, q_mask=mask_out, kv_mask=mask_out)
        att = ops.local_attention(q, k, v, mask_out, self.n_head, self.window_overlap, pair=ops.pair_mode(),
                                  rel_pe=self.rel_pe)
        return ops.conv_gemm(att, self.proj.weight, self.proj.bias, row_mask=mask_out, **epilogue), mask_out

    def forward(self, x, mask):
        assert (x.shape[-1] // self.n_kv_stride) % (2 * self.window_overlap) == 0      # reference blocks.py:828
        y, m = self.cl(_to_cl(x), _mask2d(mask))
        return _from_cl(y), m[:, None, :]


class MaskedMHCA(_ConvAttention):
    """Conv attention with FULL (global) masked attention: what TransformerBlock uses when n_mha_win_size <= 1;
    reference models/blocks.py:245-359.  Same parameter tree as LocalMaskedMHCA; the attention core is the global
    flash kernel of the SOS layers (keys outside the mask get -inf, :337)."""

    def __init__(self, n_embd, n_head, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_qx_stride == n_kv_stride and n_kv_stride in (1, 2), "built for equal query / key-value strides 1 or 2"
        assert attn_pdrop == 0.0 and proj_pdrop == 0.0
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        ks = n_kv_stride + 1 if n_kv_stride > 1 else 3
        self._build(n_embd, n_head, ks, ks, n_kv_stride)

    def cl(self, x, mask, mask_out=None, pre_ln=None, **epilogue):
        ops = _ops()
        s = self.n_kv_stride
        if mask_out is None:
            mask_out = mask if s == 1 else mask[:, ::s].contiguous()
        q, k, v = self._prep(x, x, x, mask_out, mask_out, stride=s, pre_ln=pre_ln)
        qkv_pair = ops.flash_pair_ok(self.n_head, self.n_embd, q.shape[1])
        q, k, v = self._project(q, k, v, out_pair=qkv_pair, q_mask=mask_out, kv_mask=mask_out)
        att = ops.attention(q, k, v, mask_out, self.n_head, pair=ops.pair_mode(), q_mask=mask_out)
        return ops.conv_gemm(att, self.proj.weight, self.proj.bias, row_mask=mask_out, **epilogue), mask_out

    def forward(self, x, mask):
        y, m = self.cl(_to_cl(x), _mask2d(mask))
        return _from_cl(y), m[:, None, :]


class MaskedMHA(nn.Module):
    """Plain masked multi-head attention parameters; reference models/blocks.py:177-242."""

    def __init__(self, n_embd, n_head, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__()
        assert n_embd % n_head == 0 and attn_pdrop == 0.0 and proj_pdrop == 0.0
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.attn_drop, self.proj_drop = nn.Dropout(0.0), nn.Dropout(0.0)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)

    def cl_qkv(self, q_in, k_in, v_in, q_mask, kv_mask, **epilogue):
        ops = _ops()
        q = ops.conv_gemm(q_in, self.query.weight, self.query.bias)
        k = ops.conv_gemm(k_in, self.key.weight, self.key.bias)
        v = ops.conv_gemm(v_in, self.value.weight, self.value.bias)
        att = ops.attention(q, k, v, kv_mask, self.n_head)
        return ops.conv_gemm(att, self.proj.weight, self.proj.bias, row_mask=q_mask, **epilogue), q_mask

    def forward(self, x, mask):
        m = _mask2d(mask)
        xc = _to_cl(x)
        y, _ = self.cl_qkv(xc, xc, xc, m, m)
        return _from_cl(y), mask


class TransformerBlock(nn.Module):
    """LN -> local conv attention -> (max-pool) skip, LN -> MLP; reference models/blocks.py:992-1080."""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), n_out=None, n_hidden=None, act_layer=nn.GELU,
                 attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, mha_win_size=-1, use_rel_pe=False):
        super().__init__()
        assert len(n_ds_strides) == 2 and act_layer is nn.GELU
        self.ln1 = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        if mha_win_size > 1:
            self.attn = LocalMaskedMHCA(n_embd, n_head, window_size=mha_win_size, n_qx_stride=n_ds_strides[0],
                                        n_kv_stride=n_ds_strides[1], attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                        use_rel_pe=use_rel_pe)
        else:       # reference blocks.py:1029-1036 (no shipped config takes this branch)
            self.attn = MaskedMHCA(n_embd, n_head, n_qx_stride=n_ds_strides[0], n_kv_stride=n_ds_strides[1],
                                   attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop)
        s = n_ds_strides[0]
        self.pool_skip = nn.MaxPool1d(s + 1, stride=s, padding=(s + 1) // 2) if s > 1 else nn.Identity()
        n_hidden = n_hidden or 4 * n_embd
        n_out = n_out or n_embd
        assert n_out == n_embd
        self.mlp = nn.Sequential(nn.Conv1d(n_embd, n_hidden, 1), act_layer(), nn.Dropout(proj_pdrop, inplace=True),
                                 nn.Conv1d(n_hidden, n_out, 1), nn.Dropout(proj_pdrop, inplace=True))
        if path_pdrop > 0.0:
            self.drop_path_attn = AffineDropPath(n_embd, drop_prob=path_pdrop)
            self.drop_path_mlp = AffineDropPath(n_out, drop_prob=path_pdrop)
        else:
            self.drop_path_attn = nn.Identity()
            self.drop_path_mlp = nn.Identity()

    @staticmethod
    def _scale(dp):
        return dp.scale if isinstance(dp, AffineDropPath) else None

    @staticmethod
    def _drop(dp, mask):
        """Per-row stochastic-depth factors of a branch whose rows follow `mask` (B, T'), or None."""
        return dp.row_factors(mask.shape[0], mask.shape[1], mask.device) if isinstance(dp, AffineDropPath) else None

    def cl(self, x, mask, out=None):
        ops = _ops()
        if self.attn.n_kv_stride > 1:
            skip, m_out = ops.maxpool_mask(x, mask)
        else:
            skip, m_out = x, mask
        # ln1 is applied to the rows inside the depthwise-conv kernel (its only consumer)
        y, _ = self.attn.cl(x, mask, m_out, pre_ln=(self.ln1.weight, self.ln1.bias),
                            scale=self._scale(self.drop_path_attn), row_scale=self._drop(self.drop_path_attn, m_out),
                            res=skip, res_masked=True)
        h = self.ln2.cl(y, pair=ops.pair_mode())
        h = ops.conv_gemm(h, self.mlp[0].weight, self.mlp[0].bias, act=ops.ACT_GELU, out_pair=ops.pair_mode(), skip_rows=m_out)
        y = ops.conv_gemm(h, self.mlp[3].weight, self.mlp[3].bias, row_mask=m_out, scale=self._scale(self.drop_path_mlp),
                          row_scale=self._drop(self.drop_path_mlp, m_out), res=y, out=out)
        return y, m_out

    def forward(self, x, mask, pos_embd=None):
        assert pos_embd is None
        y, m = self.cl(_to_cl(x), _mask2d(mask))
        return _from_cl(y), m[:, None, :]
