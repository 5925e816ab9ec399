"""Per-operator parity on a real MI355X: the HIP kernels (through the C ABI) against the
reference's outputs stored in tests/golden/ops.npz and against the CPU oracle.

Module-level tests use the reference's call signatures on (B, C, T) tensors, so they read like
the reference's own usage.  Tolerances are absolute, for O(1) activations in fp32."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import vrd_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


SPLIT = ("bf16x3", "f16x3")        # the split-precision modes (pair rows, LDS-DMA GEMMs, split flash attention)


@pytest.fixture(autouse=True, params=["bf16x3", "f16x3", "f32"])
def precision(request):
    """Every operator test runs in all GEMM precision modes (tolerances stated for f32 products)."""
    from vrdone_amd import ops
    global GEMM_TOL_SCALE
    old = ops.get_precision()
    ops.set_precision(request.param)
    GEMM_TOL_SCALE = 5.0 if request.param == "bf16x3" else 1.0          # f16x3 products: held to the f32 tolerances
    yield request.param
    ops.set_precision(old)
    GEMM_TOL_SCALE = 1.0


GEMM_TOL_SCALE = 1.0      # tolerances below are stated for f32 products; bf16x3 products get 5x


def seeded(module, prefix):
    keys = [(f"{prefix}.{k}", tuple(v.shape)) for k, v in module.state_dict().items()]
    sd = O.synth_state_dict(keys)
    module.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()}, strict=True)
    return module.eval().to(DEV), sd


@pytest.fixture(scope="module")
def g():
    d = dict(np.load(os.path.join(GOLDEN, "ops.npz")))
    x, y = torch.from_numpy(d["x"]), torch.from_numpy(d["y"])
    lens = torch.from_numpy(d["lens"])
    m = (torch.arange(x.shape[-1])[None] < lens[:, None])[:, None]
    d.update(xt=x, yt=y, mt=m)
    return d


def close(got, want, atol, gemm=False):
    """gemm=True: the value went through conv GEMMs, whose products follow the precision mode."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    want = want.detach().cpu().numpy() if isinstance(want, torch.Tensor) else want
    assert got.shape == want.shape
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, atol=atol * (GEMM_TOL_SCALE if gemm else 1.0), rtol=0)


# ------------------------------------------------------------------------------------------------
# modules vs reference outputs
# ------------------------------------------------------------------------------------------------
def test_layernorm_module(g):
    from vrdone_amd.models.blocks import LayerNorm
    mod, _ = seeded(LayerNorm(512), "op.ln")
    close(mod(g["xt"].to(DEV)), g["ln"], 2e-6)


def test_masked_conv1d_dense_k3(g):
    from vrdone_amd.models.blocks import MaskedConv1D
    mod, _ = seeded(MaskedConv1D(512, 512, 3, padding=1, bias=False), "op.conv3")
    out, m = mod(g["xt"].to(DEV), g["mt"].to(DEV))
    close(out, g["conv3"], 2e-5, gemm=True)
    assert torch.equal(m.cpu(), g["mt"])


@pytest.mark.parametrize("stride", [1, 2])
def test_local_mhca(g, stride):
    from vrdone_amd.models.blocks import LocalMaskedMHCA
    mod, _ = seeded(LocalMaskedMHCA(512, 4, window_size=7, n_qx_stride=stride, n_kv_stride=stride),
                    f"op.local_mhca_s{stride}")
    out, m = mod(g["xt"].to(DEV), g["mt"].to(DEV))
    close(out, g[f"local_mhca_s{stride}"], 5e-5, gemm=True)
    assert torch.equal(m.cpu(), g["mt"][..., ::stride])


def test_local_mhca_window9_heads8(g):
    from vrdone_amd.models.blocks import LocalMaskedMHCA
    mod, _ = seeded(LocalMaskedMHCA(512, 8, window_size=9), "op.local_mhca_w9")
    close(mod(g["xt"].to(DEV), g["mt"].to(DEV))[0], g["local_mhca_w9"], 5e-5, gemm=True)


@pytest.mark.parametrize("stride", [1, 2])
def test_transformer_block(g, stride):
    from vrdone_amd.models.blocks import TransformerBlock
    mod, _ = seeded(TransformerBlock(512, 4, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=7),
                    f"op.block_s{stride}")
    out, m = mod(g["xt"].to(DEV), g["mt"].to(DEV))
    close(out, g[f"block_s{stride}"], 1e-4, gemm=True)


@pytest.mark.parametrize("stride", [1, 2])
def test_transformer_block_global_attention(g, stride):
    """TransformerBlock with n_mha_win_size <= 1: MaskedMHCA (global conv attention, reference blocks.py:245-359,
    1029-1036) vs the reference's own output (every 4th channel stored)."""
    from vrdone_amd.models.blocks import MaskedMHCA, TransformerBlock
    mod, _ = seeded(TransformerBlock(512, 4, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=-1),
                    f"op.block_global_s{stride}")
    assert isinstance(mod.attn, MaskedMHCA)
    g2 = np.load(os.path.join(GOLDEN, "ops_r2.npz"))
    out, m = mod(g["xt"].to(DEV), g["mt"].to(DEV))
    close(out[:, ::4], g2[f"block_global_s{stride}"], 1e-4, gemm=True)
    assert torch.equal(m.cpu(), g["mt"][..., ::stride])


def test_mhca_qkv_global(g):
    from vrdone_amd.models.local_transformer import MaskedMHCA_QKV
    mod, _ = seeded(MaskedMHCA_QKV(512, 4, n_qx_stride=1, n_kv_stride=1), "op.mhca_qkv")
    x, y, m = g["xt"].to(DEV), g["yt"].to(DEV), g["mt"].to(DEV)
    close(mod(x, y, y, m, m)[0], g["mhca_qkv"], 5e-5, gemm=True)


@pytest.mark.parametrize("name,heads,local", [("sos", 4, False), ("sos_local", 8, True)])
def test_sos_decoder_layer(g, name, heads, local):
    from vrdone_amd.models.local_transformer import MaskedConvTransformerDecoderLayer
    mod, _ = seeded(MaskedConvTransformerDecoderLayer(512, heads, path_pdrop=0.1, n_qx_stride=1, n_kv_stride=1,
                                                      with_ffn=False, use_local=local, win_size=9 if local else None),
                    f"op.{name}")
    x, y, m = g["xt"].to(DEV), g["yt"].to(DEV), g["mt"].to(DEV)
    close(mod(x, y, m, m)[0], g[name], 1e-4, gemm=True)


# ------------------------------------------------------------------------------------------------
# kernels vs oracle / plain torch on awkward shapes
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,Cin,taps,T", [
    (1000, 512, 512, 1, 1000),      # ragged M
    (96 * 3, 133, 256, 1, 96),      # class head: N not a tile multiple
    (96 * 2, 512, 5, 3, 96),        # bbox_so_embd: scalar-load path, K = 15
    (48 * 4, 512, 8, 3, 48),        # bbox_entity_embd: K = 24
    (144 * 2, 512, 1024, 3, 144),   # visual_embd[0]
    (7, 64, 36, 1, 7),              # tiny
    (130, 2048, 512, 1, 130),       # MLP up-projection
])
def test_gemm_shapes_and_epilogue(M, N, Cin, taps, T):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(M * 31 + N)
    B = M // T
    x = torch.randn(B, T, Cin, generator=gen)
    w = torch.randn(N, Cin, taps, generator=gen) / (Cin * taps) ** 0.5
    bias = torch.randn(N, generator=gen)
    scale = torch.rand(N, generator=gen) + 0.5
    res = torch.randn(B, T, N, generator=gen)
    res2 = torch.randn(B, T, N, generator=gen)
    mask = torch.rand(B, T, generator=gen) > 0.3
    ref = torch.nn.functional.conv1d(x.transpose(1, 2), w, bias, padding=taps // 2).transpose(1, 2)
    mf = mask[..., None].float()
    # plain
    close(ops.conv_gemm(x.to(DEV), w.to(DEV), bias.to(DEV)), ref, 3e-5, gemm=True)
    # full epilogue
    want = torch.nn.functional.gelu(ref) * mf * scale + res * mf + res2
    got = ops.conv_gemm(x.to(DEV), w.to(DEV), bias.to(DEV), act=ops.ACT_GELU, row_mask=mask.to(DEV),
                        scale=scale.to(DEV), res=res.to(DEV), res_masked=True, res2=res2.to(DEV))
    close(got, want, 3e-5, gemm=True)
    # relu, unmasked residual, output into a column slab of a wider buffer
    buf = torch.zeros(B, T, N + 64, device=DEV)
    ops.conv_gemm(x.to(DEV), w.to(DEV), None, act=ops.ACT_RELU, res=res.to(DEV), out=buf[..., 64:])
    want = torch.relu(torch.nn.functional.conv1d(x.transpose(1, 2), w, None, padding=taps // 2).transpose(1, 2)) + res
    close(buf[..., 64:], want, 3e-5, gemm=True)
    assert float(buf[..., :64].abs().sum()) == 0.0


def test_gemm_reads_column_slab_input():
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(5)
    wide = torch.randn(3, 40, 1024, generator=gen)
    w = torch.randn(256, 512, 1, generator=gen) / 512 ** 0.5
    got = ops.conv_gemm(wide.to(DEV)[..., 512:], w.to(DEV))
    close(got, torch.einsum("btc,nc->btn", wide[..., 512:], w[..., 0]), 3e-5, gemm=True)


@pytest.mark.parametrize("C", [256, 512])
def test_layernorm_kernel_options(C):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(5, 9, C, generator=gen) * 3 + 1
    gam, bet = torch.randn(1, C, 1, generator=gen), torch.randn(1, C, 1, generator=gen)
    pos = torch.randn(9, C, generator=gen)
    ref = O.channel_ln(x.transpose(1, 2), gam, bet).transpose(1, 2)
    close(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV)), ref, 3e-6)
    close(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), relu=True), torch.relu(ref), 3e-6)
    close(ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), post_add=pos.to(DEV)), ref + pos[None], 3e-6)
    z = torch.zeros(5, 9, C)          # masked rows: LN(0) = beta
    close(ops.layernorm(z.to(DEV), gam.to(DEV), bet.to(DEV)), bet.view(1, 1, C).expand(5, 9, C), 0)


@pytest.mark.parametrize("C,ks,stride,gin,up", [(512, 3, 1, 1, False), (512, 3, 2, 1, False), (256, 3, 1, 1, True),
                                                 (256, 3, 1, 2, False), (256, 1, 1, 1, False), (512, 1, 1, 1, False)])
def test_dwconv_ln_variants(C, ks, stride, gin, up):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(C + ks + stride + gin)
    B, T = 3, 24
    x = torch.randn(B, T, C * gin, generator=gen)
    xu = torch.randn(B, T // 2, C * gin, generator=gen) if up else None
    lens = torch.tensor([24, 13, 2])
    mask = torch.arange(T)[None] < lens[:, None]
    m_out = mask[:, ::stride].contiguous()
    sets, wants = [], []
    xin = x + (xu.repeat_interleave(2, dim=1) if up else 0)
    for o in range(3 if gin == 1 else 1):
        w = torch.randn(C, gin, ks, generator=gen)
        b = torch.randn(C, generator=gen) if o == 1 else None
        gam = None if o == 2 else torch.randn(1, C, 1, generator=gen)
        bet = None if o == 2 else torch.randn(1, C, 1, generator=gen)
        y, _ = O.masked_conv1d(xin.transpose(1, 2), mask[:, None], w, b, stride=stride, groups=C)
        if gam is not None:
            y = O.channel_ln(y, gam, bet)
        if o == 0:
            y = torch.relu(y)
        wants.append(y.transpose(1, 2))
        dv = lambda t: None if t is None else t.to(DEV)     # noqa: E731
        sets.append(dict(weight=w.to(DEV), bias=dv(b), gamma=dv(gam), beta=dv(bet), relu=(o == 0)))
    outs = ops.dwconv_ln(x.to(DEV), sets, mask_out=m_out.to(DEV), stride=stride, x_up=None if xu is None else xu.to(DEV))
    for got, want in zip(outs, wants):
        close(got, want, 2e-5)


@pytest.mark.parametrize("C,stride,T", [(512, 1, 40), (512, 2, 48), (256, 1, 33)])
def test_dwconv_ln_input_layernorm(C, stride, T):
    """The block's ln1 applied inside the depthwise-conv kernel (rows normalised as they enter the window; the
    convolution's zero padding stays zero) against the oracle's LayerNorm -> conv -> LayerNorm chain, over
    strips that are full, partial and cross the 16-row strip length."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(C + stride + T)
    B = 3
    x = torch.randn(B, T, C, generator=gen) * 2 + 0.5
    lens = torch.tensor([T, T // 2 + 1, 2])
    mask = torch.arange(T)[None] < lens[:, None]
    m_out = mask[:, ::stride].contiguous()
    g0, b0 = torch.randn(1, C, 1, generator=gen), torch.randn(1, C, 1, generator=gen)
    h = O.channel_ln(x.transpose(1, 2), g0, b0)
    sets, wants = [], []
    for o in range(3):
        w = torch.randn(C, 1, 3, generator=gen)
        gam, bet = torch.randn(1, C, 1, generator=gen), torch.randn(1, C, 1, generator=gen)
        y, _ = O.masked_conv1d(h, mask[:, None], w, None, stride=stride, groups=C)
        wants.append(O.channel_ln(y, gam, bet).transpose(1, 2))
        sets.append(dict(weight=w.to(DEV), gamma=gam.to(DEV), beta=bet.to(DEV)))
    outs = ops.dwconv_ln(x.to(DEV), sets, mask_out=m_out.to(DEV), stride=stride, pre_ln=(g0.to(DEV), b0.to(DEV)))
    for got, want in zip(outs, wants):
        close(got, want, 3e-5)


@pytest.mark.parametrize("H,w", [(4, 3), (8, 3), (4, 4), (8, 4)])
def test_local_attention_kernel(H, w):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(H * 10 + w)
    B, T, C = 3, 8 * w, 512
    q, k, v = (torch.randn(B, T, C, generator=gen) for _ in range(3))
    lens = torch.tensor([T, T - 5, 1])
    mask = torch.arange(T)[None] < lens[:, None]
    want = O.banded_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H, w)
    got = ops.local_attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, w)
    close(got, want.transpose(1, 2), 2e-5)
    # strips that end inside the sequence, validity with holes (not a prefix), whole strips of padding, pair output
    B, T = 5, 50
    q, k, v = (torch.randn(B, T, C, generator=gen) for _ in range(3))
    mask = torch.rand(B, T, generator=gen) > 0.3
    mask[1, 16:32] = False
    mask[2] = False
    mask[2, 49] = True
    want = O.banded_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H, w).transpose(1, 2)
    got = ops.local_attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, w)
    close(got, want, 2e-5)
    got_pair = ops.local_attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, w, pair=True)
    assert float((got_pair.float() - got).abs().max()) <= 2 ** -15 * float(got.abs().max())
    # relative position bias (use_rel_pe): strip kernel and, VRD_LOCAL_STRIP=0 aside, the same arithmetic per row
    rel = torch.randn(1, 1, H, 2 * w + 1, generator=gen)
    want_rel = O.banded_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H, w, rel_pe=rel).transpose(1, 2)
    got_rel = ops.local_attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, w, rel_pe=rel.to(DEV))
    close(got_rel, want_rel, 2e-5)
    assert float((want_rel - want).abs().max()) > 0.1


@pytest.mark.parametrize("C,ks,stride,gin,up,pre", [(512, 3, 1, 1, False, True), (512, 3, 2, 1, False, True), (256, 3, 1, 1, True, False),
                                                     (256, 3, 1, 2, False, False), (512, 1, 1, 1, False, False)])
def test_dwconv_ln_over_row_groups_equals_the_calls_per_group(C, ks, stride, gin, up, pre):
    """vrd_dwconv_ln over a ragged row space (vrd_row_segs: groups of sequences of different lengths back to back, one
    launch) gives, bit for bit, what one call per group gives -- strides, the FPN's upsample-add, the 2-in-per-group conv,
    the input LayerNorm, pair-row outputs; 40 groups take two launches' worth of tables (the host side splits them)."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(C * 7 + ks + stride + gin)
    groups = [(5, 32), (1, 64), (7, 96), (3, 160), (2, 34 if stride == 1 and not up else 36)]
    R = sum(n * T for n, T in groups)
    x = torch.randn(1, R, C * gin, generator=gen).to(DEV)
    xu = torch.randn(1, R // 2, C * gin, generator=gen).to(DEV) if up else None
    mask = (torch.rand(1, R // stride, generator=gen) > 0.4).to(DEV)
    mask[0, :40] = False                                        # whole strips of padding
    dv = lambda *shape: torch.randn(*shape, generator=gen).to(DEV)     # noqa: E731
    sets = [dict(weight=dv(C, gin, ks), bias=dv(C) if o == 1 else None, gamma=None if o == 2 else dv(1, C, 1),
                 beta=None if o == 2 else dv(1, C, 1), relu=(o == 0), pair=(o == 1 and ops.pair_mode()))
            for o in range(3 if gin == 1 else 1)]
    pre_ln = (dv(1, C, 1), dv(1, C, 1)) if pre else None
    segs, off = [], 0
    for n, T in groups:
        segs.append((off, n, T))
        off += n * T
    got = ops.dwconv_ln(x, sets, mask_out=mask, stride=stride, x_up=xu, pre_ln=pre_ln, segs=segs)
    for off, n, T in segs:
        part = lambda t, d: None if t is None else t[0, off // d:(off + n * T) // d].unflatten(0, (n, T // d))   # noqa: E731
        want = ops.dwconv_ln(part(x, 1), sets, mask_out=part(mask, stride), stride=stride, x_up=part(xu, 2), pre_ln=pre_ln)
        for g, w in zip(got, want):
            raw = lambda t: t.t if isinstance(t, ops.Pair) else t     # noqa: E731
            assert torch.equal(part(raw(g), stride), raw(w))


@pytest.mark.parametrize("H,w,rel", [(4, 3, False), (8, 4, True)])
def test_local_attention_over_row_groups_equals_the_calls_per_group(H, w, rel):
    """vrd_local_attn_segs (one launch over groups of sequences of different lengths) against vrd_local_attn per group: the
    same bits; windows never reach across a sequence's end into the next group."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(H + w)
    groups = [(4, 32), (1, 50), (6, 96), (2, 17), (3, 288)]
    R = sum(n * T for n, T in groups)
    q, k, v = (torch.randn(1, R, 512, generator=gen).to(DEV) for _ in range(3))
    mask = (torch.rand(1, R, generator=gen) > 0.3).to(DEV)
    mask[0, 32:64] = False
    rel_pe = torch.randn(1, 1, H, 2 * w + 1, generator=gen).to(DEV) if rel else None
    segs, off = [], 0
    for n, T in groups:
        segs.append((off, n, T))
        off += n * T
    for pair in (False, ops.pair_mode()):
        got = ops.local_attention(q, k, v, mask, H, w, pair=pair, rel_pe=rel_pe, segs=segs)
        for off, n, T in segs:
            part = lambda t: t[0, off:off + n * T].unflatten(0, (n, T))     # noqa: E731
            want = ops.local_attention(part(q), part(k), part(v), part(mask), H, w, pair=pair, rel_pe=rel_pe)
            raw = lambda t: t.t if isinstance(t, ops.Pair) else t     # noqa: E731
            assert torch.equal(part(raw(got)), raw(want))


@pytest.mark.parametrize("algo", [1, 2])
@pytest.mark.parametrize("H,hd,Tq,Tk", [(4, 128, 96, 96), (8, 64, 144, 144), (4, 128, 288, 288), (8, 64, 512, 512),
                                        (4, 128, 40, 77), (4, 64, 9, 36)])
def test_global_attention_kernels(algo, H, hd, Tq, Tk):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(H + hd + Tq + Tk)
    B, C = 3, H * hd
    q = torch.randn(B, Tq, C, generator=gen) * 2.0          # spread-out scores
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    lens = torch.tensor([Tk, max(1, Tk // 3), 1])
    mask = torch.arange(Tk)[None] < lens[:, None]
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    got = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, algo=algo)
    close(got, want, 3e-5)
    # no mask at all
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                            torch.ones(B, 1, Tk, dtype=torch.bool), H).transpose(1, 2)
    close(ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), None, H, algo=algo), want, 3e-5)


@pytest.mark.parametrize("H,hd,Tq,Tk", [(8, 32, 10, 9), (4, 32, 9, 7), (4, 64, 5, 3)])
def test_small_lds_attention_odd_key_counts(H, hd, Tq, Tk):
    """attn_small_lds_kernel (algo 1, Tq <= 16): K is staged at pitch hd + 1, V behind it as float4 rows -- with an odd number
    of keys (tight padding runs the predictor's cross-attention at Tk = t2 / down, any integer) the V region must still
    start on a 16-byte boundary.  Key masks on, one sequence with a single valid key."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(H * hd + Tq * Tk)
    B, C = 3, H * hd
    q = torch.randn(B, Tq, C, generator=gen) * 2.0
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    lens = torch.tensor([Tk, max(1, Tk // 2), 1])
    mask = torch.arange(Tk)[None] < lens[:, None]
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    close(ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, algo=1), want, 3e-5)


def test_flash_attention_online_softmax_rescale():
    """A late key tile that dominates every earlier one forces the running-max rescale branch."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(77)
    B, H, hd, T = 2, 4, 128, 160
    q = torch.randn(B, T, H * hd, generator=gen)
    k = torch.randn(B, T, H * hd, generator=gen)
    v = torch.randn(B, T, H * hd, generator=gen)
    k[:, 130] = q[:, 5] * 3.0            # key 130 (5th tile) aligned with query 5
    k[:, 70, :hd] = q[:, 40, :hd] * 2.0
    mask = torch.ones(B, T, dtype=torch.bool)
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    close(ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), mask.to(DEV), H, algo=2), want, 3e-5)


def test_maxpool_mask_kernel():
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(3, 16, 512, generator=gen)
    lens = torch.tensor([16, 9, 1])
    mask = torch.arange(16)[None] < lens[:, None]
    y, m = ops.maxpool_mask(x.to(DEV), mask.to(DEV))
    want = torch.nn.functional.max_pool1d(x.transpose(1, 2), 3, 2, 1).transpose(1, 2) * mask[:, ::2, None]
    close(y, want, 0)
    assert torch.equal(m.cpu(), mask[:, ::2])


@pytest.mark.parametrize("Q,T", [(9, 96), (10, 512), (9, 50)])
def test_mask_head_kernel(Q, T):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(Q + T)
    emb, feat = torch.randn(3, Q, 256, generator=gen), torch.randn(3, T, 256, generator=gen)
    lens = torch.tensor([T, T // 2, 3])
    mask = torch.arange(T)[None] < lens[:, None]
    want = torch.einsum("bqc,btc->bqt", emb, feat).masked_fill(~mask[:, None], -10.0)
    close(ops.mask_head(emb.to(DEV), feat.to(DEV), mask.to(DEV)), want, 5e-5)


def test_layout_round_trip():
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(3, 77, 50, generator=gen)
    cl = ops.to_channels_last(x.to(DEV))
    assert torch.equal(cl.cpu(), x.transpose(1, 2))
    assert torch.equal(ops.btc_to_bct(cl).cpu(), x)
    slab = torch.zeros(3, 50, 20, device=DEV)
    ops.bct_to_btc(x.to(DEV), 13, 9, slab[..., 4:13])
    assert torch.equal(slab[..., 4:13].cpu(), x[:, 13:22].transpose(1, 2))
    assert float(slab[..., :4].abs().sum() + slab[..., 13:].abs().sum()) == 0.0


@pytest.mark.parametrize("K1,topk", [(133, 8), (51, 6), (51, 1)])
def test_postprocess_kernel(K1, topk):
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(K1 + topk)
    P, Q, T = 7, 9, 96
    logits = torch.randn(P, Q, K1, generator=gen) * 2
    masks = torch.randn(P, Q, T, generator=gen) * 3
    masks[0, 0] = -5.0                   # an empty mask
    masks[1, 1, 10:] = -4.0
    valid = torch.tensor([96, 50, 2, 96, 1, 33, 96], dtype=torch.int32)
    ts, tc, sf, sl = (t.cpu() for t in ops.postprocess(logits.to(DEV), masks.to(DEV), valid.to(DEV), topk))
    probs = torch.softmax(logits, -1)
    ws, wc = torch.topk(probs[..., 1:], topk, dim=-1)
    np.testing.assert_allclose(ts.numpy(), ws.numpy(), atol=1e-6)
    assert torch.equal(tc.long(), wc + 1)
    for p in range(P):
        for q in range(Q):
            on = torch.nonzero(torch.sigmoid(masks[p, q, :valid[p]]) > 0.5).flatten()
            if on.numel() == 0:
                assert sf[p, q] == -1 and sl[p, q] == -1
            else:
                assert sf[p, q] == on.min() and sl[p, q] == on.max()


def test_bad_arguments_raise():
    from vrdone_amd import ops
    x = torch.randn(2, 8, 100, device=DEV)
    g1 = torch.ones(1, 100, 1, device=DEV)
    with pytest.raises(RuntimeError, match="vrd_layernorm"):
        ops.layernorm(x, g1, g1)
    with pytest.raises(RuntimeError, match="HIP tensors"):
        ops.layernorm(x.cpu(), g1, g1)


@pytest.mark.parametrize("M,N,Cin,taps,T", [(1000, 512, 512, 1, 1000), (96 * 3, 133, 256, 1, 96),
                                            (144 * 2, 512, 1024, 3, 144), (130, 2048, 512, 1, 130)])
def test_gemm_split_bf16_precision(M, N, Cin, taps, T, precision):
    if precision != "f32":
        pytest.skip("compares both modes itself")
    """bf16x3 mode: a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, f32 accumulate.  Against an fp64 reference the
    error must be ~2^-16 relative to sum|a*w| (far below plain bf16's 2^-8), and not much worse than f32."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(M + N)
    B = M // T
    x = torch.randn(B, T, Cin, generator=gen)
    w = torch.randn(N, Cin, taps, generator=gen) / (Cin * taps) ** 0.5
    bias = torch.randn(N, generator=gen)
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), padding=taps // 2).transpose(1, 2)
    mag = torch.nn.functional.conv1d(x.abs().double().transpose(1, 2), w.abs().double(), None, padding=taps // 2).transpose(1, 2)
    try:
        ops.set_precision("bf16x3")
        got = ops.conv_gemm(x.to(DEV), w.to(DEV), bias.to(DEV)).cpu().double()
        ops.set_precision("f16x3")
        got16 = ops.conv_gemm(x.to(DEV), w.to(DEV), bias.to(DEV)).cpu().double()
    finally:
        ops.set_precision("f32")
    f32 = ops.conv_gemm(x.to(DEV), w.to(DEV), bias.to(DEV)).cpu().double()
    rel = float(((got - ref).abs() / mag).max())
    rel16 = float(((got16 - ref).abs() / mag).max())
    rel32 = float(((f32 - ref).abs() / mag).max())
    print(f"max rel err vs fp64: bf16x3 {rel:.2e}, f16x3 {rel16:.2e}, f32 {rel32:.2e}; max abs bf16x3 {float((got - ref).abs().max()):.2e}")
    assert rel < 3e-6, (rel, rel32)          # 2^-18 ~ 3.8e-6 per product before averaging
    assert rel32 < 3e-7
    # f16x3 (scaled f16 planes: 2^-22 per product before averaging): the reference-grade mode is held to the f32 kernel's bound
    assert rel16 < 3e-7, (rel16, rel32)
    close(got.float(), ref.float(), 1e-4)
    close(got16.float(), ref.float(), 1e-5)


def test_gemm_that_does_not_qualify_for_a_split_kernel_is_exact_f32(precision):
    """K % 32 == 0 (so the call carries the weight's split operand and, in the f16 format, its scale) but rows that are not
    16-byte aligned: vrd_gemm falls back to the exact-f32 kernel, which must ignore the split operand's accumulator factor."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(5)
    big = torch.randn(2, 40, 66, generator=gen).to(DEV)
    x = big[..., 1:65]                                       # 64 channels, rows start 4 bytes off a 16-byte boundary
    w, b = torch.randn(96, 64, 1, generator=gen) * 0.02, torch.randn(96, generator=gen)
    got = ops.conv_gemm(x, w.to(DEV), b.to(DEV))
    want = torch.nn.functional.conv1d(x.cpu().double().transpose(1, 2), w.double(), b.double()).transpose(1, 2)
    close(got, want.float(), 2e-6)


def test_pair_row_format_round_trip(precision):
    """Producers' pair rows decode to the f32 value within 2^-16 relative; a GEMM fed with pair rows equals the
    GEMM fed with the f32 tensor (same split, done by the producer instead of the GEMM's staging)."""
    if precision not in SPLIT:
        pytest.skip("pair rows exist in the split-precision modes only")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(123)
    x = torch.randn(3, 40, 512, generator=gen) * 3
    gam, bet = torch.randn(1, 512, 1, generator=gen), torch.randn(1, 512, 1, generator=gen)
    plain = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), relu=True)
    pr = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), relu=True, pair=True)
    assert isinstance(pr, ops.Pair) and pr.width == 512
    dec = pr.float()
    assert float(((dec - plain).abs() / plain.abs().clamp_min(1e-3)).max()) < 2 ** -15
    w = torch.randn(256, 512, 1, generator=gen) / 512 ** 0.5
    b = torch.randn(256, generator=gen)
    a = ops.conv_gemm(plain, w.to(DEV), b.to(DEV))
    c = ops.conv_gemm(pr, w.to(DEV), b.to(DEV))
    assert torch.equal(a, c)
    # k = 3 conv over pair rows, two slabs of a concatenation buffer, pair output
    cat = torch.zeros(3, 40, 1024, device=DEV)
    ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out=cat[..., :512], pair=True)
    ops.layernorm((x * 0.5 + 1).to(DEV), gam.to(DEV), bet.to(DEV), out=cat[..., 512:], pair=True)
    cat_f = torch.cat([ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV)),
                       ops.layernorm((x * 0.5 + 1).to(DEV), gam.to(DEV), bet.to(DEV))], dim=-1)
    w3 = torch.randn(512, 1024, 3, generator=gen) / 3072 ** 0.5
    want = ops.conv_gemm(cat_f, w3.to(DEV), None, act=ops.ACT_GELU)
    got = ops.conv_gemm(ops.Pair(cat, 512), w3.to(DEV), None, act=ops.ACT_GELU, out_pair=True)
    assert torch.equal(ops.conv_gemm(ops.Pair(cat, 512), w3.to(DEV), None, act=ops.ACT_GELU), want)
    assert float(((got.float() - want).abs() / want.abs().clamp_min(1e-2)).max()) < 2 ** -15


@pytest.mark.parametrize("k", [1, 3])
def test_gemm_large_tile_kernels(k, precision):
    """Shapes with enough tiles for the 256 x 256 (and, in the smaller call, the 128 x 256) LDS-DMA kernels:
    pair-row input, k = 1 / 3, mask * scale + two residuals, f32 and pair output, checked per sequence against
    an f64 reference on a sample of the sequences (rows of other sequences never enter a sequence's result)."""
    if precision not in SPLIT:
        pytest.skip("the LDS-DMA kernels are split-precision kernels")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(7 + k)
    T, Cin = 288, 512
    # (B, N): 576 tiles of 256x256 -> 256x256 kernel; M = 72,000 leaves a 64-row last tile; N = 320 a 64-column one;
    # 432 tiles of 128x256 -> 128x256 kernel
    for B, N in ((256, 512), (250, 512), (256, 320), (96, 512)):
        x = torch.randn(B, T, Cin, generator=gen)
        w = torch.randn(N, Cin, k, generator=gen) / (Cin * k) ** 0.5
        bias, scale = torch.randn(N, generator=gen), torch.rand(N, generator=gen) + 0.5
        mask = torch.rand(B, T, generator=gen) > 0.2
        res, res2 = torch.randn(B, T, N, generator=gen), torch.randn(B, T, N, generator=gen)
        xp = _to_pair(x.to(DEV))
        kw = dict(row_mask=mask.to(DEV), scale=scale.to(DEV), res=res.to(DEV), res_masked=True, res2=res2.to(DEV))
        got = ops.conv_gemm(xp, w.to(DEV), bias.to(DEV), **kw)
        got_pair = ops.conv_gemm(xp, w.to(DEV), bias.to(DEV), out_pair=True, **kw)
        sample = [0, 1, B // 2, B - 2, B - 1]
        xs = x[sample].double().transpose(1, 2)                                   # (S, Cin, T)
        y = torch.nn.functional.conv1d(xs, w.double(), bias.double(), padding=k // 2).transpose(1, 2)
        mk = mask[sample].double()[..., None]
        want = (y * mk * scale.double() + res[sample].double() * mk + res2[sample].double()).float()
        close(got[sample], want, 2e-5, gemm=True)
        # the pair output is the same result rounded to hi + lo (16 mantissa bits)
        assert float(((got_pair.float() - got).abs() / got.abs().clamp_min(1e-3)).max()) < 2 ** -15


def test_gemm_k3_short_sequences(precision):
    """k = 3 over many short sequences (T = 24 < 32): enough tiles for the LDS-DMA kernels; the 256 x 256 kernel
    must decline (it advances the position inside the sequence by 8 rows per DMA piece, which needs T >= 32)."""
    if precision not in SPLIT:
        pytest.skip("the LDS-DMA kernels are split-precision kernels")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(99)
    B, T, Cin, N = 4096, 24, 512, 512
    x = torch.randn(B, T, Cin, generator=gen)
    w = torch.randn(N, Cin, 3, generator=gen) / (Cin * 3) ** 0.5
    bias = torch.randn(N, generator=gen)
    got = ops.conv_gemm(_to_pair(x.to(DEV)), w.to(DEV), bias.to(DEV))
    sample = [0, 1, 2047, 4094, 4095]
    want = torch.nn.functional.conv1d(x[sample].double().transpose(1, 2), w.double(), bias.double(), padding=1).transpose(1, 2).float()
    close(got[sample], want, 2e-5, gemm=True)


def _to_pair(t):
    """Encode an f32 (B, T, C) tensor as pair rows (test helper; mirrors vrd::store_pair4)."""
    from vrdone_amd import _hip, ops
    f16 = ops.pair_fmt() == _hip.PAIR_F16          # the current mode's element format: f16 planes hold x * 2^F16_ACT_EXP
    dt = torch.float16 if f16 else torch.bfloat16
    t = t * 2.0 ** _hip.F16_ACT_EXP if f16 else t
    hi = t.to(dt)
    lo = (t - hi.float()).to(dt)
    C = t.shape[-1]
    raw = torch.stack([hi.reshape(*t.shape[:-1], C // 32, 32), lo.reshape(*t.shape[:-1], C // 32, 32)], dim=-2)
    raw = raw.reshape(*t.shape[:-1], 2 * C).contiguous()            # blocks of [32 hi | 32 lo]: 4C bytes per row
    return ops.Pair(raw.view(torch.float32), C)


@pytest.mark.parametrize("H,hd,Tq,Tk", [(4, 128, 96, 96), (8, 64, 144, 144), (4, 128, 288, 288), (8, 64, 512, 512),
                                        (4, 128, 40, 77)])
def test_flash_attention_pair_rows(H, hd, Tq, Tk, precision):
    if precision not in SPLIT:
        pytest.skip("pair rows exist in the split-precision modes only")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(H + hd + Tq + Tk)
    B, C = 3, H * hd
    q = torch.randn(B, Tq, C, generator=gen) * 2.0
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    k[:, min(70, Tk - 1), :hd] = q[:, 5, :hd] * 2.0          # a dominant late key: forces the running-max rescale
    lens = torch.tensor([Tk, max(1, Tk // 3), 1])
    mask = torch.arange(Tk)[None] < lens[:, None]
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    qp, kp, vp = (_to_pair(t.to(DEV)) for t in (q, k, v))
    got = ops.attention(qp, kp, vp, mask.to(DEV), H)
    close(got, want, 2e-4)
    got_pair = ops.attention(qp, kp, vp, mask.to(DEV), H, pair=True)
    assert isinstance(got_pair, ops.Pair)
    close(got_pair.float(), want, 2e-4)
    close(ops.attention(qp, kp, vp, None, H),
          O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                           torch.ones(B, 1, Tk, dtype=torch.bool), H).transpose(1, 2), 2e-4)
    # q_mask: rows of a 32-query tile that holds a valid query are bit-identical to the call without it; tiles (and
    # whole workgroups) of padding read 0
    qlens = torch.tensor([Tq, min(Tq, 33), 1])
    qm = (torch.arange(Tq)[None] < qlens[:, None])
    masked = ops.attention(qp, kp, vp, mask.to(DEV), H, q_mask=qm.to(DEV))
    tile_live = torch.nn.functional.pad(qm, (0, (-Tq) % 32)).reshape(B, -1, 32).any(-1).repeat_interleave(32, dim=1)[:, :Tq]
    assert torch.equal(masked[tile_live.to(DEV)], got[tile_live.to(DEV)])
    assert bool((masked[~tile_live.to(DEV)] == 0).all())


@pytest.mark.parametrize("H,hd,Tq,Tk", [(4, 128, 288, 288), (4, 128, 40, 77), (8, 64, 144, 144), (8, 64, 512, 512), (4, 128, 600, 330)])
def test_flash_attention_one_wave_per_simd_kernel(H, hd, Tq, Tk, precision, monkeypatch):
    """The 64-queries-per-wave kernel (vrd_attn_x3.hip, attn_flash_x3_w64_kernel) forced on every shape -- partial last key
    tile, several 256-query blocks, blocks of padding only, a dominant late key (the deferred rescale) -- against the oracle
    and against the 32-queries-per-wave kernel."""
    if precision not in SPLIT:
        pytest.skip("pair rows exist in the split-precision modes only")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(3 * H + hd + Tq + Tk)
    B, C = 3, H * hd
    q = torch.randn(B, Tq, C, generator=gen) * 2.0
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    k[:, min(70, Tk - 1), :hd] = q[:, 5, :hd] * 2.0          # a dominant late key: the reference point of the exponentials moves
    k[:, Tk - 1, hd:2 * hd] = q[:, 33, hd:2 * hd] * 3.0
    lens = torch.tensor([Tk, max(1, Tk // 3), 1])
    mask = torch.arange(Tk)[None] < lens[:, None]
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    qp, kp, vp = (_to_pair(t.to(DEV)) for t in (q, k, v))
    qlens = torch.tensor([Tq, min(Tq, 33), 1])
    qm = (torch.arange(Tq)[None] < qlens[:, None])
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("VRD_FLASH_W64", flag)
        got = ops.attention(qp, kp, vp, mask.to(DEV), H)
        close(got, want, 2e-4)
        close(ops.attention(qp, kp, vp, mask.to(DEV), H, pair=True).float(), want, 2e-4)
        masked = ops.attention(qp, kp, vp, mask.to(DEV), H, q_mask=qm.to(DEV))
        tile_live = torch.nn.functional.pad(qm, (0, (-Tq) % 32)).reshape(B, -1, 32).any(-1).repeat_interleave(32, dim=1)[:, :Tq]
        assert torch.equal(masked[tile_live.to(DEV)], got[tile_live.to(DEV)])
        assert bool((masked[~tile_live.to(DEV)] == 0).all())
        out[flag] = got
    close(out["1"], out["0"], 5e-5)
    # persistent workgroups that walk several (sequence, head) items each -- the next item's masks are staged under the current
    # item's tiles -- give the same bits as one item per workgroup, with and without the staging
    for grid, pf, walk in (("2", "1", "1"), ("5", "1", "1"), ("2", "0", "1"), ("5", "1", "0")):
        monkeypatch.setenv("VRD_FLASH_GRID", grid)
        monkeypatch.setenv("VRD_FLASH_PREFETCH", pf)
        monkeypatch.setenv("VRD_FLASH_WALK", walk)
        assert torch.equal(ops.attention(qp, kp, vp, mask.to(DEV), H), out["1"])
        masked2 = ops.attention(qp, kp, vp, mask.to(DEV), H, q_mask=qm.to(DEV))
        assert torch.equal(masked2, masked)


@pytest.mark.parametrize("Tk", [288, 2304])
def test_flash_attention_key_masks_with_holes(Tk, precision, monkeypatch):
    """Key masks that are not prefixes: leading tiles with no valid key (a row's reference point stays at its floor until the
    first valid key), a hole of whole tiles and partial tiles in the middle; at 2304 keys (more than 64 key tiles) the
    one-wave-per-SIMD kernel visits every tile, the fully masked ones included.  Both flash kernels against the oracle."""
    if precision not in SPLIT:
        pytest.skip("pair rows exist in the split-precision modes only")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(Tk)
    B, H, hd, Tq = 2, 2, 128, 256
    C = H * hd
    q = torch.randn(B, Tq, C, generator=gen) * 2.0
    k = torch.randn(B, Tk, C, generator=gen)
    v = torch.randn(B, Tk, C, generator=gen)
    mask = torch.ones(B, Tk, dtype=torch.bool)
    mask[0, :70] = False                       # two leading tiles without a valid key, then a partial one
    mask[0, 130:197] = False                   # a hole: partial, whole, partial tile
    mask[1, 5:Tk - 3] = False                  # almost everything masked
    want = O.full_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask[:, None], H).transpose(1, 2)
    qp, kp, vp = (_to_pair(t.to(DEV)) for t in (q, k, v))
    for flag in ("0", "1"):
        monkeypatch.setenv("VRD_FLASH_W64", flag)
        close(ops.attention(qp, kp, vp, mask.to(DEV), H), want, 2e-4)


def test_row_blocks_padding_map():
    """vrd_row_blocks: the 32-row blocks dealt into segments (about eight, whole 256-row tiles, the last one shorter);
    inside a segment the blocks holding a valid row first (ascending), the fully padded ones after (ascending); every
    block exactly once; each segment gets its proportional share of the valid blocks."""
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(3)
    for B, T in ((8, 288), (2048, 288), (2070, 288), (64, 32), (1000, 96), (17, 288), (1, 32)):
        lens = torch.randint(0, T + 1, (B,), generator=gen)
        mask = (torch.arange(T)[None, :] < lens[:, None]).to(DEV)
        order, count, seg_len = ops.row_blocks(mask)
        flags = mask.reshape(-1, 32).any(1).cpu()
        nblk = len(flags)
        assert seg_len % 8 == 0 and seg_len == 8 * (((nblk + 7) // 8 + 7) // 8)
        S = (nblk + seg_len - 1) // seg_len
        assert count.numel() == S <= 8
        act, pad = torch.nonzero(flags).flatten(), torch.nonzero(~flags).flatten()
        n = len(act)
        start = [min(s * seg_len, nblk) for s in range(S + 1)]
        first = [n * start[s] // nblk for s in range(S + 1)]
        assert count.cpu().tolist() == [first[s + 1] - first[s] for s in range(S)]
        want, p0 = [], 0
        for s in range(S):
            n_pad = (start[s + 1] - start[s]) - (first[s + 1] - first[s])
            assert n_pad >= 0
            want += [act[first[s]:first[s + 1]], pad[p0:p0 + n_pad]]
            p0 += n_pad
        assert torch.equal(order.cpu().long(), torch.cat(want))
        assert ops.row_blocks(mask) is ops.row_blocks(mask)            # cached on the mask object
    assert ops.row_blocks(torch.ones(3, 40, dtype=torch.bool, device=DEV)) is None       # 120 rows: not whole blocks


@pytest.mark.parametrize("k", [1, 3])
def test_gemm_padding_skip_is_exact(k, precision):
    """The 256 x 256 kernel with a padding map: blocks without a valid row skip the contraction.  With row_mask the
    result is bit-identical to the full computation (masked rows are res*mask + res2 either way); without it the
    rows that hold valid frames are bit-identical and the skipped ones finite."""
    if precision not in SPLIT:
        pytest.skip("the LDS-DMA kernels are split-precision kernels")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(21 + k)
    B, T, Cin, N = 310, 288, 512, 512         # 2790 blocks: segments of 352, the last one 326; a partial last tile
    lens = torch.randint(1, T + 1, (B,), generator=gen)
    lens[:3] = torch.tensor([288, 256, 1])
    mask = (torch.arange(T)[None, :] < lens[:, None]).to(DEV)
    assert int(ops.row_blocks(mask)[1].sum()) < B * T // 32 * 0.7          # plenty of fully padded blocks
    x = torch.randn(B, T, Cin, generator=gen).to(DEV)             # padded rows hold data too (e.g. LayerNorm's beta)
    w = (torch.randn(N, Cin, k, generator=gen) / (Cin * k) ** 0.5).to(DEV)
    bias, scale = torch.randn(N, generator=gen).to(DEV), (torch.rand(N, generator=gen) + 0.5).to(DEV)
    res, res2 = torch.randn(B, T, N, generator=gen).to(DEV), torch.randn(B, T, N, generator=gen).to(DEV)
    xp = _to_pair(x)
    cases = [dict(row_mask=mask), dict(row_mask=mask, scale=scale, res=res, res_masked=True, res2=res2),
             dict(row_mask=mask, res=res), dict(row_mask=mask, out_pair=True)]
    if k == 1:
        cases += [dict(skip_rows=mask), dict(skip_rows=mask, act=ops.ACT_GELU, out_pair=True)]
    for kw in cases:
        old = ops._skip_padding
        try:
            ops._skip_padding = True
            got = ops.conv_gemm(xp, w, bias, **kw)
            ops._skip_padding = False
            want = ops.conv_gemm(xp, w, bias, **kw)
        finally:
            ops._skip_padding = old
        # decoded values: a masked row is (acc + bias) * 0, whose zero carries the sign of acc + bias resp. bias
        g, wnt = (got.float(), want.float()) if isinstance(got, ops.Pair) else (got, want)
        if "row_mask" in kw:
            assert torch.equal(g, wnt), kw.keys()
        else:
            valid = mask.reshape(-1, 32).any(1).repeat_interleave(32).reshape(B, T)
            assert torch.equal(g[valid], wnt[valid]) and bool(torch.isfinite(g).all())
            assert not torch.equal(g, wnt)                       # the skip really happened


def test_gemm_batch_equals_single_calls(precision):
    """vrd_gemm_batch: three GEMMs that differ only in input / weight / bias / output as one launch of the 256 x 256
    kernel (with a padding map), and a mixed list that has to fall back to one launch per problem: same bits as
    calling conv_gemm on each."""
    if precision not in SPLIT:
        pytest.skip("the 256 x 256 kernel is a split-precision kernel")
    from vrdone_amd import ops
    gen = torch.Generator().manual_seed(5)
    B, T, Cin, N = 300, 288, 512, 512
    lens = torch.randint(1, T + 1, (B,), generator=gen)
    mask = (torch.arange(T)[None, :] < lens[:, None]).to(DEV)
    xs = [_to_pair(torch.randn(B, T, Cin, generator=gen).to(DEV)) for _ in range(3)]
    ws = [(torch.randn(N, Cin, 1, generator=gen) / Cin ** 0.5).to(DEV) for _ in range(3)]
    bs = [torch.randn(N, generator=gen).to(DEV) for _ in range(3)]
    for kw in (dict(out_pair=True, skip_rows=mask), dict(row_mask=mask), dict(act=ops.ACT_GELU, out_pair=True)):
        got = ops.conv_gemm_batch([((x, w, b), dict(kw)) for x, w, b in zip(xs, ws, bs)])
        for g, x, w, b in zip(got, xs, ws, bs):
            want = ops.conv_gemm(x, w, b, **kw)
            assert torch.equal(g.t if isinstance(g, ops.Pair) else g, want.t if isinstance(want, ops.Pair) else want)
    # different epilogues / shapes in one list: one launch each
    w_small = (torch.randn(256, Cin, 1, generator=gen) / Cin ** 0.5).to(DEV)
    got = ops.conv_gemm_batch([((xs[0], ws[0], bs[0]), dict(out_pair=True)), ((xs[1], w_small, None), dict(row_mask=mask))])
    assert torch.equal(got[0].t, ops.conv_gemm(xs[0], ws[0], bs[0], out_pair=True).t)
    assert torch.equal(got[1], ops.conv_gemm(xs[1], w_small, None, row_mask=mask))

@pytest.mark.parametrize("Cin,N,ln,pair", [(8, 512, True, True), (8, 512, True, False), (5, 512, False, True), (5, 512, False, False),
                                           (7, 256, True, False), (10, 256, False, False)])
def test_few_channel_conv_layernorm_row_kernel(Cin, N, ln, pair, precision):
    """vrd_conv_ln (the box-feature embeddings: k = 3 conv with 5 / 8 input channels * mask -> [LayerNorm -> ReLU]) against the GEMM +
    LayerNorm launches it replaces and against the oracle's Conv1d: sequence ends (zero padding), masked rows, pair-row output."""
    from vrdone_amd import ops
    if pair and precision not in SPLIT:
        pytest.skip("pair rows exist in the split-precision modes")
    g = torch.Generator().manual_seed(Cin * 10 + N)
    B, T = 37, 53
    lens = torch.randint(1, T + 1, (B,), generator=g)
    mask = (torch.arange(T)[None, :] < lens[:, None])
    x = torch.randn(B, T, Cin, generator=g) * 3
    w = torch.randn(N, Cin, 3, generator=g) / (3 * Cin) ** 0.5
    b = torch.randn(N, generator=g) * 0.3
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    xd, wd, bd, md = x.to(DEV), w.to(DEV), b.to(DEV), mask.to(DEV)
    assert ops.conv_ln_ok(xd, wd, bd)
    got = ops.conv_ln(xd, wd, bd, row_mask=md, gamma=gamma.to(DEV) if ln else None, beta=beta.to(DEV) if ln else None, relu=ln, pair=pair)
    want = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), b.double(), padding=1).transpose(1, 2) * mask[..., None]
    if ln:
        mu = want.mean(-1, keepdim=True)
        var = ((want - mu) ** 2).mean(-1, keepdim=True)
        want = torch.relu((want - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
    gotf = (got.float() if isinstance(got, ops.Pair) else got).cpu().double()
    tol = 2e-5 if not pair else (2e-4 if precision == "bf16x3" else 2e-5)
    assert float((gotf - want).abs().max()) <= tol * max(1.0, float(want.abs().max()))
    # ... and the launches it replaces (exact-f32 GEMM, then the LayerNorm kernel)
    h = ops.conv_gemm(xd, wd, bd, row_mask=md)
    if ln:
        h = ops.layernorm(h, gamma.to(DEV), beta.to(DEV), relu=True, pair=pair)
    elif pair:
        h = ops.conv_gemm(xd, wd, bd, row_mask=md, out_pair=True)
    hf = (h.float() if isinstance(h, ops.Pair) else h).cpu().double()
    assert float((gotf - hf).abs().max()) <= tol * max(1.0, float(want.abs().max()))
