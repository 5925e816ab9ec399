"""Backward kernels (csrc/vrd_backward.hip) and the autograd layer (vrdone_amd/autograd.py) on a real MI355X.

Every differentiable op is compared -- outputs and the gradients of every input and parameter -- with torch.autograd
through the oracle's functional restatement of the same reference code (oracle/vrd_oracle.py), evaluated on the CPU in
float64.  Tolerances are relative to the largest gradient entry of the tensor: 2e-5 in f32 mode, 2e-4 in bf16x3 mode
(split-bf16 products in the GEMMs of the forward and of the input gradients; weight gradients use exact f32 MFMA
products in both modes)."""
import numpy as np
import pytest
import torch

from oracle import vrd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _autograd_on():
    """Other GPU test modules switch autograd off globally at import; these tests need it recording."""
    with torch.enable_grad():
        yield


@pytest.fixture(params=["f32", "bf16x3", "f16x3"])
def precision(request):
    from vrdone_amd import ops
    old = ops.get_precision()
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(old)


def tol(precision):
    return 2e-5 if precision == "f32" else 2e-4


def rel_close(got, want, rtol, what=""):
    got = got.detach().double().cpu()
    want = want.detach().double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert bool(torch.isfinite(got).all()), what
    scale = float(want.abs().max()) + 1e-12
    err = float((got - want).abs().max()) / scale
    assert err <= rtol, f"{what}: max error {err:.3e} of the largest entry (tolerance {rtol:.1e})"


def leaf(t, grad=True):
    return t.clone().to(DEV).requires_grad_(grad)


def ref64(t):
    return t.clone().double().requires_grad_(True)


def cl(x):          # (B, C, T) -> (B, T, C)
    return x.transpose(1, 2).contiguous()


def mask_for(B, T, lens):
    return (torch.arange(T)[None] < torch.tensor(lens)[:, None])


# --------------------------------------------------------------------------------------------------------- dense conv
@pytest.mark.parametrize("k,Cin,N", [(1, 512, 512), (3, 64, 96), (3, 8, 512), (1, 256, 133)])
def test_linear_backward(k, Cin, N, precision):
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(k * 100 + Cin)
    B, T = 3, 48
    m = mask_for(B, T, [48, 31, 5])
    x = torch.randn(B, Cin, T, generator=g) * m[:, None]
    w = torch.randn(N, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    dy = torch.randn(B, N, T, generator=g)
    xr, wr, br = ref64(x), ref64(w), ref64(b)
    yr, _ = O.masked_conv1d(xr, m[:, None], wr, br)
    yr.backward(dy.double())
    xd, wd, bd = leaf(cl(x)), leaf(w), leaf(b)
    with torch.enable_grad():
        y = ops.conv_gemm(xd, wd, bd, row_mask=m.to(DEV))
    y.backward(cl(dy).to(DEV))
    rel_close(y, cl(yr), tol(precision), "y")
    rel_close(xd.grad, cl(xr.grad), tol(precision), "dx")
    rel_close(wd.grad, wr.grad, tol(precision), "dW")
    rel_close(bd.grad, br.grad, tol(precision), "db")


@pytest.mark.parametrize("k,Cin,N", [(1, 64, 130), (3, 8, 512)])          # shapes the fused input-gradient GEMM does not take
def test_linear_backward_of_an_f32_forward_stays_exact_in_another_mode(k, Cin, N):
    """A step the range guard repeats in f32 (MaskVRD.forward_training) is differentiated after the f32 block has been left:
    the unfused input-gradient GEMM must still form exact f32 products -- with the mode's fixed-scale f16 planes a gradient of
    ~1e-6 would land in f16 subnormals and nothing would flag it."""
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(77 + k)
    B, T = 3, 40
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(N, Cin, k, generator=g) / (Cin * k) ** 0.5
    dy = torch.randn(B, T, N, generator=g) * 1e-6
    xd, wd = leaf(x), leaf(w)
    with ops.use_precision("f32"), torch.enable_grad():
        y = ops.conv_gemm(xd, wd, None)
    with ops.use_precision("f16x3"):
        y.backward(dy.to(DEV))
    xr, wr = ref64(x.transpose(1, 2).contiguous()), ref64(w)
    yr = torch.nn.functional.conv1d(xr, wr, padding=k // 2)
    yr.backward(dy.transpose(1, 2).double())
    rel_close(xd.grad, cl(xr.grad), 2e-5, "dx")
    rel_close(wd.grad, wr.grad, 2e-5, "dW")


@pytest.mark.parametrize("k,Cin,N,B,T", [(3, 96, 160, 7, 100), (1, 64, 64, 40, 90), (3, 21, 132, 7, 100), (1, 130, 66, 9, 64),
                                         (3, 64, 128, 40, 9), (3, 32, 64, 100, 3)])      # (sequences shorter than a 32-row step)
def test_linear_weight_gradient_over_many_row_chunks(k, Cin, N, B, T, precision):
    """dW when the rows span several waves and blocks (row chunks end inside sequences; T is not a multiple of the 16 rows of a
    step; masked tails), both wgrad kernels (exact f32 / split precision) by mode."""
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(B * T + k)
    lens = torch.randint(1, T + 1, (B,), generator=g).tolist()
    lens[0] = T
    m = mask_for(B, T, lens)
    x = torch.randn(B, Cin, T, generator=g) * m[:, None]
    w = torch.randn(N, Cin, k, generator=g) / (Cin * k) ** 0.5
    dy = torch.randn(B, N, T, generator=g)
    xr, wr = ref64(x), ref64(w)
    yr, _ = O.masked_conv1d(xr, m[:, None], wr, None)
    yr.backward(dy.double())
    xd, wd = leaf(cl(x)), leaf(w)
    with torch.enable_grad():
        y = ops.conv_gemm(xd, wd, None, row_mask=m.to(DEV))
    y.backward(cl(dy).to(DEV))
    rel_close(wd.grad, wr.grad, tol(precision), "dW")
    rel_close(xd.grad, cl(xr.grad), tol(precision), "dx")


@pytest.mark.parametrize("M,N,Cin,k,T", [(24576, 512, 512, 1, 512), (3000, 133, 63, 1, 100), (4096, 96, 40, 3, 64), (640, 2048, 512, 1, 64),
                                         (16400, 384, 320, 1, 100), (24576, 256, 256, 3, 512)])      # (the last two: 256 x 256 tiles with ragged edges / k = 3)
def test_weight_gradient_through_partial_tiles(M, N, Cin, k, T):
    """vrd_gemm_wgrad_x3 with a scratch buffer (the row chunks' partial tiles stored, then summed in chunk order by a second
    launch) against the same call without one (float atomics) and against float64; with the buffer the result is the same bits
    from run to run, it accumulates into dW like the atomics do, and a buffer that is too small falls back to the atomics."""
    from vrdone_amd import _hip
    g = torch.Generator().manual_seed(M + N)
    G, X = torch.randn(M, N, generator=g), torch.randn(M, Cin, generator=g)
    mask = (torch.rand(M, generator=g) < 0.8).to(torch.uint8)
    Gd, Xd, md = G.to(DEV), X.to(DEV), mask.to(DEV)
    part = torch.full((4 * 256 * 16384 + N * k * Cin + 8,), float("nan"), device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def run(scratch, floats, g_scale=None, init=1.0):
        dW = torch.full((N, k * Cin), init, device=DEV)
        db = torch.zeros(N, device=DEV)
        _hip.check(_hip.lib.vrd_gemm_wgrad_x3(Gd.data_ptr(), N, Xd.data_ptr(), Cin, md.data_ptr(), M, N, Cin, k, T, dW.data_ptr(),
                                              db.data_ptr(), scratch, floats, g_scale, stream), "vrd_gemm_wgrad_x3")
        return dW, db

    a, ba = run(None, 0)
    b, bb = run(part.data_ptr(), part.numel())
    b2, _ = run(part.data_ptr(), part.numel())
    c, _ = run(part.data_ptr(), 1000)                       # too small: atomics
    Gm = (G * mask[:, None]).double()
    Xs = X.double().view(M // T, T, Cin)
    taps = [torch.nn.functional.pad(Xs, (0, 0, 1, 1))[:, t:t + T].reshape(M, Cin) for t in range(3)] if k == 3 else [X.double()]
    want = torch.cat([Gm.t() @ xt for xt in taps], 1) + 1.0
    rel_close(b, want, 2e-4, "dW (partial tiles)")
    rel_close(a, want, 2e-4, "dW (atomics)")
    rel_close(c, want, 2e-4, "dW (scratch too small)")
    rel_close(bb, Gm.sum(0), 2e-5, "db")
    assert torch.equal(b, b2), "the chunk-ordered sum is not reproducible"
    assert float((a - b).abs().max()) <= 1e-4 * float(want.abs().max())
    # the f16x3 mode's form: f16 planes, the gradient at the power-of-two factor of vrd_absmax_scale -- a tensor far outside the
    # f16 range as it stands (x 1e-9), the products ~2^-22 instead of the bf16 planes' ~2^-17
    if N % 4 == 0:
        tiny = 1e-9
        Gs = (Gd * tiny).contiguous()
        scale = torch.zeros(_hip.ABSMAX_SCALE_FLOATS, device=DEV)
        _hip.check(_hip.lib.vrd_absmax_scale(Gs.data_ptr(), N, M, N, scale.data_ptr(), stream), "vrd_absmax_scale")
        mx = float(Gs.abs().max())
        assert 2.0 ** 13 <= mx * float(scale[0]) < 2.0 ** 14 and float(scale[0] * scale[1]) == 1.0 and float(scale[3]) == 0.0
        keep = Gd
        Gd = Gs
        try:
            f, bf = run(part.data_ptr(), part.numel(), scale.data_ptr(), init=0.0)
            e, _ = run(part.data_ptr(), part.numel(), init=0.0)
        finally:
            Gd = keep
        want_s = (want - 1.0) * tiny
        err16 = float((f.double().cpu() - want_s).abs().max()) / float(want_s.abs().max())
        errbf = float((e.double().cpu() - want_s).abs().max()) / float(want_s.abs().max())
        print(f"dW error relative to the largest entry: f16 planes {err16:.2e}, bf16 planes {errbf:.2e}")
        assert err16 <= 2e-6 and err16 <= 0.5 * errbf + 1e-7
        rel_close(bf, Gm.sum(0) * tiny, 2e-5, "db (scaled gradient)")


@pytest.mark.parametrize("rows,cols,ldx", [(1, 4, 4), (7, 12, 12), (768, 512, 512), (24576, 2048, 2048), (100003, 512, 512),
                                           (3000, 132, 160), (4097, 8, 1024), (24576, 512, 1536)])
def test_absmax_scale_over_flat_and_padded_rows(rows, cols, ldx):
    """vrd_absmax_scale: the power-of-two factor of max |x| over the (rows, cols) window of a row-major buffer of pitch ldx --
    values outside the window (far larger) must not count; the buffer is reusable without clearing (ticket back at zero); an
    all-zero window gives e = 0."""
    from vrdone_amd import _hip
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(rows * 31 + cols)
    buf = torch.full((rows, ldx), 1e30)
    win = torch.randn(rows, cols, generator=g) * torch.exp(4 * torch.randn(rows, 1, generator=g))
    scale = torch.zeros(_hip.ABSMAX_SCALE_FLOATS, device=DEV)
    for trial, factor in enumerate([1.0, 1e-12, 3e7, 0.0]):
        w = win * factor
        if trial == 1:                                            # the maximum as the window's very last element
            w[-1, -1] = 5e-3
        buf[:, :cols] = w
        x = buf.to(DEV)
        _hip.check(_hip.lib.vrd_absmax_scale(x.data_ptr(), ldx, rows, cols, scale.data_ptr(), stream), "vrd_absmax_scale")
        s0, s1, ticket = float(scale[0]), float(scale[1]), float(scale[3])
        mx = float(w.abs().max())
        assert ticket == 0.0 and s0 * s1 == 1.0
        if mx == 0.0:
            assert s0 == 1.0
        else:
            assert 2.0 ** 13 <= mx * s0 < 2.0 ** 14, (rows, cols, ldx, trial, mx, s0)


@pytest.mark.parametrize("ks,gin,stride,C,B,T", [(3, 1, 1, 512, 64, 64), (3, 1, 1, 260, 37, 50), (3, 2, 1, 256, 40, 50), (1, 1, 1, 256, 33, 64),
                                                 (3, 1, 2, 512, 16, 96), (3, 1, 1, 512, 300, 5)])
def test_column_sums_with_and_without_scratch(ks, gin, stride, C, B, T):
    """vrd_colsum and vrd_dwconv_wgrad with a scratch buffer (four channels per lane, the row blocks' partial sums stored, then
    added up by a second launch -- the cases that qualify) against the same calls without one (one float atomic per column and
    workgroup) and against float64; ragged row counts, strided inputs, sequences shorter than a block's rows."""
    from vrdone_amd import _hip
    lib = _hip.lib
    g = torch.Generator().manual_seed(ks * 7 + gin + C)
    rows = B * T                                           # output rows; the conv input has stride * T rows per sequence
    dD = torch.randn(rows, C, generator=g)
    x = torch.randn(B * stride * T, C * gin, generator=g)
    other = torch.randn(rows, C, generator=g)
    mask = (torch.rand(rows, generator=g) < 0.8).to(torch.uint8)
    rscale = torch.rand(rows, generator=g)
    dDd, xd, od, md, rd = dD.to(DEV), x.to(DEV), other.to(DEV), mask.to(DEV), rscale.to(DEV)
    part = torch.full((1024 * 4 * C + 64,), float("nan"), device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def dww(scratch, floats):
        dw, db = torch.ones(C, gin, ks, device=DEV), torch.ones(C, device=DEV)
        _hip.check(lib.vrd_dwconv_wgrad(dDd.data_ptr(), C, xd.data_ptr(), C * gin, ks, stride, gin, T, md.data_ptr(), rows, C, dw.data_ptr(),
                                        db.data_ptr(), scratch, floats, stream), "vrd_dwconv_wgrad")
        return dw, db

    def cs(b, scratch, floats):
        out = torch.ones(C, device=DEV)
        _hip.check(lib.vrd_colsum(dDd.data_ptr(), C, b.data_ptr() if b is not None else None, C, 1, 0, 1, 0, T, md.data_ptr(), rd.data_ptr(),
                                  rows, C, out.data_ptr(), scratch, floats, stream), "vrd_colsum")
        return out

    Gm = (dD * mask[:, None]).double()
    xs = x.double().view(B, stride * T, C, gin)
    want_w = torch.zeros(C, gin, ks, dtype=torch.float64)
    Gs = Gm.view(B, T, C)
    for kk in range(ks):
        for t in range(T):
            ti = stride * t + kk - ks // 2
            if 0 <= ti < stride * T:
                want_w[:, :, kk] += torch.einsum("bc,bcg->cg", Gs[:, t], xs[:, ti])
    (wa, ba), (wb, bb) = dww(None, 0), dww(part.data_ptr(), part.numel())
    rel_close(wa, want_w + 1, 2e-5, "dw (atomics)")
    rel_close(wb, want_w + 1, 2e-5, "dw (partial sums)")
    rel_close(ba, Gm.sum(0) + 1, 2e-5, "dbias (atomics)")
    rel_close(bb, Gm.sum(0) + 1, 2e-5, "dbias (partial sums)")
    scaled = Gm * rscale.double()[:, None]
    for b, want in ((None, scaled.sum(0) + 1), (od, (scaled * other.double()).sum(0) + 1)):
        rel_close(cs(b, None, 0), want, 2e-5, "colsum (atomics)")
        rel_close(cs(b, part.data_ptr(), part.numel()), want, 2e-5, "colsum (partial sums)")
        rel_close(cs(b, part.data_ptr(), 100), want, 2e-5, "colsum (scratch too small)")


def test_conv_gemm_epilogue_backward(precision):
    """GELU + mask + AffineDropPath scale + per-sample keep factors + masked residual + second residual."""
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(5)
    B, T, Cin, N = 4, 24, 128, 256
    m = mask_for(B, T, [24, 17, 3, 9])
    x = torch.randn(B, T, Cin, generator=g)
    w = torch.randn(N, Cin, 1, generator=g) / Cin ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    scale = torch.rand(1, N, 1, generator=g) + 0.5
    res, res2 = torch.randn(B, T, N, generator=g), torch.randn(B, T, N, generator=g)
    keep = torch.tensor([1.0, 0.0, 1.0, 1.0]) / 0.9
    dy = torch.randn(B, T, N, generator=g)
    xr, wr, br, sr, r1, r2 = (ref64(t) for t in (x, w, b, scale, res, res2))
    mf = m.double()[:, :, None]
    v = torch.nn.functional.gelu(xr @ wr[:, :, 0].T + br) * mf
    yr = v * sr.view(1, 1, N) * keep.double()[:, None, None] + r1 * mf + r2
    yr.backward(dy.double())
    xd, wd, bd, sd, d1, d2 = (leaf(t) for t in (x, w, b, scale, res, res2))
    rs = keep[:, None].expand(B, T).contiguous().view(-1).to(DEV)
    with torch.enable_grad():
        y = ops.conv_gemm(xd, wd, bd, act=ops.ACT_GELU, row_mask=m.to(DEV), scale=sd, row_scale=rs, res=d1, res_masked=True, res2=d2)
    y.backward(dy.to(DEV))
    for name, a, r in (("y", y, yr), ("dx", xd.grad, xr.grad), ("dW", wd.grad, wr.grad), ("db", bd.grad, br.grad),
                       ("dscale", sd.grad, sr.grad), ("dres", d1.grad, r1.grad), ("dres2", d2.grad, r2.grad)):
        rel_close(a, r, tol(precision), name)


# ---------------------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("C,relu,post,B,T", [(512, False, False, 5, 9), (256, True, False, 5, 9), (256, False, True, 5, 9),
                                             (512, True, False, 37, 71), (256, False, False, 64, 96)])
def test_layernorm_backward(C, relu, post, B, T):
    """(the last two: enough rows for the two-step column sums -- per-workgroup partial sums, then their reduction)"""
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(C + relu)
    x = torch.randn(B, C, T, generator=g) * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(1, C, 1, generator=g)
    beta = 0.1 * torch.randn(1, C, 1, generator=g)
    pa = torch.randn(T, C, generator=g) if post else None
    dy = torch.randn(B, C, T, generator=g)
    xr, gr, br = ref64(x), ref64(gamma), ref64(beta)
    yr = O.channel_ln(xr, gr, br)
    if relu:
        yr = torch.relu(yr)
    if post:
        par = ref64(pa)
        yr = yr + par.T[None]
    yr.backward(dy.double())
    xd, gd, bd = leaf(cl(x)), leaf(gamma), leaf(beta)
    pd = leaf(pa) if post else None
    with torch.enable_grad():
        y = ops.layernorm(xd, gd, bd, relu=relu, post_add=pd)
    y.backward(cl(dy).to(DEV))
    rel_close(y, cl(yr), 2e-5, "y")
    rel_close(xd.grad, cl(xr.grad), 2e-5, "dx")
    rel_close(gd.grad, gr.grad, 2e-5, "dgamma")
    rel_close(bd.grad, br.grad, 2e-5, "dbeta")
    if post:
        rel_close(pd.grad, par.grad, 2e-5, "dpost")


# ------------------------------------------------------------------------------------------------- depthwise conv (+LN)
def _dw_ref(x, m, w, bias, stride, groups):
    return O.masked_conv1d(x, m, w, bias, stride=stride, groups=groups)[0]


@pytest.mark.parametrize("case", ["qkv_s1", "qkv_s2", "fpn_top", "fpn_up", "mask_features", "k1"])
def test_dwconv_ln_backward(case):
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(len(case))
    B, T = 3, 24
    lens = [24, 14, 5]
    stride, gin, k, n_out, C, pre, up, with_ln, with_bias = {
        "qkv_s1": (1, 1, 3, 3, 512, True, False, True, False),
        "qkv_s2": (2, 1, 3, 3, 512, True, False, True, False),
        "fpn_top": (1, 2, 3, 1, 256, False, False, True, False),
        "fpn_up": (1, 1, 3, 1, 256, False, True, True, False),
        "mask_features": (1, 1, 3, 1, 256, False, False, False, True),
        "k1": (1, 1, 1, 1, 256, False, False, True, False),
    }[case]
    Cin = C * gin
    m_in = mask_for(B, T, lens)
    m_out = m_in[:, ::stride]
    x = torch.randn(B, Cin, T, generator=g) * m_in[:, None]
    x_up = torch.randn(B, Cin, T // 2, generator=g) if up else None
    ws = [torch.randn(C, gin, k, generator=g) / (gin * k) ** 0.5 for _ in range(n_out)]
    bs = [torch.randn(C, generator=g) * 0.1 if with_bias else None for _ in range(n_out)]
    gs = [1 + 0.1 * torch.randn(1, C, 1, generator=g) for _ in range(n_out)]
    es = [0.1 * torch.randn(1, C, 1, generator=g) for _ in range(n_out)]
    pg, pb = 1 + 0.1 * torch.randn(1, Cin, 1, generator=g), 0.1 * torch.randn(1, Cin, 1, generator=g)
    dys = [torch.randn(B, C, T // stride, generator=g) for _ in range(n_out)]
    # reference (float64 autograd through the oracle's ops)
    xr = ref64(x)
    ur = ref64(x_up) if up else None
    wr, gr, er = [ref64(t) for t in ws], [ref64(t) for t in gs], [ref64(t) for t in es]
    br = [ref64(t) if t is not None else None for t in bs]
    pgr, pbr = ref64(pg), ref64(pb)
    xin = O.channel_ln(xr, pgr, pbr) if pre else xr
    if up:
        xin = xin + ur.repeat_interleave(2, dim=2)
    loss = 0
    outs_r = []
    for i in range(n_out):
        d = _dw_ref(xin, m_in[:, None], wr[i], br[i], stride, C)
        if with_ln:
            d = O.channel_ln(d, gr[i], er[i])
        outs_r.append(d)
        loss = loss + (d * dys[i].double()).sum()
    loss.backward()
    # HIP
    xd = leaf(cl(x))
    ud = leaf(cl(x_up)) if up else None
    wd, gd, ed = [leaf(t) for t in ws], [leaf(t) for t in gs], [leaf(t) for t in es]
    bd = [leaf(t) if t is not None else None for t in bs]
    pgd, pbd = leaf(pg), leaf(pb)
    sets = [dict(weight=wd[i], bias=bd[i], gamma=gd[i] if with_ln else None, beta=ed[i] if with_ln else None) for i in range(n_out)]
    with torch.enable_grad():
        outs = ops.dwconv_ln(xd, sets, mask_out=m_out.contiguous().to(DEV), stride=stride, x_up=ud, pre_ln=(pgd, pbd) if pre else None)
        total = sum((o * cl(dy).to(DEV)).sum() for o, dy in zip(outs, dys))
    total.backward()
    for i in range(n_out):
        rel_close(outs[i], cl(outs_r[i]), 2e-5, f"y{i}")
        rel_close(wd[i].grad, wr[i].grad, 2e-5, f"dw{i}")
        if with_ln:
            rel_close(gd[i].grad, gr[i].grad, 2e-5, f"dgamma{i}")
            rel_close(ed[i].grad, er[i].grad, 2e-5, f"dbeta{i}")
        if with_bias:
            rel_close(bd[i].grad, br[i].grad, 2e-5, f"dbias{i}")
    rel_close(xd.grad, cl(xr.grad), 2e-5, "dx")
    if up:
        rel_close(ud.grad, cl(ur.grad), 2e-5, "dx_up")
    if pre:
        rel_close(pgd.grad, pgr.grad, 2e-5, "dpre_gamma")
        rel_close(pbd.grad, pbr.grad, 2e-5, "dpre_beta")


# ---------------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("with_rel", [False, True])
@pytest.mark.parametrize("n_head,half_win", [(4, 3), (8, 4)])
def test_local_attention_backward(n_head, half_win, with_rel):
    """with_rel: the `use_rel_pe` bias (1, 1, n_head, window) on the scores; its gradient is the sum of dS over the rows."""
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(n_head)
    B, T, C = 3, 24, 512
    m = mask_for(B, T, [24, 13, 2])
    q, k, v, dO = (torch.randn(B, C, T, generator=g) for _ in range(4))
    rel = torch.randn(1, 1, n_head, 2 * half_win + 1, generator=g) if with_rel else None
    qr, kr, vr = ref64(q), ref64(k), ref64(v)
    relr = ref64(rel) if with_rel else None
    outr = O.banded_attention(qr, kr, vr, m[:, None], n_head, half_win, rel_pe=relr)
    outr.backward(dO.double())
    qd, kd, vd = leaf(cl(q)), leaf(cl(k)), leaf(cl(v))
    reld = leaf(rel) if with_rel else None
    with torch.enable_grad():
        out = ops.local_attention(qd, kd, vd, m.to(DEV), n_head, half_win, rel_pe=reld)
    out.backward(cl(dO).to(DEV))
    rel_close(out, cl(outr), 2e-5, "out")
    for name, a, r in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        rel_close(a, cl(r), 2e-5, name)
    if with_rel:
        rel_close(reld.grad, relr.grad, 2e-5, "d rel_pe")


@pytest.mark.parametrize("n_head,C,Tq,Tk,masked", [(4, 512, 96, 96, True), (8, 512, 40, 64, True), (4, 256, 9, 12, True), (4, 256, 9, 9, False)])
def test_global_attention_backward(n_head, C, Tq, Tk, masked):
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(Tq + Tk)
    B = 3
    km = mask_for(B, Tk, [Tk, max(Tk // 2, 1), 2]) if masked else None
    q, dO = torch.randn(B, C, Tq, generator=g), torch.randn(B, C, Tq, generator=g)
    k, v = torch.randn(B, C, Tk, generator=g), torch.randn(B, C, Tk, generator=g)
    qr, kr, vr = ref64(q), ref64(k), ref64(v)
    mk = km[:, None] if masked else torch.ones(B, 1, Tk, dtype=torch.bool)
    outr = O.full_attention(qr, kr, vr, mk, n_head)
    outr.backward(dO.double())
    qd, kd, vd = leaf(cl(q)), leaf(cl(k)), leaf(cl(v))
    with torch.enable_grad():
        out = ops.attention(qd, kd, vd, km.to(DEV) if masked else None, n_head)
    out.backward(cl(dO).to(DEV))
    rel_close(out, cl(outr), 2e-5, "out")
    for name, a, r in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        rel_close(a, cl(r), 2e-5, name)


@pytest.mark.parametrize("n_head,C,Tq,Tk,masked", [(8, 512, 96, 96, True), (8, 512, 40, 77, True), (4, 256, 130, 64, False),
                                                   (8, 512, 288, 512, True)])
def test_global_attention_backward_fused(n_head, C, Tq, Tk, masked, precision):
    """vrd_attention_bwd (head_dim 64, the split modes' backward): dq, dk, dv against float64 autograd of the oracle's attention,
    key masks with a short and a nearly empty sequence, partial last tiles on both axes, and against the five-product form."""
    from vrdone_amd import autograd, ops
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    B = 3
    km = mask_for(B, Tk, [Tk, max(Tk // 2 + 3, 1), 2]) if masked else None
    q, dO = torch.randn(B, C, Tq, generator=g), torch.randn(B, C, Tq, generator=g)
    k, v = torch.randn(B, C, Tk, generator=g), torch.randn(B, C, Tk, generator=g)
    qr, kr, vr = ref64(q), ref64(k), ref64(v)
    mk = km[:, None] if masked else torch.ones(B, 1, Tk, dtype=torch.bool)
    outr = O.full_attention(qr, kr, vr, mk, n_head)
    outr.backward(dO.double())

    def run():
        qd, kd, vd = leaf(cl(q)), leaf(cl(k)), leaf(cl(v))
        with torch.enable_grad():
            out = ops.attention(qd, kd, vd, km.to(DEV) if masked else None, n_head)
        out.backward(cl(dO).to(DEV))
        return qd.grad, kd.grad, vd.grad, out.detach()
    got = run()
    fused = precision != "f32"
    # bf16-split products (2^-17 each, five in a chain) against exact f32 ones; the f16x3 mode's f16 planes (dO and dS at
    # power-of-two factors from the absolute maxima of dO and v) are held to the exact-f32 form's bound
    tol = 1e-4 if precision == "bf16x3" else 2e-5
    for name, a, r in zip(("dq", "dk", "dv"), got, (qr.grad, kr.grad, vr.grad)):
        rel_close(a, cl(r), tol, name)
    # the forward of the pair (vrd_attention_rows in the split modes): f16 planes are held to the f32 kernels' bound
    rel_close(got[3], cl(outr), 1e-4 if precision == "bf16x3" else 2e-5, "out")
    if fused:
        try:
            autograd.FUSED_ATTN_BWD = False
            old = run()
        finally:
            autograd.FUSED_ATTN_BWD = True
        for name, a, r in zip(("dq", "dk", "dv", "out"), got, old):
            rel_close(a, r, 1e-4, name + " vs five-product form")
        assert not bool(got[1][2, 2:].any()) and not bool(got[2][2, 2:].any()) if masked else True       # masked keys: zero dk / dv


def test_maxpool_and_mask_head_backward():
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(3)
    B, T, C = 3, 16, 512
    m = mask_for(B, T, [16, 9, 2])
    x = torch.randn(B, C, T, generator=g) * m[:, None]
    dy = torch.randn(B, C, T // 2, generator=g)
    xr = ref64(x)
    yr = torch.nn.functional.max_pool1d(xr, 3, 2, 1) * m[:, None, ::2].double()
    yr.backward(dy.double())
    xd = leaf(cl(x))
    with torch.enable_grad():
        y, m2 = ops.maxpool_mask(xd, m.to(DEV))
    y.backward(cl(dy).to(DEV))
    assert torch.equal(m2.cpu(), m[:, ::2])
    rel_close(y, cl(yr), 1e-6, "pool")
    rel_close(xd.grad, cl(xr.grad), 1e-6, "dpool")
    # mask head
    Q, Dp = 9, 256
    emb, feat = torch.randn(B, Q, Dp, generator=g), torch.randn(B, T, Dp, generator=g)
    dseg = torch.randn(B, Q, T, generator=g)
    er, fr = ref64(emb), ref64(feat)
    sr = torch.einsum("bqc,btc->bqt", er, fr).masked_fill(~m[:, None], -10.0)
    sr.backward(dseg.double())
    ed, fd = leaf(emb), leaf(feat)
    with torch.enable_grad():
        seg = ops.mask_head(ed, fd, m.to(DEV), -10.0)
    seg.backward(dseg.to(DEV))
    rel_close(seg, sr, 2e-5, "seg")
    rel_close(ed.grad, er.grad, 2e-5, "demb")
    rel_close(fd.grad, fr.grad, 2e-5, "dfeat")


@pytest.mark.parametrize("k", [1, 3])
def test_split_weight_operands_are_the_tensor_expression_bit_for_bit(k):
    """vrd_split_weight (from the parameter) against what it replaces: hi = bf16(W), lo = bf16(W - hi) of the tap-major
    packed weight in blocks of 32, for the forward operand and for the transposed, tap-flipped operand of the
    input-gradient GEMM; in the f16 format hi = f16(y), lo = f16(y - hi) of y = W * 2^e_w with max |W| * 2^e_w in [2^14, 2^15)."""
    import math
    from vrdone_amd import _hip, ops

    def expression(w, f16=False):              # w: Conv1d weight (N, Cin, k), contiguous
        packed = w.permute(0, 2, 1).contiguous().reshape(w.shape[0], -1)
        dt = torch.float16 if f16 else torch.bfloat16
        ew = 15 - math.frexp(float(packed.abs().max()))[1] if f16 else 0       # frexp: max = m * 2^e, m in [0.5, 1)
        packed = packed * 2.0 ** ew
        hi = packed.to(dt)
        lo = (packed - hi.float()).to(dt)
        n, kk = hi.shape
        return torch.stack([hi.reshape(n, kk // 32, 32), lo.reshape(n, kk // 32, 32)], dim=2), ew

    g = torch.Generator().manual_seed(11)
    w = (torch.randn(96, 64, k, generator=g) * 3).to(DEV)
    with ops.use_precision("bf16x3"):
        got = ops.split_conv_weight(w)
        assert got.fmt == _hip.PAIR_BF16 and got.scale is None and torch.equal(got.t, expression(w)[0])
        assert torch.equal(ops.split_conv_weight_dgrad(w).t, expression(w.flip(2).permute(1, 0, 2).contiguous())[0])
        # cached per weight version: an in-place update rebuilds both
        first = ops.split_conv_weight(w)
        assert ops.split_conv_weight(w) is first
        w.mul_(0.5)
        assert ops.split_conv_weight(w) is not first and torch.equal(ops.split_conv_weight(w).t, expression(w)[0])
    with ops.use_precision("f16x3"):
        for scale in (1.0, 1e-3, 700.0):          # per-tensor exponent: whatever the weight's magnitude, the planes are full
            w2 = (w * scale).contiguous()
            got = ops.split_conv_weight(w2)
            want, ew = expression(w2, f16=True)
            assert got.fmt == _hip.PAIR_F16 and torch.equal(got.t, want)
            sc = got.scale.cpu().tolist()
            assert sc[0] == 2.0 ** -(ew + _hip.F16_ACT_EXP) and sc[1] == 2.0 ** ew
            assert 2 ** 14 <= float(got.t[:, :, 0].float().abs().max()) <= 2 ** 15
        z = ops.split_conv_weight(torch.zeros(32, 32, 1, device=DEV))          # an all-zero weight: exponent 0
        assert z.scale.cpu().tolist()[:2] == [2.0 ** -_hip.F16_ACT_EXP, 1.0] and not bool(z.t.any())
        # the input-gradient operand is in the mode's format too (ops.backward_fmt): f16 planes of the transposed, tap-flipped
        # weight at its own power of two
        gd = ops.split_conv_weight_dgrad(w)
        want_t, ew_t = expression(w.flip(2).permute(1, 0, 2).contiguous(), f16=True)
        assert gd.fmt == _hip.PAIR_F16 and torch.equal(gd.t, want_t) and gd.scale.cpu().tolist()[1] == 2.0 ** ew_t
