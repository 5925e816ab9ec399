"""Training step on the HIP path (BASELINE config 3): module-level forward+backward against autograd
through the oracle, and one whole optimisation step -- losses and the gradient of EVERY parameter -- against gradients
produced by the REAL reference's forward_training + autograd (tests/golden/train_step_vidvrd.*,
scripts/make_golden_train.py), with stochastic depth off and with pinned keep decisions.

Stated tolerances: losses 1e-5 relative (same Hungarian assignments: the reference's recorded ones are replayed, so a
near-tie between equivalent queries cannot change the loss function); gradients, per parameter, relative to the l2 norm
of the reference gradient: MEDIAN over the 521 parameters <= 2e-5 in f32 mode (measured 8e-7) and <= 5e-4 in bf16x3
mode (measured 4e-5); WORST parameter <= 3e-2.  Why the worst is loose: the network has two kinks -- ReLU after the
embedding LayerNorms and the arg-max of MaxPool1d in the branch blocks -- and with activations that differ from the
reference's by 1e-6 one element per step or so lands on the other side of one (measured: one ReLU gate of channel 206
of visual_embd_norm.0 in the "nodrop" case, 8e-4 on that weight; one max-pool arg-max of pair 21 at level 1 in the
"pinned" case, 1e-3 on everything below it).  Kernel-level exactness is what tests/test_gpu_backward.py and the two
module tests below pin (float64 oracle, 2e-5 / 5e-5), and scripts/dev/block_isolate.py shows each block of this very
step matching the float64 oracle to 1e-6 on its real activations."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_case
from golden_cases import compare_grads, replay_matching, train_batch
from oracle import vrd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _autograd_on():
    """Other GPU test modules switch autograd off globally at import; these tests need it recording."""
    with torch.enable_grad():
        yield


def c_in(mc):
    cc = mc["clip_dim"] if mc.get("with_clip_feature", False) else 0
    return 2 * mc["visual_dim"] + 2 * cc + mc["bbox_so_dim"] + 2 * mc["bbox_entity_dim"]


@pytest.fixture(params=["f32", "bf16x3", "f16x3"])
def precision(request):
    from vrdone_amd import ops
    old = ops.get_precision()
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision(old)


def build(name="vidvrd"):
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, ic, keys = load_case(name)
    sd = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(sd, strict=True)
    return model.to(DEV), mc, sd


def rel(got, want, floor=0.0):
    """l2 error relative to the l2 norm of `want` (+ floor: some gradients are mathematically zero -- a key LayerNorm's
    bias shifts every score of a query equally -- and hold only rounding noise on both sides)."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape and bool(torch.isfinite(got).all())
    return float((got - want).norm()) / (float(want.norm()) + floor + 1e-12)


def check_param_grads(module, ref_grads, prefix, tol):
    floor = 1e-3 * max(float(g.norm()) for g in ref_grads.values())
    for name, p in module.named_parameters():
        assert p.grad is not None, name
        assert rel(p.grad, ref_grads[prefix + name], floor) < tol, name


@pytest.mark.parametrize("stride,win", [(1, 7), (2, 7), (2, -1)])
def test_transformer_block_forward_backward_vs_oracle(stride, win, precision):
    """Stage-1 criterion: one TransformerBlock (LN -> depthwise conv + LN -> q/k/v GEMMs -> banded attention -> projection
    + drop-path scale + (max-pooled) skip -> LN -> MLP) forward and backward, every parameter gradient and the input
    gradient, against float64 autograd of the oracle's restatement of reference blocks.py:1070-1080."""
    from vrdone_amd.models.blocks import TransformerBlock
    torch.manual_seed(0)
    C, H, B, T = 512, 4, 3, 48
    blk = TransformerBlock(C, H, n_ds_strides=(stride, stride), path_pdrop=0.1, mha_win_size=win)     # win -1: global attention
    keys = [(f"blk.{k}", list(v.shape)) for k, v in blk.state_dict().items()]
    sd = O.synth_state_dict(keys)
    blk.load_state_dict({k[4:]: v for k, v in sd.items()})
    blk = blk.to(DEV).eval()                      # eval: no stochastic depth (the oracle has none); autograd still records
    lens = torch.tensor([48, 31, 6])
    m = (torch.arange(T)[None] < lens[:, None])[:, None]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, T, generator=g) * m
    dy = torch.randn(B, C, T // stride, generator=g)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    yr, _ = O.transformer_block(sd64, "blk", x64, m, H, win, stride)
    yr.backward(dy.double())
    xd = x.to(DEV).requires_grad_(True)
    with torch.enable_grad():
        y, _ = blk(xd, m.to(DEV))
    y.backward(dy.to(DEV))
    tol = 5e-5 if precision == "f32" else 5e-4
    assert rel(y, yr) < tol
    assert rel(xd.grad, x64.grad) < tol
    check_param_grads(blk, {k: v.grad for k, v in sd64.items()}, "blk.", tol)


def test_sos_decoder_layer_forward_backward_vs_oracle(precision):
    """The subject-object mutual attention layer (self-attention + cross-attention with global masked attention, no FFN;
    reference local_transformer.py:807-835) forward and backward."""
    from vrdone_amd.models.local_transformer import MaskedConvTransformerDecoderLayer
    C, H, B, T = 512, 4, 2, 48
    layer = MaskedConvTransformerDecoderLayer(C, H, path_pdrop=0.1, n_qx_stride=1, n_kv_stride=1, with_ffn=False, use_local=False)
    keys = [(f"sos.{k}", list(v.shape)) for k, v in layer.state_dict().items()]
    sd = O.synth_state_dict(keys)
    layer.load_state_dict({k[4:]: v for k, v in sd.items()})
    layer = layer.to(DEV).eval()
    lens = torch.tensor([48, 17])
    m = (torch.arange(T)[None] < lens[:, None])[:, None]
    g = torch.Generator().manual_seed(2)
    x, y_in = torch.randn(B, C, T, generator=g) * m, torch.randn(B, C, T, generator=g) * m
    dy = torch.randn(B, C, T, generator=g)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64, y64 = x.double().requires_grad_(True), y_in.double().requires_grad_(True)
    outr = O.decoder_layer(sd64, "sos", x64, y64, m, m, H)
    outr = outr[0] if isinstance(outr, tuple) else outr
    outr.backward(dy.double())
    xd, yd = x.to(DEV).requires_grad_(True), y_in.to(DEV).requires_grad_(True)
    with torch.enable_grad():
        out, _ = layer(xd, yd, m.to(DEV), m.to(DEV))
    out.backward(dy.to(DEV))
    tol = 5e-5 if precision == "f32" else 5e-4
    assert rel(out, outr) < tol
    assert rel(xd.grad, x64.grad) < tol and rel(yd.grad, y64.grad) < tol
    check_param_grads(layer, {k: v.grad for k, v in sd64.items()}, "sos.", tol)


# Parameters whose gradient does not pass through a max-pool of the branch blocks on its way back from the loss.  Why the
# "at most three parameters beyond 1e-3" bound of the pinned case is restricted to them: a max-pool arg-max between two frames
# whose values differ by less than the forward's rounding error (~1e-6 relative) can resolve differently than in the reference,
# which moves one element's gradient from one frame to another -- 1.7e-4 of the gradient arriving at branch.0's output -- and the
# subject / object LayerNorms in front of the fusion MLP shrink the gradient norm 250-fold there (55 -> 0.22), so every parameter
# further upstream sees that one flip at 1e-3 .. 6e-3.  Measured for the pinned case in f32 mode: sequence 21, frames 21 / 23,
# branch.1's skip pool (scripts/dev/pinned_bisect4.py against float64 gradients of the reference at every block output;
# branch.1 alone on the same tensors agrees with float64 autograd to 8e-7).  The reference's own f32 gradients sit 1e-6 from its
# float64 ones, so the golden is exact at this level; which precision mode meets a flip on a given batch is chance (the bf16x3
# mode does for other drop patterns).  The worst-case bound (3e-2) and the median bound stay on every parameter.
POOL_FREE = r"(backbone\.branch\.[12]\.|neck\.|predictor\.)"


@pytest.mark.parametrize("case", ["nodrop", "pinned"])
def test_training_step_matches_reference_gradients(case, precision):
    """model.train()(batch) -> total_loss.backward() on the HIP path vs the reference's own training step."""
    from vrdone_amd.models.blocks import AffineDropPath
    model, mc, _ = build()
    with open(os.path.join(GOLDEN, "train_step_vidvrd.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "train_step_vidvrd.npz"))
    lens, _, _, data = train_batch(mc, c_in(mc), device=DEV)
    assert lens == meta["lengths"]
    model.train()
    n_dp = 0
    for name, mod in model.named_modules():
        if isinstance(mod, AffineDropPath):
            n_dp += 1
            if case == "nodrop":
                mod.drop_prob = 0.0
            else:
                mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
    assert n_dp == len(meta["keep"]) == 30
    differing = replay_matching(model, meta["cases"][case]["indices"])
    with torch.enable_grad():
        loss = model(data)
        loss["total_loss"].backward()
    want = meta["cases"][case]["losses"]
    assert set(loss) == set(want)
    for k, v in want.items():
        assert abs(float(loss[k].detach()) - v) <= (1e-5 if precision == "f32" else 2e-4) * max(1.0, abs(v)), (k, float(loss[k]), v)
    # own matching = the reference's except a few near-ties (pairs of < 16 frames have one valid frame at the predictor's
    # T/8 level, and the first decoder layers' queries are still close to each other): with the reference's assignments
    # replayed the losses agree to 1e-6, so a flipped assignment is a tie, not a different prediction
    assert all(len(call) <= 6 for call in differing), differing
    worst, median = compare_grads(((n, p.grad) for n, p in model.named_parameters()), g, meta, case,
                                  rtol=3e-2, atol_frac=1e-4, median_tol=2e-5 if precision == "f32" else 5e-4, outlier_tol=1e-3,
                                  max_outliers=3, outlier_scope=None if case == "nodrop" and precision != "bf16x3" else POOL_FREE)
    print(f"[{case}/{precision}] relative gradient error: worst {worst:.2e}, median {median:.2e}")


def grads_vs_float64(model, g32, d64, meta, case):
    """Per parameter: (|ours - ref64|, |ref32 - ref64|, |ref64|) as l2 norms over the stored entries (the full gradient up to 2048
    elements, else the stride-499 sample); ref64 = ref32 + d (tests/golden/train_step_vidvrd_f64.npz)."""
    stride = meta["sample_stride"]
    out = {}
    for name, p in model.named_parameters():
        g = p.grad.detach().double().cpu()
        got = (g if g.numel() <= 2048 else g.flatten()[::stride]).numpy()
        r32 = g32[f"{case}/{name}"].astype(np.float64)
        r64 = r32 + d64[f"{case}/{name}"].astype(np.float64)
        out[name] = (float(np.linalg.norm(got - r64)), float(np.linalg.norm(r32 - r64)), float(np.linalg.norm(r64)))
    return out


F64_REPORT = {}


@pytest.mark.parametrize("case", ["nodrop", "pinned"])
def test_training_step_gradients_against_float64_reference(case, precision):
    """Every parameter's gradient against the REFERENCE differentiated in float64 (scripts/make_golden_train_f64.py: same batch,
    same pinned stochastic depth, same replayed matching), measured in units of the reference's own float32 error e32 =
    |ref32 - ref64| of that parameter: |ours - ref64| <= FACTOR * e32 + EPS * |ref64|.  The parameters that exceed the bound
    must be upstream of a branch block's max-pool (the arg-max flip documented at POOL_FREE) and are listed by name in the
    failure message."""
    from vrdone_amd.models.blocks import AffineDropPath
    model, mc, _ = build()
    with open(os.path.join(GOLDEN, "train_step_vidvrd.json")) as f:
        meta = json.load(f)
    g32 = np.load(os.path.join(GOLDEN, "train_step_vidvrd.npz"))
    d64 = np.load(os.path.join(GOLDEN, "train_step_vidvrd_f64.npz"))
    lens, _, _, data = train_batch(mc, c_in(mc), device=DEV)
    model.train()
    for name, mod in model.named_modules():
        if isinstance(mod, AffineDropPath):
            if case == "nodrop":
                mod.drop_prob = 0.0
            else:
                mod.keep = torch.tensor(meta["keep"][name], dtype=torch.float32)
    replay_matching(model, meta["cases"][case]["indices"])
    with torch.enable_grad():
        loss = model(data)
        loss["total_loss"].backward()
    want64 = float(d64[f"{case}/total_loss"])
    e_loss, e32_loss = abs(float(loss["total_loss"].detach().double()) - want64), abs(meta["cases"][case]["losses"]["total_loss"] - want64)
    res = grads_vs_float64(model, g32, d64, meta, case)
    biggest = max(v[2] for v in res.values())
    # in units of e32 (+ a floor of 1e-7 of the parameter's own gradient norm and 1e-9 of the largest: gradients that are
    # mathematically zero hold rounding noise only)
    ratio = {n: e / (e32 + 1e-7 * r + 1e-9 * biggest) for n, (e, e32, r) in res.items()}
    v = np.array(list(ratio.values()))
    pool_free = np.array([ratio[n] for n in ratio if re.match(POOL_FREE, n)])
    F64_REPORT[(case, precision)] = (float(np.median(v)), float(np.percentile(v, 90)), float(v.max()), float(np.median(pool_free)),
                                     float(pool_free.max()), e_loss, e32_loss)
    print(f"[{case}/{precision}] |ours - ref64| in units of the reference's own f32 error: median {np.median(v):.1f}, 90 % "
          f"{np.percentile(v, 90):.1f}, max {v.max():.0f}; downstream of the pools: median {np.median(pool_free):.1f}, max {pool_free.max():.0f}; "
          f"total_loss |ours - ref64| {e_loss:.2e} (ref32: {e32_loss:.2e})")
    # Bounds in units of e32, by mode (measured, profiles/r05_grad_f64_distance.txt): f32 -- downstream of the pools at most
    # 2 x, median 1.0-1.7 x: the backward kernels are at the reference's own float32 error; f16x3 -- its backward GEMMs form
    # their products on f16 planes too, the gradient at a per-tensor power of two (ops.grad_scale) -- at most 2 x, median
    # 0.8-1.1 x, held to the f32 mode's bounds (on bf16 planes, VRDONE_F16_BACKWARD=0, it sat at 5-11 x median, 16-26 x
    # worst); bf16x3 at most 80 x, median 21-49 x.
    # Held to the bound: every parameter downstream of the branch pools (POOL_FREE) in both cases, and in the no-drop case all
    # parameters except the first visual embedding layer (FIRST_LAYER: its LayerNorm feeds a ReLU whose gates, ~2 million
    # elements, flip where |y| < 1e-6: 12-388 x in f32).  Upstream of branch.1's pool the pinned case carries the arg-max flip
    # of POOL_FREE's comment: every such parameter sits at 400-3,700 x in f32 -- they are the complement of POOL_FREE, listed in
    # the failure message if one of them is NOT the reason.
    FIRST_LAYER = r"backbone\.visual_embd(_norm)?\.0\."
    bound = {"f32": 4.0, "f16x3": 4.0, "bf16x3": 500.0}[precision]
    median_bound = {"f32": 3.0, "f16x3": 3.0, "bf16x3": 80.0}[precision]
    # (the 17-bit bf16x3 mode meets an arg-max tie of the branch pools in the no-drop case too -- which mode meets one on a given
    # batch is chance: POOL_FREE's comment --, so it is held to the bound downstream of the pools only, in both cases)
    wide = case == "nodrop" and precision != "bf16x3"
    held = [n for n in ratio if re.match(POOL_FREE, n) or (wide and not re.match(FIRST_LAYER, n))]
    assert len(held) >= (500 if wide else 250)
    beyond = sorted((n, round(ratio[n], 1)) for n in held if ratio[n] > bound)
    assert not beyond, f"{len(beyond)} parameters beyond {bound} x the reference's own f32 error: {beyond[:10]}"
    assert float(np.median([ratio[n] for n in held])) <= median_bound
    assert e_loss <= 4.0 * e32_loss + (1e-6 if precision != "bf16x3" else 5e-4) * abs(want64)
    free = sorted(n for n in ratio if n not in held and ratio[n] > bound)
    print(f"   not held to the bound and beyond it: {len(free)} parameters" + (f", e.g. {free[:3]}" if free else ""))


@pytest.mark.parametrize("name", ["vidor", "vidor_x", "vidor_local"])
def test_training_step_vidor_matches_reference_gradients(name, precision):
    """The other shipped training shapes, T = 512, ragged pairs, stochastic depth off, against the reference's own step
    (tests/golden/train_step_<name>.*, scripts/make_golden_train.py --vidor / --vidor-variants).  vidor.yaml: 8 heads of 64
    channels, window 9, 50 classes -- the attention backward as matrix-core products (512 x 512 scores), the weight-gradient
    tiles on 6,144 rows; vidor_x.yaml: the CLIP backbone (10 queries); vidor_local.yaml: banded attention in the SOS layers."""
    from golden_cases import TRAIN_SPECS
    from vrdone_amd.models.blocks import AffineDropPath
    model, mc, _ = build(name)
    with open(os.path.join(GOLDEN, f"train_step_{name}.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, f"train_step_{name}.npz"))
    lens, _, _, data = train_batch(mc, c_in(mc), device=DEV, spec=TRAIN_SPECS[name])
    assert lens == meta["lengths"] and meta["T"] == 512
    model.train()
    for mod in model.modules():
        if isinstance(mod, AffineDropPath):
            mod.drop_prob = 0.0
    differing = replay_matching(model, meta["cases"]["nodrop"]["indices"])
    with torch.enable_grad():
        loss = model(data)
        loss["total_loss"].backward()
    want = meta["cases"]["nodrop"]["losses"]
    assert set(loss) == set(want)
    for k, v in want.items():
        assert abs(float(loss[k].detach()) - v) <= (1e-5 if precision == "f32" else 2e-4) * max(1.0, abs(v)), (k, float(loss[k]), v)
    assert all(len(call) <= 2 for call in differing), differing
    # At T = 512 a max-pool arg-max tie in a branch block (POOL_FREE above) resolves against the reference on about every other
    # batch and precision mode (seen while choosing the seeds: vidor_x in f32 mode with one input seed, in bf16x3 mode with
    # another) and lifts every parameter upstream to 1e-4 .. 5e-3; the embedding LayerNorms' ReLU gates flip likewise (~2
    # million elements per gate, |y| < 1e-6 decides).  So: the strict bounds on the parameters downstream of the branch pools,
    # the worst-case and a 5e-4 median bound on all of them.
    named = [(n, p.grad) for n, p in model.named_parameters()]
    worst, median = compare_grads(named, g, meta, "nodrop", rtol=3e-2, atol_frac=1e-4, median_tol=5e-4)
    strict = [(n, gr) for n, gr in named if re.match(POOL_FREE, n)]
    assert len(strict) > 250
    w2, m2 = compare_grads(strict, g, meta, "nodrop", rtol=3e-2, atol_frac=1e-4, median_tol=2e-5 if precision == "f32" else 5e-4,
                           outlier_tol=1e-3, max_outliers=3)
    print(f"[{name}/{precision}] relative gradient error: worst {worst:.2e}, median {median:.2e}; downstream of the pools: {w2:.2e}, {m2:.2e}")


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_presplit_weights_equal_the_per_weight_launches(mode):
    """ops.presplit_weights (one vrd_split_weights call for all dense conv weights of a training step) leaves, in every
    weight's operand caches, exactly what split_conv_weight / split_conv_weight_dgrad build one weight at a time -- planes and,
    in the f16 format, the per-tensor scale; a later in-place update of a weight invalidates its entries; a change of mode
    builds a new plan and drops the old one."""
    from vrdone_amd import ops
    fwd = "_vrd_split_f16" if mode == "f16x3" else "_vrd_split"
    bits = lambda sw: sw.t.view(torch.int16)        # noqa: E731
    with ops.use_precision(mode):
        # (the backward GEMMs' operands are in the mode's format too, unless VRDONE_F16_BACKWARD=0 keeps them on bf16 planes)
        from vrdone_amd import _hip
        bwd = "_vrd_split_t_f16" if ops.backward_fmt() == _hip.PAIR_F16 else "_vrd_split_t"
        slots = (fwd, bwd)
        model, _, _ = build()
        ws = model._dense_conv_weights()
        assert len(ws) > 100
        want = {}
        for w in ws:
            N, Cin, k = w.shape
            if (Cin * k) % 32 == 0:
                sw = ops.split_conv_weight(w)
                want[(id(w), fwd)] = (sw.t.clone(), None if sw.scale is None else sw.scale.clone(), sw.fmt)
            if (N * k) % 32 == 0:
                sw = ops.split_conv_weight_dgrad(w)
                want[(id(w), bwd)] = (sw.t.clone(), None if sw.scale is None else sw.scale.clone(), sw.fmt)
            for slot in slots:
                if hasattr(w, slot):
                    delattr(w, slot)
        plans = {}
        ops.presplit_weights(ws, plans)
        assert len(plans) == 1 and len(want) > 200
        for w in ws:
            for slot in slots:
                if (id(w), slot) in want:
                    key, val = getattr(w, slot)[:2]
                    t, scale, fmt = want[(id(w), slot)]
                    assert key == (w.data_ptr(), w._version) and val.fmt == fmt
                    assert torch.equal(bits(val), t.view(torch.int16)), slot
                    if scale is not None:
                        assert torch.equal(val.scale[:2], scale[:2]), slot
        w0 = ws[0]
        assert ops.split_conv_weight(w0) is getattr(w0, fwd)[1]                    # served from the cache
        with torch.no_grad():
            w0.mul_(1.5)
        fresh = ops.split_conv_weight(w0)                                          # stale entry: rebuilt for the new version
        assert not torch.equal(bits(fresh), want[(id(w0), fwd)][0].view(torch.int16))
        ops.presplit_weights(ws, plans)                                            # the next step: same plan, new values
        assert len(plans) == 1 and torch.equal(bits(getattr(w0, fwd)[1]), bits(fresh))
        with ops.use_precision("f16x3" if mode == "bf16x3" else "bf16x3"):         # the other element format: its own plan
            ops.presplit_weights(ws, plans)
            assert len(plans) == 1


def test_drop_path_sampling_statistics():
    """AffineDropPath in training mode: per-sample factors are 0 or 1/keep_prob, E[factor] = 1, one decision per sample
    repeated over its rows; nothing is sampled in eval or under no_grad."""
    from vrdone_amd.models.blocks import AffineDropPath
    dp = AffineDropPath(512, drop_prob=0.1).to(DEV).train()
    torch.manual_seed(0)
    with torch.enable_grad():
        f = dp.row_factors(20000, 3, DEV)
    assert f.shape == (60000,)
    per = f.view(20000, 3)
    assert bool((per == per[:, :1]).all())
    vals = sorted(per[:, 0].unique().cpu().tolist())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / 0.9) < 1e-6
    assert abs(float(per[:, 0].mean()) - 1.0) < 0.02
    with torch.no_grad():
        assert dp.row_factors(4, 3, DEV) is None
    with torch.enable_grad():
        assert dp.eval().row_factors(4, 3, DEV) is None


def test_optimisation_steps_reduce_the_loss():
    """scripts/train_step.py (own code mirroring reference train.py:176-191: forward, zero_grad, backward, grad-norm
    clipping, AdamW with the reference's weight-decay grouping, EMA update): four steps on the 24-pair batch (stochastic
    depth off, so that the loss is the same function at every step); every parameter gets a finite gradient each step,
    parameters move, the EMA lags, the loss goes down."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_step", os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "train_step.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    log = ts.run(steps=4, seed=0, device=DEV, verbose=False, lr=2e-5, drop_path=False)
    assert len(log["total_loss"]) == 4 and all(np.isfinite(log["total_loss"]))
    assert log["total_loss"][-1] < log["total_loss"][0]
    assert log["params_without_grad"] == [] and log["nonfinite_grads"] == []
    assert log["param_delta_norm"] > 0 and 0 < log["ema_delta_norm"] < log["param_delta_norm"]


def test_training_graphs_take_the_same_steps():
    """MaskVRD.enable_training_graphs() (vrdone_amd/train_graph.py): the network's forward and backward recorded as two HIP
    graphs and replayed.  Four optimisation steps with and without (stochastic depth off): the same losses and the same
    parameter movement -- the weights change between the steps, so a replay that used operands derived from the weights at
    recording time (the per-weight caches of ops.py) would show from step 1 on; a second batch shape records its own
    graphs; eval and no_grad calls stay eager."""
    import importlib.util
    from vrdone_amd import train_graph
    spec = importlib.util.spec_from_file_location("train_step", os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "train_step.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    eager = ts.run(steps=4, seed=0, device=DEV, verbose=False, lr=2e-5, drop_path=False)
    graphed = ts.run(steps=4, seed=0, device=DEV, verbose=False, lr=2e-5, drop_path=False, graphs=True)
    np.testing.assert_allclose(graphed["total_loss"], eager["total_loss"], rtol=2e-5)
    assert graphed["total_loss"][-1] < graphed["total_loss"][0]
    assert graphed["params_without_grad"] == [] and graphed["nonfinite_grads"] == []
    np.testing.assert_allclose(graphed["param_delta_norm"], eager["param_delta_norm"], rtol=1e-3)
    np.testing.assert_allclose(graphed["ema_delta_norm"], eager["ema_delta_norm"], rtol=1e-3)

    # gradients of one step, graph against eager, on two batch shapes
    from vrdone_amd import configs, synth
    from vrdone_amd.models.blocks import AffineDropPath
    from vrdone_amd.models.maskvrd import MaskVRD
    cfg = configs.model_config("vidvrd")
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=DEV)).to(DEV).train()
    for mod in model.modules():
        if isinstance(mod, AffineDropPath):
            mod.drop_prob = 0.0
    for n_pairs in (24, 7, 24):
        data = ts.synthetic_batch(cfg, configs.input_channels(cfg), DEV, n_pairs=n_pairs, seed=3 + n_pairs)
        grads = []
        for on in (False, True):
            model.enable_training_graphs(on)
            model.zero_grad(set_to_none=True)
            loss = model(data)["total_loss"]
            loss.backward()
            grads.append((float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}))
        assert abs(grads[0][0] - grads[1][0]) <= 2e-5 * abs(grads[0][0])
        floor = 1e-6 * max(float(g.abs().max()) for g in grads[0][1].values())      # (atomics order: rounding-level noise)
        for k, g in grads[0][1].items():
            assert float((grads[1][1][k] - g).abs().max()) <= 2e-4 * float(g.abs().max()) + floor, k
        with torch.no_grad():                   # weights move: the next replay must see them
            for p in model.parameters():
                p.mul_(1.01)
    assert len(train_graph.recordings(model)) == 2          # 24 pairs, 7 pairs; the third batch replayed the first recording
    with torch.no_grad():
        assert np.isfinite(float(model(data)["total_loss"]))          # validation pass: eager, fused kernels
    # parameters that move to new memory invalidate the recordings (re-recorded on the next step)
    model.enable_training_graphs(True)
    with torch.no_grad():
        model.backbone.stem[0].mlp[0].weight.data = model.backbone.stem[0].mlp[0].weight.data.clone()
    model.zero_grad(set_to_none=True)
    loss = model(data)["total_loss"]
    loss.backward()
    assert len(train_graph.recordings(model)) == 1 and np.isfinite(float(loss.detach()))
    assert model.backbone.stem[0].mlp[0].weight.grad is not None
    model.enable_training_graphs(False)
    assert not train_graph.enabled(model)
    train_graph.forget(model)
    assert not train_graph.recordings(model)


def test_training_graph_survives_a_step_in_another_precision_mode():
    """A recording holds raw device pointers into the split-operand plan it captured (job table, chunk tables, operand
    buffers).  ops.presplit_weights drops the model's plans of other modes, so a step in bf16x3 evicts the f16x3 plan from
    the model's dict: the f16x3 recording must keep it alive itself (`_Recording.split_plans`) and its replay must still
    give the eager step's loss and gradients."""
    import importlib.util
    from vrdone_amd import configs, ops, synth, train_graph
    from vrdone_amd.models.blocks import AffineDropPath
    from vrdone_amd.models.maskvrd import MaskVRD
    spec = importlib.util.spec_from_file_location("train_step", os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "train_step.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    cfg = configs.model_config("vidvrd")
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=DEV)).to(DEV).train()
    for mod in model.modules():
        if isinstance(mod, AffineDropPath):
            mod.drop_prob = 0.0
    data = ts.synthetic_batch(cfg, configs.input_channels(cfg), DEV, n_pairs=12, seed=11)

    def step(graphs):
        model.enable_training_graphs(graphs)
        model.zero_grad(set_to_none=True)
        loss = model(data)["total_loss"]
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in model.named_parameters()}

    old = ops.get_precision()
    try:
        ops.set_precision("f16x3")
        step(True)                                              # records the f16x3 graphs
        rec = next(iter(train_graph.recordings(model).values()))
        assert rec.split_plans, "the recording must hold the plan its captures point into"
        ops.set_precision("bf16x3")
        step(False)                                             # an eager step in the other mode: evicts the f16x3 plan
        assert len(model._split_plans) == 1
        torch.cuda.empty_cache()                                # freed operand buffers really go back to the device
        junk = torch.full((64 << 20,), float("nan"), device=DEV)       # ... and whatever reuses their memory is not a weight operand
        ops.set_precision("f16x3")
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.01)                                    # the replay must re-split the weights as they are now
        want = step(False)
        got = step(True)
        assert len(train_graph.recordings(model)) == 1          # replayed, not re-recorded
        del junk
    finally:
        ops.set_precision(old)
        train_graph.forget(model)
    assert np.isfinite(got[0]) and abs(got[0] - want[0]) <= 2e-5 * abs(want[0])
    floor = 1e-6 * max(float(g.abs().max()) for g in want[1].values())
    for k, g in want[1].items():
        assert float((got[1][k] - g).abs().max()) <= 2e-4 * float(g.abs().max()) + floor, k


def test_device_assignment_equals_scipy():
    """vrd_assign (one thread per pair, Hungarian with potentials) against scipy.optimize.linear_sum_assignment -- what the
    reference's matcher calls per pair (models/maskvrd.py:492) -- on random cost blocks of every size N <= Q, Q = 9, 10, 16."""
    from scipy.optimize import linear_sum_assignment
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(0)
    for Q in (9, 10, 16):
        sizes = [int(n) for n in torch.randint(0, Q + 1, (200,), generator=g)]
        sizes[:3] = [Q, 1, 0]
        cost = torch.randn(sum(sizes), Q, generator=g)
        got = ops.assign(cost.to(DEV), sizes).cpu()
        at = 0
        for n in sizes:
            if n:
                rows, cols = linear_sum_assignment(cost[at:at + n].T.numpy())       # rows = queries, cols = relations
                want = torch.empty(n, dtype=torch.int32)
                want[torch.as_tensor(cols)] = torch.as_tensor(rows, dtype=torch.int32)
                assert torch.equal(got[at:at + n], want), (Q, n)
            at += n


def test_device_assignment_terminates_on_nan_and_infinite_costs():
    """A diverged step hands the matcher NaN / infinite costs.  scipy's linear_sum_assignment raises ValueError on such a
    matrix; vrd_assign has to come back (its augmenting search used to spin for ever there) with those pairs unassigned,
    the pairs around them solved as usual."""
    from scipy.optimize import linear_sum_assignment
    from vrdone_amd import ops
    g = torch.Generator().manual_seed(1)
    Q, sizes = 9, [3, 4, 9, 2, 5, 1]
    cost = torch.randn(sum(sizes), Q, generator=g)
    first = [0, 3, 7, 16, 18, 23]
    cost[first[1] + 2] = float("nan")                       # pair 1: one relation's whole row
    cost[first[2]:first[2] + 9] = float("inf")              # pair 2: everything
    cost[first[4] + 1, 3] = float("nan")                    # pair 4: a single entry, which the search itself need not visit
    cost[first[3], 8] = float("-inf")                       # pair 3: a single -inf (scipy: "invalid numeric entries" too)
    got = ops.assign(cost.to(DEV), sizes)
    torch.cuda.synchronize()                                # returns at all
    got = got.cpu()
    assert bool((got[first[1]:first[1] + 4] == -1).all()) and bool((got[first[2]:first[2] + 9] == -1).all())
    for p in (3, 4):        # scipy raises on ANY NaN / -inf entry of a pair's block: so the whole pair comes back unassigned
        with pytest.raises(ValueError):
            linear_sum_assignment(cost[first[p]:first[p] + sizes[p]].T.numpy())
        assert bool((got[first[p]:first[p] + sizes[p]] == -1).all())
    for p in (0, 5):
        n, at = sizes[p], first[p]
        rows, cols = linear_sum_assignment(cost[at:at + n].T.numpy())
        want = torch.empty(n, dtype=torch.int32)
        want[torch.as_tensor(cols)] = torch.as_tensor(rows, dtype=torch.int32)
        assert torch.equal(got[at:at + n], want)
    # the matcher's host form raises like scipy does
    model, mc, _ = build()
    lens, x, m, data = train_batch(mc, c_in(mc), device=DEV)
    with torch.no_grad():
        out = model.eval()._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
        logits = out["pred_logits"].clone()
        logits[1] = float("nan")
        with pytest.raises(ValueError):
            model.bipartite_match(logits, data["preds_list"], out["pred_masks"], data["masks_list"], data["segs_list"], out["output_mask"])


@pytest.mark.parametrize("fuzzy", [False, True])
@pytest.mark.parametrize("L,Q,K1,T", [(4, 9, 133, 96), (2, 10, 51, 200)])
def test_fused_criterion_equals_the_tensor_form(fuzzy, L, Q, K1, T):
    """csrc/vrd_criterion.hip (costs -> vrd_assign -> losses, and the gradient kernel) against the tensor code of
    models/losses.py, which the reference goldens pin: same assignments as scipy on the tensor costs, same loss values, same
    gradients with respect to every layer's logits and masks (hard and fuzzy targets, ragged valid lengths)."""
    from scipy.optimize import linear_sum_assignment
    from vrdone_amd.models import losses
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7 + L + Q + int(fuzzy))
    B = 7
    sizes = [3, 1, 0, Q, 2, 4, 1]
    G = sum(sizes)
    lens = torch.tensor([T, T - 5, 40, T, 17, T // 2, 3])
    valid = (torch.arange(T)[None] < lens[:, None])
    owner = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes))
    ids = torch.randint(1, K1, (G,), generator=g)
    segs = torch.zeros(G, 2, dtype=torch.int64)
    tgt = torch.zeros(G, T)
    for i in range(G):
        n = int(lens[owner[i]])
        a = int(torch.randint(0, max(n - 1, 1), (1,), generator=g))
        b = int(torch.randint(a + 1, n + 1, (1,), generator=g))
        segs[i] = torch.tensor([a, b])
        tgt[i, a:b] = 1.0
    weight = torch.ones(K1)
    weight[0] = 0.1
    cost_w = (2.0, 5.0, 5.0)
    layers = [(torch.randn(B, Q, K1, generator=g).to(DEV).requires_grad_(True), (2.0 * torch.randn(B, Q, T, generator=g)).to(DEV).requires_grad_(True))
              for _ in range(L)]
    seg_d = segs.to(DEV) if fuzzy else None
    vals, q_of, failed = losses.device_criterion(layers, valid.to(DEV), sizes, ids.to(DEV), tgt.to(DEV), seg_d, 0.85, weight, cost_w)
    assert not bool(failed)
    coef = torch.randn(L, 3, generator=g).to(DEV)
    (vals * coef).sum().backward()
    got_grads = [(lg.grad.clone(), mk.grad.clone()) for lg, mk in layers]
    for l, (lg, mk) in enumerate(layers):
        lg.grad = mk.grad = None
        with torch.no_grad():
            cc, cm, cd = losses.pair_costs(lg, mk, valid.to(DEV), ids.to(DEV), tgt.to(DEV), owner.to(DEV), seg_d, 0.85)
            cost = (cost_w[0] * cc + cost_w[1] * cm + cost_w[2] * cd).cpu()
        want_q = torch.empty(G, dtype=torch.int64)
        at = 0
        for n in sizes:
            if n:
                rows, cols = linear_sum_assignment(cost[at:at + n].T.numpy())
                want_q[at + torch.as_tensor(cols)] = torch.as_tensor(rows)
            at += n
        assert torch.equal(q_of[l].cpu().long(), want_q), l
        target = torch.zeros(B, Q, dtype=torch.int64, device=DEV)
        target[owner.to(DEV), want_q.to(DEV)] = ids.to(DEV)
        ce = F.cross_entropy(lg.transpose(1, 2), target, weight.to(DEV))
        focal, dice = losses.matched_losses(mk[owner.to(DEV), want_q.to(DEV)], tgt.to(DEV), float(G), valid.to(DEV)[owner.to(DEV)], seg_d, 0.85)
        want = torch.stack([ce, focal, dice])
        assert float((vals[l] - want).detach().abs().max()) < 2e-6 * max(1.0, float(want.detach().abs().max())), (l, vals[l], want)
        (want * coef[l]).sum().backward()
        for a, b in zip(got_grads[l], (lg.grad, mk.grad)):
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max())) + 1e-7, l


def test_fused_criterion_poisons_a_class_id_out_of_range_and_large_batches_take_the_tensor_form():
    """A ground-truth class id outside [0, K1): the reference's F.cross_entropy raises (models/maskvrd.py:510); the fused
    criterion, which never waits for the device, reads nothing out of bounds and returns NaN losses for the step.  A batch
    whose (pair, query) table exceeds the fused kernel's LDS goes through the tensor form instead of raising."""
    from vrdone_amd.models import losses
    g = torch.Generator().manual_seed(3)
    B, Q, K1, T = 5, 9, 133, 96
    sizes = [2, 1, 3, 0, 1]
    G = sum(sizes)
    valid = torch.ones(B, T, dtype=torch.bool)
    tgt = (torch.rand(G, T, generator=g) > 0.5).float()
    layers = [(torch.randn(B, Q, K1, generator=g).to(DEV).requires_grad_(True), torch.randn(B, Q, T, generator=g).to(DEV).requires_grad_(True))]
    weight = torch.ones(K1)
    for bad in (K1, K1 + 1000, -1):
        ids = torch.randint(1, K1, (G,), generator=g)
        ids[3] = bad
        vals, q_of, failed = losses.device_criterion(layers, valid.to(DEV), sizes, ids.to(DEV), tgt.to(DEV), None, 1.0, weight, (2.0, 5.0, 5.0))
        torch.cuda.synchronize()
        assert bool(failed) and bool((q_of[0, 3:6] == -1).all()) and bool((q_of[0, :3] >= 0).all())
    # 1,400 pairs x 9 queries = 12,600 rows > the 12,224 the fused kernel's table holds: MaskVRD.criterion falls through
    model, mc, _ = build()
    Bb = 1400
    preds = {"pred_logits": torch.randn(Bb, Q, K1, generator=g).to(DEV), "pred_masks": torch.randn(Bb, Q, T, generator=g).to(DEV),
             "output_mask": torch.ones(Bb, 1, T, dtype=torch.bool, device=DEV),
             "aux_outputs": [{"pred_logits": torch.randn(Bb, Q, K1, generator=g).to(DEV), "pred_masks": torch.randn(Bb, Q, T, generator=g).to(DEV)}
                             for _ in range(3)]}
    data = {"preds_list": [torch.randint(1, K1, (1,), generator=g) for _ in range(Bb)],
            "masks_list": [(torch.rand(1, T, generator=g) > 0.5).float() for _ in range(Bb)],
            "segs_list": [torch.tensor([[3, 40]]) for _ in range(Bb)]}
    with torch.no_grad():
        big = model.criterion(preds, data)
    assert all(bool(torch.isfinite(v)) for v in big.values()) and "total_loss" in big


def test_device_matching_gives_the_host_matchings():
    """MaskVRD.bipartite_match with the device assignment vs the scipy path on the training batch's predictions."""
    model, mc, _ = build()
    lens, x, m, data = train_batch(mc, c_in(mc), device=DEV)
    with torch.no_grad():
        out = model.eval()._mask_vrd(x.to(DEV), m.to(DEV), with_aux=False)
        args = (out["pred_logits"], data["preds_list"], out["pred_masks"], data["masks_list"], data["segs_list"])
        model.device_matching = True
        a, _ = model.bipartite_match(*args, _mask=out["output_mask"])
        model.device_matching = False
        b, _ = model.bipartite_match(*args, _mask=out["output_mask"])
    for n, ((ai, aj), (bi, bj)) in enumerate(zip(a, b)):
        if lens[n] >= 16:          # shorter pairs price their queries identically to ~1e-5: ties
            assert ai.tolist() == bi.tolist() and aj.tolist() == bj.tolist(), n
        assert sorted(aj.tolist()) == sorted(bj.tolist()) and len(set(ai.tolist())) == len(ai)


def test_criterion_on_device_equals_the_host_round_trip():
    """MaskVRD._criterion_on_device (losses straight from the device assignment, relations in batch order) against
    bipartite_match + loss (indices to the host and back, matched rows in the reference's per-pair query order): same
    keys in the same order, same values, same gradients -- on the 24-pair training batch with all auxiliary layers."""
    import importlib.util
    from vrdone_amd import configs, synth
    from vrdone_amd.models.blocks import AffineDropPath
    from vrdone_amd.models.maskvrd import MaskVRD
    spec = importlib.util.spec_from_file_location("train_step", os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "train_step.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    cfg = configs.model_config("vidvrd")
    model = synth.load_synthetic_weights(MaskVRD(cfg, device=DEV)).to(DEV).train()
    for mod in model.modules():
        if isinstance(mod, AffineDropPath):
            mod.drop_prob = 0.0
    data = ts.synthetic_batch(cfg, configs.input_channels(cfg), DEV, seed=5)
    runs = []
    for on in (False, True):
        model.device_criterion = on
        model.zero_grad(set_to_none=True)
        losses = model(data)
        losses["total_loss"].backward()
        runs.append(({k: float(v.detach()) for k, v in losses.items()}, {k: p.grad.clone() for k, p in model.named_parameters()}))
    (host, g_host), (devc, g_dev) = runs
    assert list(host) == list(devc) and len(host) == 3 * cfg["predictor"]["num_layers"] + 1
    for k in host:
        assert abs(host[k] - devc[k]) <= 2e-6 * abs(host[k]) + 1e-7, (k, host[k], devc[k])
    floor = 1e-6 * max(float(g.abs().max()) for g in g_host.values())
    for k, g in g_host.items():
        assert float((g_dev[k] - g).abs().max()) <= 2e-4 * float(g.abs().max()) + floor, k


def test_ema_update_is_one_launch_and_bit_identical():
    """vrdone_amd.ema.ModelEma.update (vrd_ema_update over a pointer table) vs the reference's per-tensor expression
    decay * e + (1 - decay) * m (utils/train_utils.py:21-29), three updates in a row."""
    import copy
    from vrdone_amd import _hip
    from vrdone_amd.ema import ModelEma
    model, mc, _ = build()
    ema = ModelEma(model, decay=0.999)
    ref = copy.deepcopy(model).eval()
    g = torch.Generator(device=DEV).manual_seed(0)
    for step in range(3):
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn(p.shape, device=DEV, generator=g) * 0.01)
            for e, mv in zip(ref.state_dict().values(), model.state_dict().values()):
                e.copy_(0.999 * e + (1.0 - 0.999) * mv)
        _hip.prof_enable(True)
        _hip.prof_reset()
        ema.update(model)
        torch.cuda.synchronize()
        launches = _hip.prof_read()["backward"]["launches"]
        _hip.prof_enable(False)
        assert launches == 1
        for (k, a), b in zip(ema.module.state_dict().items(), ref.state_dict().values()):
            assert torch.equal(a, b), (step, k)
    ema.set(model)
    for a, b in zip(ema.module.state_dict().values(), model.state_dict().values()):
        assert torch.equal(a, b)


def test_ema_module_forward_follows_its_updates():
    """The update kernel writes the averaged weights through raw pointers.  ops caches derived operands on the parameter
    objects (split bf16 weights, packed k = 3 weights) keyed on their version counters: a forward of ema.module AFTER an
    update has to see the new weights -- i.e. equal a fresh model loaded from ema.module.state_dict()."""
    from vrdone_amd import synth
    from vrdone_amd.ema import ModelEma
    from vrdone_amd.models.maskvrd import MaskVRD
    model, mc, _ = build()
    ema = ModelEma(model, decay=0.5)
    lens, x, m, _ = train_batch(mc, c_in(mc), device=DEV)
    x, m = x.to(DEV), m.to(DEV)
    with torch.no_grad():
        before = ema.module._mask_vrd(x, m, with_aux=False)["pred_logits"].clone()      # fills the derived-operand caches
        g = torch.Generator(device=DEV).manual_seed(5)
        for p in model.parameters():
            p.add_(torch.randn(p.shape, device=DEV, generator=g) * 0.05)
        ema.update(model)
        after = ema.module._mask_vrd(x, m, with_aux=False)["pred_logits"]
        fresh = MaskVRD(mc, device=DEV).to(DEV).eval()
        fresh.load_state_dict(ema.module.state_dict())
        want = fresh._mask_vrd(x, m, with_aux=False)["pred_logits"]
    assert float((after - before).abs().max()) > 1e-3          # the update is visible at all
    assert torch.equal(after, want)


def test_training_sample_from_annotation_files_takes_a_step(tmp_path):
    """The training-side data path end to end: annotation + ground-truth feature files -> cache entry ->
    sample lists (proposals.load_train_video / train_getitem, pinned against the reference dataloader on the CPU) ->
    `forward_training` + backward: a finite loss and a finite gradient for every parameter."""
    import random
    from oracle import proposal as P
    from vrdone_amd.models.maskvrd import MaskVRD
    from vrdone_amd.proposals import load_train_video, train_getitem
    mc, _, keys = load_case("vidvrd")
    anno_dir, feat_dir, ent, pred = P.write_synth_train_files(str(tmp_path), n_visual=mc["visual_dim"])
    video = load_train_video(f"{anno_dir}/vid0.json", f"{feat_dir}/vid0.pkl", ent, pred)
    random.seed(0)
    sample = train_getitem(video, 1, mc["max_seq_len"])
    assert len(sample["so_features_list"]) >= 3
    model = MaskVRD(mc, device=DEV)
    model.load_state_dict(O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"]))
    model = model.to(DEV).train()
    loss = model({k: [t.to(DEV) for t in v] for k, v in sample.items()})
    assert np.isfinite(float(loss["total_loss"].detach()))
    loss["total_loss"].backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.requires_grad)
