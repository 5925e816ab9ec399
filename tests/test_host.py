"""CPU-side checks of the drop-in boundary: parameter tree / checkpoint layout, config
handling, C-ABI exports.  No kernels are launched here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO, load_case
from oracle import vrd_oracle as O


@pytest.mark.parametrize("name", ["vidvrd", "vidor_x", "vidor_local"])
def test_state_dict_layout_matches_reference(name):
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, keys = load_case(name)
    model = MaskVRD(mc, device="cpu")
    sd = model.state_dict()
    got = [(k, tuple(v.shape)) for k, v in sd.items()]
    assert got == keys          # names, shapes AND order (ModelEma zips two state_dicts)
    synth = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    model.load_state_dict(synth, strict=True)
    assert model.max_div_factor == O.max_div_factor(mc)


def test_weight_decay_grouping_contract():
    """utils/train_utils.py:44-65 classifies parameters by module type imported from models.blocks
    and by the suffix of the parameter name; every parameter must land in exactly one group."""
    from vrdone_amd.models.maskvrd import MaskVRD
    from vrdone_amd.models.blocks import MaskedConv1D, Scale, AffineDropPath, LayerNorm
    mc, _, _ = load_case("vidvrd")
    model = MaskVRD(mc, device="cpu")
    decay, no_decay = set(), set()
    for mn, m in model.named_modules():
        for pn, _ in m.named_parameters():
            full = f"{mn}.{pn}" if mn else pn
            if pn.endswith("bias"):
                no_decay.add(full)
            elif pn.endswith("weight") and isinstance(m, (torch.nn.Linear, torch.nn.Conv1d, MaskedConv1D)):
                decay.add(full)
            elif pn.endswith("weight") and isinstance(m, (LayerNorm, torch.nn.GroupNorm)):
                no_decay.add(full)
            elif pn.endswith("scale") and isinstance(m, (Scale, AffineDropPath)):
                no_decay.add(full)
    params = dict(model.named_parameters())
    assert not (decay & no_decay)
    left = params.keys() - (decay | no_decay)
    assert left == {"predictor.query_embed.weight"}      # the reference's only leftover (nn.Embedding)
    # initial values the training recipe relies on
    assert float(model.backbone.stem[0].drop_path_attn.scale.detach().mean()) == pytest.approx(1e-4)
    assert float(model.predictor.class_embed.bias.detach().mean()) == pytest.approx(-4.59512, abs=1e-4)
    assert all(float(p.detach().abs().sum()) == 0.0 for n, p in params.items()
               if n.endswith("bias") and p.dim() == 1 and "class_embed" not in n)


def test_no_cpu_fallback_and_no_oracle_import():
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, ic, _ = load_case("vidvrd")
    model = MaskVRD(mc, device="cpu").eval()
    x, m = O.synth_pairs(1, 2069, 96)
    with pytest.raises(RuntimeError, match="HIP device only"):
        model._mask_vrd(x, m)
    model.train()
    with torch.enable_grad(), pytest.raises(RuntimeError, match="HIP device only"):      # a training step: same device rule
        model({"so_features_list": [x[0]]})
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device only"):   # loss values: same network path
        model({"so_features_list": [x[0]]})
    # the product never imports the oracle
    for root, _, files in os.walk(os.path.join(REPO, "vrdone_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "vrdone_hip.h")).read()
    declared = set(re.findall(r"\b(vrd_[a-z0-9_]+)\s*\(", header))
    declared -= {"vrd_act", "vrd_kernel_id"}
    assert len(declared) >= 14
    path = os.path.join(REPO, "vrdone_amd", "csrc", "libvrdone_hip.so")
    assert os.path.exists(path), "build the extension first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(path)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vrdone_hip.h but not exported"
    from vrdone_amd import _hip
    assert set(_hip._SIGNATURES) == declared
    assert _hip.lib.vrd_abi_version() == _hip.ABI_VERSION


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype of include/vrdone_hip.h against its ctypes binding: the same number of parameters, pointers bound as
    pointers, 64-bit integers as c_int64, ints / enums as c_int, floats as c_float / c_double (a binding that drifts from
    the header passes garbage in registers -- nothing else would notice before a kernel faults)."""
    from vrdone_amd import _hip
    header = open(os.path.join(REPO, "include", "vrdone_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = dict(re.findall(r"\b(?:int|const char\*|void)\s+(vrd_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header))
    assert set(_hip._SIGNATURES) <= set(protos), sorted(set(_hip._SIGNATURES) - set(protos))

    def kind(param):
        param = " ".join(param.split())
        if "*" in param:
            return "ptr"
        base = param.rsplit(" ", 1)[0] if " " in param else param
        return {"int64_t": "i64", "unsigned long long": "i64", "uint64_t": "i64", "int": "i32", "int32_t": "i32", "unsigned": "i32",
                "float": "f32", "double": "f64"}[base.replace("const ", "")]

    def ckind(t):
        if t in (ctypes.c_int64, ctypes.c_longlong, ctypes.c_uint64, ctypes.c_ulonglong):
            return "i64"
        if t in (ctypes.c_int, ctypes.c_int32, ctypes.c_uint):
            return "i32"
        if t is ctypes.c_float:
            return "f32"
        if t is ctypes.c_double:
            return "f64"
        return "ptr"                      # c_void_p, c_char_p, POINTER(...)

    for name, (_, argtypes) in _hip._SIGNATURES.items():
        params = [p for p in protos[name].split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(argtypes), f"{name}: header has {len(params)} parameters, the binding {len(argtypes)}"
        got = [ckind(t) for t in argtypes]
        want = [kind(p) for p in params]
        assert got == want, f"{name}: header {want} vs binding {got}"


def test_ctypes_struct_layout_matches_header():
    """Field order of the ctypes mirrors vs the C structs in the header."""
    from vrdone_amd import _hip
    header = open(os.path.join(REPO, "include", "vrdone_hip.h")).read()
    for cname, cls in (("vrd_gemm_args", _hip.GemmArgs), ("vrd_dwconv_ln_args", _hip.DwconvLnArgs),
                       ("vrd_criterion_args", _hip.CriterionArgs), ("vrd_criterion_grads", _hip.CriterionGrads),
                       ("vrd_split_job", _hip.SplitJob), ("vrd_pack_args", _hip.PackArgs), ("vrd_gather_args", _hip.GatherArgs),
                       ("vrd_row_segs", _hip.RowSegs), ("vrd_conv_ln_args", _hip.ConvLnArgs)):
        body = re.search(r"typedef struct \{([^}]*)\} " + cname, header).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.sub(r"\[\w+\]", "", part.strip().split()[-1].lstrip("*")))
        assert names == [f[0] for f in cls._fields_], cname


def test_validation_errors_are_reported_without_a_gpu():
    from vrdone_amd import _hip
    a = _hip.GemmArgs()
    a.taps = 2
    rc = _hip.lib.vrd_gemm(ctypes.byref(a), None)
    assert rc != 0 and b"vrd_gemm" in _hip.lib.vrd_last_error()


@pytest.mark.parametrize("name", ["vidvrd", "vidor_x", "vidor_local"])
def test_python_configs_equal_the_reference_yaml(name):
    from vrdone_amd import configs
    mc, ic, _ = load_case(name)
    assert configs.model_config(name) == mc
    assert configs.inference_config(name) == ic
    assert configs.input_channels(mc) == {"vidvrd": 2069, "vidor_x": 3093, "vidor_local": 2069}[name]


def test_synthetic_weights_match_the_oracle_recipe():
    from vrdone_amd import configs, synth
    from vrdone_amd.models.maskvrd import MaskVRD
    mc, _, keys = load_case("vidvrd")
    model = synth.load_synthetic_weights(MaskVRD(configs.model_config("vidvrd"), device="cpu"))
    want = O.synth_state_dict(keys, eos_coef=mc["loss_coeff_dict"]["eos_coef"])
    for k, v in model.state_dict().items():
        assert torch.equal(v, want[k]), k
    x1, m1 = synth.synth_pairs(3, 7, 12, [12, 5, 1], seed=9)
    x2, m2 = O.synth_pairs(3, 7, 12, [12, 5, 1], seed=9)
    assert torch.equal(x1, x2) and torch.equal(m1, m2)


def test_dropin_launcher_takes_precedence_over_the_checkouts_models(tmp_path):
    """a mock reference tree whose own models/ package must NOT be the one imported when its
    eval.py runs through dropin/run.py (plain `python eval.py` with PYTHONPATH would import the checkout's package:
    the script directory precedes PYTHONPATH)."""
    import subprocess
    import sys
    ref = tmp_path / "vrdone"
    (ref / "models").mkdir(parents=True)
    (ref / "utils").mkdir()
    (ref / "models" / "__init__.py").write_text("")
    (ref / "models" / "maskvrd.py").write_text("class MaskVRD:\n    pass\n")
    (ref / "models" / "blocks.py").write_text("class MaskedConv1D: pass\nclass Scale: pass\nclass AffineDropPath: pass\nclass LayerNorm: pass\n")
    (ref / "utils" / "__init__.py").write_text("")
    (ref / "utils" / "train_utils.py").write_text("from models.blocks import MaskedConv1D, Scale, AffineDropPath, LayerNorm\nKIND = LayerNorm.__module__\n")
    (ref / "eval.py").write_text(
        "import sys\nfrom models.maskvrd import MaskVRD\nfrom utils import train_utils\n"
        "print('MODEL', MaskVRD.__module__)\nprint('BLOCKS', train_utils.KIND)\nprint('ARGS', sys.argv[1:])\n")
    run = os.path.join(REPO, "dropin", "run.py")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    out = subprocess.run([sys.executable, run, "eval.py", "--cfg_path", "x.yaml"], cwd=ref, env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "MODEL vrdone_amd.models.maskvrd" in out.stdout
    assert "BLOCKS vrdone_amd.models.blocks" in out.stdout
    assert "ARGS ['--cfg_path', 'x.yaml']" in out.stdout
    # the failure mode the launcher exists for
    env["PYTHONPATH"] = os.path.join(REPO, "dropin") + os.pathsep + REPO
    plain = subprocess.run([sys.executable, "eval.py"], cwd=ref, env=env, capture_output=True, text=True, timeout=300)
    assert "MODEL models.maskvrd" in plain.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="needs the reference checkout (build container only)")
def test_reference_optimizer_and_ema_accept_the_dropin():
    """utils/train_utils.py of the REFERENCE (build_optimizer's weight-decay grouping, ModelEma's state_dict zip) run
    against the drop-in model, in a child process with dropin/ ahead of the checkout (nothing of the reference is
    imported into this test process)."""
    import subprocess
    import sys
    code = (
        "import sys, os\n"
        f"sys.path[:0] = [{os.path.join(REPO, 'dropin')!r}, {REPO!r}, '/root/reference']\n"
        "import yaml, torch\n"
        "from models.maskvrd import MaskVRD\n"
        "assert MaskVRD.__module__.startswith('vrdone_amd.')\n"
        "from utils.train_utils import build_optimizer, ModelEma\n"
        "cfg = yaml.safe_load(open('/root/reference/configs/vidvrd.yaml'))\n"
        "model = MaskVRD(cfg['model_config'], device='cpu')\n"
        "opt = build_optimizer(model, cfg['training_config'])\n"
        "n = [len(g['params']) for g in opt.param_groups]\n"
        "ema = ModelEma(model)\n"
        "with torch.no_grad():\n"
        "    for p in model.parameters(): p.add_(1.0)\n"
        "ema.update(model)\n"
        "k = 'backbone.stem.0.ln1.bias'\n"
        "d = float((ema.module.state_dict()[k] - model.state_dict()[k]).abs().max())\n"
        "print('GROUPS', n, 'EMA_LAG', round(d, 4))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "GROUPS [176, 345]" in out.stdout and "EMA_LAG 0.999" in out.stdout, out.stdout


def test_synthetic_raw_video_has_the_dataloader_layout():
    """synth_raw_video hands out what `_prepare_test` does: per-tracklet features / boxes of the tracklet's own length,
    half-open durations, and only pairs that share a frame."""
    from vrdone_amd import synth
    raw = synth.synth_raw_video(7, 16, 20, 40, seed=3, n_clip=8)
    spans = raw["traj_durations"]
    assert len(raw["visual_features_list"]) == len(raw["bboxes_list"]) == len(raw["clip_features_list"]) == 7
    for i in range(7):
        n = int(spans[i, 1] - spans[i, 0])
        assert raw["visual_features_list"][i].shape == (n, 16) and raw["bboxes_list"][i].shape == (n, 4)
        assert (raw["bboxes_list"][i][:, 2:] > raw["bboxes_list"][i][:, :2]).all()
    assert len(raw["sids"]) == len(raw["oids"]) > 0
    for s, o in zip(raw["sids"].tolist(), raw["oids"].tolist()):
        assert s != o and min(spans[s, 1], spans[o, 1]) > max(spans[s, 0], spans[o, 0])


def test_row_space_layout_and_filler_buckets():
    """Host side of the ragged row space (vrdone_amd/models/ragged.py): bucket offsets, the stacked [subject | object] groups,
    the last rows of the sequences whose k = 3 convs run flat, and the filler that rounds the row count to a multiple of 256."""
    import torch
    from vrdone_amd.models import ragged
    from vrdone_amd import _hip
    lay = ragged.Layout([(3, 32, True), (2, 96, True), (1, 48, False)])
    assert lay.segs == [(0, 3, 32), (96, 2, 96), (288, 1, 48)] and lay.rows == 336 and lay.rows_flat == 288
    assert lay.twice()[3:] == [(336, 3, 32), (432, 2, 96), (624, 1, 48)]
    assert lay.tail_rows(1, "cpu").tolist() == [31, 63, 95, 191, 287]
    assert lay.tail_rows(2, "cpu").tolist() == [31, 63, 95, 191, 287, 367, 399, 431, 527, 623]
    with __import__("pytest").raises(AssertionError):
        ragged.Layout([(1, 48, False), (3, 32, True)])            # the buckets with flat k = 3 convs come first
    for rows in (65536, 65536 + 32, 65536 + 48, 317344, 311104, 23792 * 4, 100000):
        fill = ragged.filler_buckets(rows, 288)
        assert (rows + sum(n * t for n, t in fill)) % 256 == 0 and all(t % 8 == 0 and 32 <= t <= 56 for _, t in fill), (rows, fill)
        assert sum(n * t for n, t in fill) < 512
    assert ragged.filler_buckets(4096 + 32, 288) == [] and ragged.filler_buckets(65536 + 48, 40) == []
    segs = _hip.RowSegs.of(lay.twice())
    assert segs.count == 6 and list(segs.row)[:6] == [0, 96, 288, 336, 432, 624] and list(segs.T)[:3] == [32, 96, 48]


def test_flat_k3_conv_with_zeroed_tails_equals_the_per_sequence_conv_on_valid_rows():
    """The invariant behind the row-space form's dense k = 3 convs (vrdone_amd/models/ragged.py): sequences of different
    padded lengths back to back, the last row of every sequence zeroed in the conv's INPUT, one Conv1d(padding=1) over all
    rows -- equals the reference's per-sequence Conv1d(padding=1) * mask (models/blocks.py:99-113) on every row whose mask is
    set, provided the last TWO frames of every sequence are padded ones; with only one padded frame the last valid frame
    differs (which is why such pairs are bucketed apart).  float64, plain torch."""
    import torch
    g = torch.Generator().manual_seed(5)
    C, N = 6, 5
    w, b = torch.randn(N, C, 3, generator=g, dtype=torch.float64), torch.randn(N, generator=g, dtype=torch.float64)
    conv = lambda x: torch.nn.functional.conv1d(x, w, b, padding=1)          # noqa: E731  x (B, C, T)
    seqs = [(32, 30), (32, 7), (64, 62), (96, 40), (32, 2)]                   # (padded length, valid frames <= padded - 2)
    xs = [torch.randn(1, C, T, generator=g, dtype=torch.float64) + 3.0 for T, _ in seqs]     # padded frames hold junk, like LN(0) = beta
    want = [conv(x)[0, :, :L] for x, (T, L) in zip(xs, seqs)]
    flat = torch.cat([x.clone() for x in xs], dim=2)
    at = 0
    for T, _ in seqs:
        at += T
        flat[0, :, at - 1] = 0.0                                              # the sequence's last row
    got, at = conv(flat)[0], 0
    for (T, L), wnt in zip(seqs, want):
        assert torch.allclose(got[:, at:at + L], wnt, rtol=0, atol=1e-12)
        at += T
    # one padded frame only: the last valid frame reads the zeroed row instead of the padded frame's value
    T, L = 32, 31
    x = torch.randn(1, C, T, generator=g, dtype=torch.float64) + 3.0
    z = x.clone()
    z[0, :, T - 1] = 0.0
    assert not torch.allclose(conv(z)[0, :, L - 1], conv(x)[0, :, L - 1], atol=1e-6)


def test_every_host_side_switch_is_documented():
    """Every VRDONE_* environment switch the host code reads is listed in INTEGRATION.md's table of switches."""
    import glob
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    seen = set()
    for f in glob.glob(os.path.join(REPO, "vrdone_amd", "**", "*.py"), recursive=True):
        seen |= set(re.findall(r"VRDONE_[A-Z0-9_]+", open(f).read()))
    assert len(seen) >= 12
    missing = sorted(s for s in seen if s not in text)
    assert not missing, f"switches read by the host code but not in INTEGRATION.md: {missing}"
