"""ctypes binding of libvrdone_hip.so (C ABI declared in include/vrdone_hip.h).

There is no CPU fallback: importing this module without the built library, or calling an
op on a non-HIP tensor, raises.  Build with ``make -C vrdone_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VRDONE_HIP_LIB: another build of the same library (kernel experiments, A/B runs in one box session)
LIB_PATH = os.environ.get("VRDONE_HIP_LIB") or os.path.join(_HERE, "csrc", "libvrdone_hip.so")

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
PAIR_NONE, PAIR_BF16, PAIR_F16 = 0, 1, 2          # enum vrd_pair_format
F16_ACT_EXP = 4                                   # VRD_F16_ACT_EXP
ABSMAX_SCALE_FLOATS = 516                         # VRD_ABSMAX_SCALE_FLOATS: vrd_absmax_scale's buffer (factors, ticket, partial maxima)
(K_GEMM, K_LAYERNORM, K_DWCONV_LN, K_LOCAL_ATTN, K_ATTN_SMALL, K_ATTN_FLASH, K_POOL, K_MASK_HEAD,
 K_TRANSPOSE, K_POSTPROC, K_GEMM_X3, K_GEMM_X3_DMA, K_GEMM_X3_BIG, K_BACKWARD, K_COUNT) = range(15)   # enum vrd_kernel_id
KERNEL_NAMES = ["gemm_f32_mfma", "layernorm", "dwconv_ln", "local_attn", "attn_small", "attn_flash",
                "maxpool_mask", "mask_head", "transpose", "postprocess", "gemm_x3_mfma", "gemm_x3_dma",
                "gemm_x3_big", "backward"]
assert len(KERNEL_NAMES) == K_COUNT

c_f32p = C.c_void_p      # device pointers travel as plain integers
c_u8p = C.c_void_p
c_i32p = C.c_void_p


class CriterionArgs(C.Structure):                     # vrd_criterion_args
    _fields_ = [("logits", C.c_void_p * 4), ("masks", C.c_void_p * 4),
                ("n_layers", C.c_int32), ("B", C.c_int32), ("Q", C.c_int32), ("K1", C.c_int32), ("T", C.c_int32), ("G", C.c_int32),
                ("out_valid", C.c_void_p), ("tgt_ids", C.c_void_p), ("tgt_masks", C.c_void_p), ("owner", C.c_void_p),
                ("segs", C.c_void_p), ("scale_range", C.c_float), ("alpha", C.c_float), ("gamma", C.c_float),
                ("w_class", C.c_float), ("w_mask", C.c_float), ("w_dice", C.c_float)]


class CriterionGrads(C.Structure):                    # vrd_criterion_grads
    _fields_ = [("logits", C.c_void_p * 4), ("masks", C.c_void_p * 4)]


class GemmArgs(C.Structure):
    _fields_ = [("A", c_f32p), ("lda", C.c_int64), ("W", c_f32p), ("bias", c_f32p),
                ("C", c_f32p), ("ldc", C.c_int64),
                ("M", C.c_int64), ("N", C.c_int32), ("Cin", C.c_int32), ("taps", C.c_int32), ("T", C.c_int32),
                ("act", C.c_int32), ("row_mask", c_u8p), ("scale", c_f32p),
                ("res", c_f32p), ("ldres", C.c_int64), ("res_masked", C.c_int32),
                ("res2", c_f32p), ("ldres2", C.c_int64), ("W_split", C.c_void_p),
                ("a_pair_width", C.c_int32), ("c_pair", C.c_int32),
                ("row_blocks", C.c_void_p), ("row_blocks_active", C.c_void_p), ("row_block_seg_len", C.c_int32),
                ("split_fmt", C.c_int32), ("w_scale", c_f32p), ("a_scale", c_f32p)]


MAX_SEGS = 32                                     # VRD_MAX_SEGS


class RowSegs(C.Structure):
    """vrd_row_segs: a ragged row space as groups of sequences (n[i] sequences of T[i] frames from row row[i] on)"""
    _fields_ = [("count", C.c_int32), ("n", C.c_int32 * MAX_SEGS), ("T", C.c_int32 * MAX_SEGS), ("row", C.c_int64 * MAX_SEGS)]

    _made = {}

    @classmethod
    def of(cls, segs):
        """segs: [(first row, sequences, frames)]; the structs are read-only and kept (a network pass asks for the same few
        tables ~25 times)"""
        key = tuple(segs)
        s = cls._made.get(key)
        if s is None:
            assert 1 <= len(segs) <= MAX_SEGS
            if len(cls._made) > 4096:
                cls._made.clear()
            s = cls._made[key] = cls()
            s.count = len(segs)
            for i, (row, n, T) in enumerate(segs):
                s.row[i], s.n[i], s.T[i] = row, n, T
        return s


class DwconvLnArgs(C.Structure):
    _fields_ = [("x", c_f32p), ("ldx", C.c_int64), ("x_up", c_f32p), ("ldx_up", C.c_int64),
                ("B", C.c_int32), ("Tin", C.c_int32), ("C", C.c_int32), ("ksize", C.c_int32),
                ("stride", C.c_int32), ("group_in", C.c_int32),
                ("mask_out", c_u8p), ("n_out", C.c_int32),
                ("w", c_f32p * 3), ("bias", c_f32p * 3), ("gamma", c_f32p * 3), ("beta", c_f32p * 3),
                ("relu", C.c_int32 * 3), ("y", c_f32p * 3), ("ldy", C.c_int64 * 3), ("out_pair", C.c_int32 * 3),
                ("pre_gamma", c_f32p), ("pre_beta", c_f32p), ("packed", c_f32p * 3), ("segs", C.POINTER(RowSegs))]


class ConvLnArgs(C.Structure):
    _fields_ = [("x", c_f32p), ("ldx", C.c_int64), ("rows", C.c_int64),
                ("Cin", C.c_int32), ("taps", C.c_int32), ("T", C.c_int32), ("N", C.c_int32),
                ("w", c_f32p), ("bias", c_f32p), ("row_mask", c_u8p), ("gamma", c_f32p), ("beta", c_f32p),
                ("relu", C.c_int32), ("y", c_f32p), ("ldy", C.c_int64), ("out_pair", C.c_int32)]


class PackArgs(C.Structure):
    _fields_ = [("src", C.c_void_p), ("lens", C.c_void_p),
                ("P", C.c_int32), ("C_in", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("Cc", C.c_int32),
                ("S", C.c_int32), ("E", C.c_int32),
                ("vis", c_f32p), ("clip", c_f32p), ("so_box", c_f32p), ("ent", c_f32p), ("pair_wide", C.c_int32)]


class DwconvBwdArgs(C.Structure):
    _fields_ = [("dD", c_f32p * 3), ("lddd", C.c_int64 * 3), ("w", c_f32p * 3),
                ("n_out", C.c_int32), ("B", C.c_int32), ("Tin", C.c_int32), ("C", C.c_int32), ("ksize", C.c_int32),
                ("stride", C.c_int32), ("group_in", C.c_int32),
                ("mask_out", c_u8p), ("dx", c_f32p), ("lddx", C.c_int64), ("dx_up", c_f32p), ("lddx_up", C.c_int64)]


class BmmArgs(C.Structure):
    _fields_ = [("A", c_f32p), ("a_z0", C.c_int64), ("a_z1", C.c_int64), ("a_row", C.c_int64), ("a_col", C.c_int64),
                ("B", c_f32p), ("b_z0", C.c_int64), ("b_z1", C.c_int64), ("b_row", C.c_int64), ("b_col", C.c_int64),
                ("C", c_f32p), ("c_z0", C.c_int64), ("c_z1", C.c_int64), ("c_row", C.c_int64), ("c_col", C.c_int64),
                ("Z0", C.c_int32), ("Z1", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("alpha", C.c_float), ("accumulate", C.c_int32)]


class SplitJob(C.Structure):
    _fields_ = [("src", c_f32p), ("out", C.c_void_p), ("R", C.c_int32), ("Q", C.c_int32), ("taps", C.c_int32), ("fmt", C.c_int32),
                ("sr", C.c_int64), ("st", C.c_int64), ("sq", C.c_int64), ("scale", c_f32p)]


class GatherArgs(C.Structure):
    _fields_ = [("vis", c_f32p), ("clip", c_f32p), ("boxes", c_f32p), ("s_row", C.c_void_p), ("o_row", C.c_void_p),
                ("lens", C.c_void_p), ("P", C.c_int32), ("T", C.c_int32), ("V", C.c_int32), ("Cc", C.c_int32),
                ("stride", C.c_int32), ("w", C.c_float), ("h", C.c_float),
                ("out_vis", c_f32p), ("out_clip", c_f32p), ("out_so_box", c_f32p), ("out_ent", c_f32p),
                ("pair_wide", C.c_int32)]


class AssembleArgs(C.Structure):
    _fields_ = [("streams", c_f32p), ("snippets", c_f32p), ("stream_row", C.c_void_p), ("lens", C.c_void_p),
                ("P", C.c_int32), ("T", C.c_int32), ("D", C.c_int32), ("L", C.c_int32), ("piece", C.c_int32), ("reach", C.c_int32),
                ("out", c_f32p)]


_SIGNATURES = {
    "vrd_abi_version": (C.c_int, []),
    "vrd_last_error": (C.c_char_p, []),
    "vrd_f16_range_flag": (C.c_int, [C.POINTER(C.c_void_p)]),
    "vrd_prof_enable": (C.c_int, [C.c_int]),
    "vrd_prof_reset": (C.c_int, []),
    "vrd_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                C.POINTER(C.c_double)]),
    "vrd_prof_read_skipped": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "vrd_prof_select": (C.c_int, [C.c_uint64]),
    "vrd_bct_to_btc": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, C.c_int64, C.c_int, C.c_int, c_i32p,
                                 C.c_void_p]),
    "vrd_btc_to_bct": (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, c_f32p, C.c_void_p]),
    "vrd_pack_pairs": (C.c_int, [C.POINTER(PackArgs), C.c_void_p]),
    "vrd_gather_pairs": (C.c_int, [C.POINTER(GatherArgs), C.c_void_p]),
    "vrd_assemble_pairs": (C.c_int, [C.POINTER(AssembleArgs), C.c_void_p]),
    "vrd_split_weight": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, c_f32p,
                                   C.c_void_p]),
    "vrd_split_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vrd_gemm": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "vrd_gemm_batch": (C.c_int, [C.POINTER(GemmArgs), C.c_int, C.c_void_p]),
    "vrd_row_blocks": (C.c_int, [c_u8p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vrd_conv_ln": (C.c_int, [C.POINTER(ConvLnArgs), C.c_void_p]),
    "vrd_layernorm": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int, c_f32p, c_f32p, C.c_int,
                                c_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vrd_dwconv_ln": (C.c_int, [C.POINTER(DwconvLnArgs), C.c_void_p]),
    "vrd_local_attn": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_u8p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    "vrd_local_attn_segs": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_u8p, c_f32p, C.POINTER(RowSegs), C.c_int, C.c_int,
                                      C.c_int, c_f32p, C.c_int64, C.c_int, C.c_void_p]),
    "vrd_attention": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_u8p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vrd_attention_pair": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_u8p, c_u8p, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "vrd_maxpool_mask": (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, c_u8p, c_f32p, C.c_int64, c_u8p,
                                   C.c_void_p]),
    "vrd_mask_head": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_float, c_f32p, C.c_void_p]),
    "vrd_postprocess": (C.c_int, [c_f32p, c_f32p, c_i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f32p, c_i32p,
                                  c_i32p, c_i32p, C.c_void_p]),
    # ---- backward kernels (training step)
    "vrd_gemm_wgrad": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_u8p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                 c_f32p, C.c_void_p]),
    "vrd_gemm_wgrad_x3": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, c_u8p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                    c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "vrd_absmax_scale": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int, c_f32p, C.c_void_p]),
    "vrd_dwconv_wgrad": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, c_u8p, C.c_int64,
                                   C.c_int, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "vrd_colsum": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_u8p, c_f32p,
                             C.c_int64, C.c_int, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "vrd_rowcol_scale": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int, c_f32p, c_f32p, c_u8p, c_f32p, C.c_int64, C.c_int,
                                   c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_void_p]),
    "vrd_activation": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int64, C.c_void_p]),
    "vrd_layernorm_bwd": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int, c_f32p, c_f32p, C.c_int, c_f32p,
                                    C.c_int64, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p]),
    "vrd_dwconv_bwd": (C.c_int, [C.POINTER(DwconvBwdArgs), C.c_void_p]),
    "vrd_local_attn_bwd": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, c_u8p, c_f32p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "vrd_attn_bwd_probs": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int64, c_u8p, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, C.c_void_p]),
    "vrd_bmm": (C.c_int, [C.POINTER(BmmArgs), C.c_void_p]),
    "vrd_attention_bwd": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_u8p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p]),
    "vrd_attention_rows": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, c_f32p, C.c_int64, c_f32p, C.c_void_p]),
    "vrd_assign": (C.c_int, [c_f32p, C.c_int64, c_i32p, c_i32p, C.c_int, C.c_int, c_i32p, C.c_void_p]),
    "vrd_ema_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_i32p, c_i32p, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "vrd_attn_bwd_softmax": (C.c_int, [c_f32p, c_f32p, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vrd_criterion_costs": (C.c_int, [C.c_void_p, c_f32p, C.c_void_p]),
    "vrd_criterion_losses": (C.c_int, [C.c_void_p, c_i32p, c_f32p, C.c_float, c_f32p, C.c_void_p]),
    "vrd_criterion_backward": (C.c_int, [C.c_void_p, c_i32p, c_f32p, C.c_float, c_f32p, c_f32p, C.c_void_p, C.c_void_p]),
    "vrd_maxpool_bwd": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int, c_u8p, c_f32p, C.c_int64,
                                  C.c_void_p]),
}

ABI_VERSION = 34


class HipLibraryError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} is missing: the HIP extension was not built "
            "(run `make -C vrdone_amd/csrc`); there is no CPU fallback for this path")
    # PyTorch ships its own libamdhip64.so: it has to be in the process first, so that the library binds to that runtime.
    # Loaded the other way round, the process holds two HIP runtimes and the library's launches fail on torch's memory
    # ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.vrd_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"libvrdone_hip.so ABI {got} != expected {ABI_VERSION}: rebuild it")
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib.vrd_last_error().decode()}")


# tags of the f16 operand-range flag word (vrd_f16_range_flag)
RANGE_TAGS = {1: "boundary tensors (bct_to_btc / pack_pairs / gather_pairs)", 2: "layernorm", 4: "dwconv_ln", 8: "gemm outputs",
              16: "f32 rows split inside a gemm", 32: "attention outputs", 64: "other"}


def prof_enable(on=True):
    check(lib.vrd_prof_enable(1 if on else 0), "vrd_prof_enable")


def prof_select(families=None):
    """Record events only for the named kernel families (None = all)."""
    mask = (1 << 64) - 1 if families is None else sum(1 << KERNEL_NAMES.index(f) for f in families)
    check(lib.vrd_prof_select(mask), "vrd_prof_select")


def prof_reset():
    check(lib.vrd_prof_reset(), "vrd_prof_reset")


def prof_read():
    """{family: dict(ms=, launches=, flops=, bytes=, flops_skipped=)} of everything recorded since the last reset.
    flops = what the launches were sized for (2*M*N*K); flops_skipped = the part padding maps made the kernel skip."""
    out = {}
    for kid, name in enumerate(KERNEL_NAMES):
        ms, n, fl, by, sk = C.c_double(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
        check(lib.vrd_prof_read(kid, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "vrd_prof_read")
        check(lib.vrd_prof_read_skipped(kid, C.byref(sk)), "vrd_prof_read_skipped")
        out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value, "flops_skipped": sk.value}
    return out
