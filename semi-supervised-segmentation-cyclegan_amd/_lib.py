"""ctypes binding of libsscg.so (the C ABI declared in include/sscg.h).

The library is the product: there is no Python/torch fallback for any kernel.  If the shared object
is missing or does not export a declared symbol, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSCG_LIB") or os.path.join(_HERE, "libsscg.so")   # SSCG_LIB: kernel-ablation builds (tools/)

ABI_VERSION = 17

F32, BF16, BF16X3 = 0, 1, 2     # SSCG_F32 / SSCG_BF16 / SSCG_BF16X3 (split weight operand)

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
PAD_ZEROS, PAD_REFLECT = 0, 1


class ConvDesc(C.Structure):
    """Mirror of `sscg_conv_desc` (include/sscg.h)."""

    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("K", C.c_int32), ("R", C.c_int32), ("S", C.c_int32),
        ("P", C.c_int32), ("Q", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("pad_mode", C.c_int32), ("act", C.c_int32), ("slope", C.c_float),
        ("x_dtype", C.c_int32), ("w_dtype", C.c_int32), ("y_dtype", C.c_int32), ("precision", C.c_int32),
        ("tuning", C.c_int32), ("w_plane", C.c_int64), ("wgrad_tuning", C.c_int32),
    ]


_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float
_sz = C.c_size_t
_dp = C.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every symbol of include/sscg.h (tests/test_abi.py checks)
SIGNATURES = {
    "sscg_abi_version": (_i, []),
    "sscg_set_dry_run": (_i, [_i]),
    "sscg_resize_channels": (_i, [_p, _p, _i64, _i, _i, _p]),
    "sscg_conv2d_fwd_workspace": (_sz, [_dp]),
    "sscg_conv2d_fwd": (_i, [_dp, _p, _p, _p, _p, _p, _sz, _p]),
    "sscg_conv2d_fwd_stats_bytes": (_sz, [_dp, _i, _i64]),
    "sscg_conv2d_fwd_stats_workspace": (_sz, [_dp]),
    "sscg_conv2d_fwd_stats": (_i, [_dp, _p, _p, _p, _p, _i, _i64, _p, _sz, _p, _sz, _p]),
    "sscg_norm_stats_from_conv": (_i, [_dp, _p, _i, _i64, _f, _p, _p, _p, _p, _f, _p]),
    "sscg_conv2d_front_applies": (_i, [_dp, _i]),
    "sscg_conv2d_front_fwd": (_i, [_dp, _p, _i, _p, _p, _f, _p, _p, _p, _p, _i, _i64, _p, _sz, _p]),
    "sscg_conv2d_dgrad_workspace": (_sz, [_dp]),
    "sscg_conv2d_dgrad": (_i, [_dp, _p, _p, _p, _p, _i, _f, _p, _sz, _p]),
    "sscg_conv2d_dgrad_bsums_bytes": (_sz, [_dp, _i, _i64]),
    "sscg_conv2d_dgrad_bsums": (_i, [_dp, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _f, _p, _sz, _p, _sz, _p]),
    "sscg_conv2d_dgrad_add_applies": (_i, [_dp]),
    "sscg_conv2d_dgrad_add": (_i, [_dp, _p, _p, _p, _p, _p, _sz, _p]),
    "sscg_norm_bwd_from_sums": (_i, [_dp, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _i, _p, _sz, _p]),
    "sscg_conv2d_wgrad_workspace": (_sz, [_dp]),
    "sscg_conv2d_wgrad": (_i, [_dp, _p, _p, _p, _f, _p, _sz, _p]),
    "sscg_weight_krsc_to_crsk": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "sscg_weight_krsc_to_crsk_batch": (_i, [_p, _i, _i, _p]),
    "sscg_split3": (_i, [_p, _p, _i64, _i64, _p]),
    "sscg_conv2d_split_applies": (_i, [_dp, _i]),
    "sscg_cast": (_i, [_p, _i, _p, _i, _i64, _p]),
    "sscg_colsum_workspace": (_sz, [_i64, _i]),
    "sscg_colsum": (_i, [_p, _i, _p, _i64, _i, _f, _p, _sz, _p]),
    "sscg_norm_stats_workspace": (_sz, [_i, _i64, _i]),
    "sscg_norm_stats": (_i, [_p, _i, _i, _i64, _i, _f, _p, _p, _p, _p, _f, _p, _sz, _p]),
    "sscg_norm_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _p]),
    "sscg_rstd_from_var": (_i, [_p, _p, _i, _f, _p]),
    "sscg_norm_bwd_workspace": (_sz, [_i, _i64, _i]),
    "sscg_norm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _i, _p, _sz, _p]),
    "sscg_norm_head_applies": (_i, [_i]),
    "sscg_norm_head_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _f, _p]),
    "sscg_norm_head_bwd_workspace": (_sz, [_i, _i64, _i]),
    "sscg_norm_head_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _i, _p, _sz, _p]),
    "sscg_act_fwd": (_i, [_p, _p, _i, _i64, _i, _f, _p]),
    "sscg_act_bwd": (_i, [_p, _p, _p, _i, _i64, _i, _f, _p]),
    "sscg_add": (_i, [_p, _p, _p, _i, _i64, _p]),
    "sscg_dropout": (_i, [_p, _p, _i, _i64, _f, C.c_uint64, _p]),
    "sscg_gauss_noise": (_i, [_p, _p, _i64, _f, C.c_uint64, _p]),
    "sscg_maxpool2x2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "sscg_maxpool2x2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "sscg_maxpool3x3s2_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_maxpool3x3s2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_upsample_bilinear_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_upsample_bilinear_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_reflect_pad": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_reflect_pad_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "sscg_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "sscg_softmax_fwd": (_i, [_p, _p, _i64, _i, _p]),
    "sscg_softmax_bwd": (_i, [_p, _p, _p, _i64, _i, _p]),
    "sscg_argmax_onehot": (_i, [_p, _p, _p, _i64, _i, _p]),
    "sscg_label_onehot": (_i, [_p, _p, _i64, _i, _p]),
    "sscg_confusion_hist": (_i, [_p, _p, _i64, _i, _p, _p]),
    "sscg_image_u8_to_f32": (_i, [_p, _p, _i64, _i, _p, _p, _p]),
    "sscg_label_lut": (_i, [_p, _p, _i64, _p, _p]),
    "sscg_loss_workspace": (_sz, [_i64]),
    "sscg_ce_fwd": (_i, [_p, _p, _i64, _i, _p, _p, _p, _sz, _p]),
    "sscg_ce_bwd": (_i, [_p, _p, _i64, _i, _p, _f, _p, _p, _p]),
    "sscg_upsample_head_workspace": (_sz, [_i, _i, _i]),
    "sscg_upsample_head_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "sscg_upsample_head_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sscg_mse_const_fwd": (_i, [_p, _i64, _f, _p, _p, _sz, _p]),
    "sscg_mse_const_bwd": (_i, [_p, _i64, _f, _p, _f, _p, _p]),
    "sscg_mse_fwd": (_i, [_p, _p, _i64, _p, _p, _sz, _p]),
    "sscg_mse_bwd": (_i, [_p, _p, _i64, _p, _f, _p, _p, _p]),
    "sscg_l1_fwd": (_i, [_p, _p, _i64, _p, _p, _sz, _p]),
    "sscg_l1_bwd": (_i, [_p, _p, _i64, _p, _f, _p, _p]),
    "sscg_weighted_sum": (_i, [C.POINTER(_p), C.POINTER(_f), _i, _p, _p]),
    "sscg_adam_step": (_i, [_p, _p, _p, _p, _p, _i, _i64, C.c_double, C.c_double, C.c_double, C.c_double, _i, _f, _p]),
    "sscg_fill": (_i, [_p, _i64, _f, _p]),
}


class SscgError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsscg.so not found at %s - build it with `make -C %s/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no fallback path." % (LIB_PATH, _HERE))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.sscg_abi_version()
    if v != ABI_VERSION:
        raise ImportError("libsscg.so ABI version %d != binding version %d" % (v, ABI_VERSION))
    if os.environ.get("SSCG_TRACE"):
        return _Traced(lib)
    if os.environ.get("SSCG_RACECHECK"):        # debug: log every launch into the stream-ordering checker (tools/racecheck.py)
        racecheck = dev_tool("racecheck")
        racecheck.install()
        return racecheck._Checked(lib)
    if os.environ.get("SSCG_FUZZ"):             # debug: schedule fuzzer (racecheck.fuzz(seed) switches it on)
        return dev_tool("racecheck")._Fuzzed(lib)
    return lib


def dev_tool(name):
    """A development tool from the repository's tools/ directory (not part of the library: the stream-ordering checker and schedule
    fuzzer wrap the ctypes handle when SSCG_RACECHECK / SSCG_FUZZ ask for them), loaded once under its plain module name."""
    import importlib.util
    import sys
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(os.path.dirname(_HERE), "tools", name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
    if spec is None or not os.path.exists(path):
        raise ImportError("development tool %s not found at %s" % (name, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Traced:
    """SSCG_TRACE=1: print every entry point before it runs and synchronise after it (fault localisation)."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def call(*a):
            import sys
            import torch
            sys.stderr.write("[sscg] %s%s\n" % (name, tuple(x if isinstance(x, (int, float)) else "." for x in a)))
            sys.stderr.flush()
            r = fn(*a)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            return r
        return call


lib = _load()

_ERRS = {-1: "SSCG_ERR_BAD_ARG", -2: "SSCG_ERR_UNSUPPORTED", -3: "SSCG_ERR_WORKSPACE"}


def check(rc, what):
    """Raise on a non-zero return code of an int-returning entry point."""
    if rc != 0:
        raise SscgError("%s failed: %s" % (what, _ERRS.get(rc, "hipError_t %d" % rc)))
