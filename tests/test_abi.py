"""The C-ABI library loads on a GPU-less host and exports exactly the symbols include/sscg.h declares
(no compute call is made here)."""
import os
import re
import subprocess

from conftest import ROOT, load_sub


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sscg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sscg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = load_sub("_lib")
    syms = header_symbols()
    assert len(syms) >= 40
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sscg_[a-z0-9_]+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert exported <= set(syms), "exported but undeclared: %s" % sorted(exported - set(syms))
    assert set(lib.SIGNATURES) == set(syms), set(lib.SIGNATURES) ^ set(syms)
    assert lib.lib.sscg_abi_version() == lib.ABI_VERSION


def test_conv_desc_layout_matches_header():
    import ctypes
    lib = load_sub("_lib")
    assert ctypes.sizeof(lib.ConvDesc) == 96          # 20 four-byte members, the 8-byte plane stride at offset 80, wgrad_tuning, padding
    assert lib.ConvDesc.w_plane.offset == 80 and lib.ConvDesc.wgrad_tuning.offset == 88
    names = [f[0] for f in lib.ConvDesc._fields_]
    assert names == ["N", "H", "W", "C", "K", "R", "S", "P", "Q", "stride", "pad", "dil", "pad_mode", "act", "slope",
                     "x_dtype", "w_dtype", "y_dtype", "precision", "tuning", "w_plane", "wgrad_tuning"]
    # the header declares the same members in the same order
    src = open(os.path.join(ROOT, "include", "sscg.h")).read()
    body = src[src.index("typedef struct sscg_conv_desc {"):src.index("} sscg_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    members = [m.strip() for decl in re.findall(r"(?:int32_t|int64_t|float)\s+([^;]+);", body) for m in decl.split(",")]
    assert members == names


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the operators raise instead of computing on the host."""
    import pytest
    import torch
    F = load_sub("functional")
    lib = load_sub("_lib")
    with pytest.raises(lib.SscgError):
        F.conv2d(torch.zeros(1, 4, 8, 8), torch.zeros(4, 4, 3, 3))
    with pytest.raises(lib.SscgError):
        F.softmax2d(torch.zeros(1, 4, 8, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semi-supervised-segmentation-cyclegan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f


def test_argument_errors_are_returned_not_raised():
    """include/sscg.h: "return value: 0 = ok, <0 = library error (SSCG_ERR_*), >0 = hipError_t.  Never throws/aborts."
    Argument checks come before any HIP call, so they can be exercised without a GPU."""
    import ctypes as C
    L = load_sub("_lib")
    lib = L.lib
    BAD_ARG, UNSUPPORTED, WORKSPACE = -1, -2, -3
    d = L.ConvDesc(N=2, H=16, W=16, C=32, K=32, R=3, S=3, P=16, Q=16, stride=1, pad=1, dil=1, pad_mode=0, act=0, slope=0.0)
    assert lib.sscg_conv2d_fwd(C.byref(d), None, None, None, None, None, 0, None) == BAD_ARG          # null tensors
    assert lib.sscg_conv2d_fwd(None, None, None, None, None, None, 0, None) == BAD_ARG                # null descriptor
    assert lib.sscg_conv2d_wgrad(C.byref(d), None, None, None, 0.0, None, 0, None) == BAD_ARG
    big = L.ConvDesc(N=64, H=4096, W=4096, C=64, K=64, R=1, S=1, P=4096, Q=4096, stride=1, pad=0, dil=1, pad_mode=0, act=0, slope=0.0)
    one = C.c_void_p(16)                                                                              # never dereferenced
    assert lib.sscg_conv2d_wgrad(C.byref(big), one, one, one, 0.0, None, 0, None) == UNSUPPORTED      # >= 2^31 elements
    # a split plan needs its workspace: the bench-size DeepLab conv without one is refused before any launch
    dl = L.ConvDesc(N=8, H=33, W=33, C=256, K=256, R=3, S=3, P=33, Q=33, stride=1, pad=2, dil=2, pad_mode=0, act=0, slope=0.0)
    assert lib.sscg_conv2d_fwd_workspace(C.byref(dl)) > 0 and lib.sscg_conv2d_wgrad_workspace(C.byref(dl)) > 0
    assert lib.sscg_conv2d_fwd(C.byref(dl), one, one, None, one, None, 0, None) == WORKSPACE
    assert lib.sscg_conv2d_wgrad(C.byref(dl), one, one, one, 0.0, None, 0, None) == WORKSPACE
    # normalisation / class ops / optimiser / input pipeline
    assert lib.sscg_norm_stats(None, 0, 1, 10, 4, 1e-5, None, None, None, None, 0.1, None, 0, None) == BAD_ARG
    assert lib.sscg_norm_stats(one, 7, 1, 10, 4, 1e-5, one, one, None, None, 0.1, one, 1 << 20, None) == BAD_ARG   # unknown dtype code
    assert lib.sscg_norm_apply(one, one, one, one, None, None, one, 0, 1, 10, 4, 0, 0.0, None) == BAD_ARG  # gamma without beta
    assert lib.sscg_softmax_fwd(None, None, 10, 4, None) == BAD_ARG
    assert lib.sscg_confusion_hist(one, one, 10, 65, one, None) == BAD_ARG                            # C > 64
    assert lib.sscg_label_lut(one, one, 10, None, None) == BAD_ARG
    # dtype combinations the kernels do not cover are refused, not guessed: bf16 input with an fp32 weight operand
    mix = L.ConvDesc(N=2, H=16, W=16, C=64, K=64, R=3, S=3, P=16, Q=16, stride=1, pad=1, dil=1, pad_mode=0, act=0, slope=0.0,
                     x_dtype=L.BF16, w_dtype=L.F32, y_dtype=L.BF16, precision=0)
    assert lib.sscg_conv2d_fwd(C.byref(mix), one, one, None, one, None, 0, None) == UNSUPPORTED
    bad = L.ConvDesc(N=2, H=16, W=16, C=64, K=64, R=3, S=3, P=16, Q=16, stride=1, pad=1, dil=1, pad_mode=0, act=0, slope=0.0,
                     precision=3)
    assert lib.sscg_conv2d_fwd(C.byref(bad), one, one, None, one, None, 0, None) == BAD_ARG
    # the fused-statistics query answers 0 where the fusion does not apply (3-channel head) and > 0 where it does
    head = L.ConvDesc(N=2, H=16, W=16, C=64, K=3, R=3, S=3, P=16, Q=16, stride=1, pad=1, dil=1, pad_mode=0, act=0, slope=0.0)
    assert lib.sscg_conv2d_fwd_stats_bytes(C.byref(head), 2, 256) == 0
    assert lib.sscg_conv2d_fwd_stats_bytes(C.byref(dl), 1, 8 * 33 * 33) > 0
    assert lib.sscg_conv2d_fwd_stats_bytes(C.byref(dl), 1, 77) == 0                                   # G * L != N * P * Q
