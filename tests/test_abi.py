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
    assert ctypes.sizeof(lib.ConvDesc) == 15 * 4
    names = [f[0] for f in lib.ConvDesc._fields_]
    assert names == ["N", "H", "W", "C", "K", "R", "S", "P", "Q", "stride", "pad", "dil", "pad_mode", "act", "slope"]


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
