"""CPU-side checks of the C-ABI library: it loads and exports every symbol that
include/dge_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dge_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dge_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import dge_amd
    from dge_amd import _lib
    so = _lib.LIB_PATH
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dge_hip.h but not exported"
    # the ctypes binding table covers the header (dge_last_error is bound separately)
    bound = set(_lib.SIGNATURES) | {"dge_last_error", "dge_last_kernel"}
    assert set(names) <= bound, sorted(set(names) - bound)


def test_product_path_fails_loudly_without_gpu():
    import torch
    import dge_amd
    from dge_amd._lib import DgeError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    G = dge_amd.StyleGAN2Generator(8, fmaps_base=256, fmaps_max=32)
    with pytest.raises(DgeError):
        G.synthesis(torch.zeros(1, G.num_layers, 512))


def test_invalid_arguments_mirror_reference_errors():
    import torch
    import dge_amd
    with pytest.raises(ValueError):
        dge_amd.StyleGAN2Generator(100)                    # stylegan2_generator.py:99-101
    G = dge_amd.StyleGAN2Generator(8, fmaps_base=256, fmaps_max=32)
    with pytest.raises(ValueError):
        G.synthesis(torch.zeros(1, 3, 512))                # :493-498
    with pytest.raises(ValueError):
        G.mapping(torch.zeros(1, 7))                       # :247-251
