"""The C-ABI library builds for gfx950 here (no GPU needed), loads, and exports every symbol include/tgcn.h
declares; the Python binding declares the same set.  No compute call is made."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def built_lib():
    from textgcn_amd import build
    return build.build_lib()


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'tgcn.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tgcn_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert 'tgcn_spmm_csr_f32' in syms and 'tgcn_score_dense_f32' in syms and 'tgcn_topk_f32' in syms


def test_library_exports_every_declared_symbol(built_lib):
    handle = ctypes.CDLL(built_lib)
    for s in header_symbols():
        assert hasattr(handle, s), f'{s} declared in tgcn.h but not exported by libtgcn.so'


def test_binding_matches_header(built_lib):
    from textgcn_amd import _capi
    assert sorted(_capi.EXPORTED_SYMBOLS) == header_symbols()
    lib = _capi.lib()
    assert lib.tgcn_abi_version() == _capi.TGCN_ABI_VERSION
    assert lib.tgcn_last_error() is not None


def test_library_is_gfx950_code_object(built_lib):
    blob = open(built_lib, 'rb').read()
    assert b'gfx950' in blob
    assert b'gfx90a' not in blob and b'gfx942' not in blob and b'sm_' not in blob


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: calling a kernel wrapper with CPU tensors raises."""
    import torch
    from textgcn_amd import scoring
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError):
        scoring.score_dense(torch.zeros(2, 4), torch.zeros(3, 4))
