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


def test_c_abi_rejects_bad_arguments_with_message(built_lib):
    """Argument validation happens before any launch, so it can be exercised without a GPU: error code + message."""
    from textgcn_amd import _capi
    lib = _capi.lib()
    rc = lib.tgcn_spmm_csr_f32(None, None, None, 5, None, 5, 64, None, None, None, 1.0, None, None, 0, None)
    assert rc == -1 and b'NULL' in lib.tgcn_last_error()
    rc = lib.tgcn_spmm_csr_f32(None, None, None, 5, None, 5, 0, None, None, None, 1.0, None, None, 0, None)
    assert rc == -1 and b'd out of range' in lib.tgcn_last_error()
    assert lib.tgcn_spmm_csr_f32(None, None, None, 0, None, 0, 64, None, None, None, 1.0, None, None, 0, None) == 0   # empty: no-op
    rc = lib.tgcn_topk_f32(None, 10, 2, 10, 65, 0, None, None, None)
    assert rc == -1 and b'k must be in [1, 64]' in lib.tgcn_last_error()
    rc = lib.tgcn_topk_f32(None, 10, 2, 10, 11, 0, None, None, None)
    assert rc == -1 and b'exceeds the number of items' in lib.tgcn_last_error()
    rc = lib.tgcn_score_dense_f32(None, None, 4, None, 4, 8, None, 2, None)
    assert rc == -1
    assert lib.tgcn_score_topk_workspace_bytes(2048, 50000, 64, 40) > 0
    assert lib.tgcn_score_topk_workspace_bytes(0, 50000, 64, 40) == 0
    rc = lib.tgcn_score_topk_f32(None, None, 4, None, 100, 64, None, None, 10, 0, None, None, None, 0, None)
    assert rc == -1
    rc = lib.tgcn_score_topk_prefilter_f32(None, None, 4, None, 100, 64, None, None, 10, 0, None, None, None, None, 0, None)
    assert rc == -1
    assert lib.tgcn_score_topk_prefilter_f32(None, None, 0, None, 100, 64, None, None, 10, 0, None, None, None, None, 0, None) == 0
    rc = lib.tgcn_score_topk_prefilter_f32(8, None, 4, 8, 100, 64, None, None, 10, 0, 24, 8, 8, None, 0, None)
    assert rc == -1 and b'item_pack' in lib.tgcn_last_error()
    assert lib.tgcn_item_pack_bytes(50000, 64) == 50000 * 144 and lib.tgcn_item_pack_bytes(60000, 128) == 60000 * 272
    assert lib.tgcn_item_pack_bytes(100, 50) == 100 * 144 and lib.tgcn_item_pack_bytes(100, 960) == 100 * (2 * 960 + 16) and lib.tgcn_item_pack_bytes(100, 2048) == 0 and lib.tgcn_item_pack_bytes(-1, 64) < 0
    rc = lib.tgcn_item_pack_bf16(None, 10, 64, None, None)
    assert rc == -1 and b'NULL' in lib.tgcn_last_error()
    rc = lib.tgcn_item_pack_bf16(None, 10, 2048, None, None)
    assert rc == -1 and b'width' in lib.tgcn_last_error()
    rc = lib.tgcn_item_norms_f32(None, 10, 64, None, None)
    assert rc == -1 and b'NULL' in lib.tgcn_last_error()
    rc = lib.tgcn_score_topk_fallback_count(None, 4, 100, 64, 10, None, None)
    assert rc == -1
    assert lib.tgcn_ltr_folded_width(128, 384) == 960 and lib.tgcn_ltr_folded_width(64, 384) == 896
    # training-step entries
    rc = lib.tgcn_dropout_values_f32(None, None, None, 1, 1.5, None, None, None, None, None, 10, 0, None, None, None, None, None)
    assert rc == -1 and b'keep_prob' in lib.tgcn_last_error()
    rc = lib.tgcn_dropout_values_f32(None, None, None, 1, 0.6, None, None, None, None, None, 10, 0, None, None, None, None, None)
    assert rc == -1 and b'NULL' in lib.tgcn_last_error()
    assert lib.tgcn_dropout_values_f32(None, None, None, 1, 0.6, None, None, None, None, None, 0, 0, None, None, None, None, None) == 0
    rc = lib.tgcn_bpr_pairs_f32(None, None, None, None, None, 8, 1, 1024, 1.0, None, None, None, None, None)
    assert rc == -1 and b'd out of range' in lib.tgcn_last_error()
    rc = lib.tgcn_bpr_pairs_f32(None, None, None, None, None, 8, 0, 64, 1.0, None, None, None, None, None)
    assert rc == -1
    assert lib.tgcn_bpr_pairs_f32(None, None, None, None, None, 0, 1, 64, 1.0, None, None, None, None, None) == 0
    rc = lib.tgcn_reg_rows_f32(None, None, None, None, None, 8, 1, 64, 0.1, None, None, None, None, None)
    assert rc == -1
    # collective entries: argument checks come before RCCL is touched
    rc = lib.tgcn_allgather_rows(None, None, None, 4, 64, None)
    assert rc == -1 and b'comm is NULL' in lib.tgcn_last_error()
    rc = lib.tgcn_comm_init_rank(None, 2, 0, None)
    assert rc == -1
    assert lib.tgcn_comm_destroy(None) == 0


def test_workspace_plan_over_a_grid_of_shapes(built_lib):
    """tgcn_score_topk_workspace_bytes is host arithmetic (the plan of the fused top-k: splits, the threshold sample of either entry
    point, log / list / bitmap regions): positive, finite, monotone in the number of users, and the same on every call, over
    shapes on both sides of every regime switch (small catalogues, the in-kernel sample's lower and upper catalogue bounds, wide rows)."""
    from textgcn_amd import _capi
    lib = _capi.lib()
    for d in (1, 6, 64, 100, 128, 136, 256, 960, 1024):
        for n_items in (1, 63, 8192, 8193, 20_000, 35_000, 50_000, 131_072, 131_073, 800_000, 1_500_000, 1_600_000, 2_000_000):
            for k in (1, 10, 40, 64):
                if k > n_items:
                    continue
                prev = 0
                for b in (1, 255, 2048, 4097, 16384, 65536):
                    need = lib.tgcn_score_topk_workspace_bytes(b, n_items, d, k)
                    assert 0 < need < (1 << 42), (b, n_items, d, k, need)
                    assert need == lib.tgcn_score_topk_workspace_bytes(b, n_items, d, k)
                    assert need >= prev, (b, n_items, d, k)
                    prev = need
    assert lib.tgcn_item_pack_bytes(50_000, 64) == 50_000 * 144 and lib.tgcn_item_pack_bytes(60_000, 960) == 60_000 * (2 * 960 + 16)
