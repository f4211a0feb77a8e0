"""ctypes binding of libtgcn.so (include/tgcn.h).  There is NO fallback: if the HIP library is missing
or a call fails, a RuntimeError is raised -- the product path never routes through torch ops or the
CPU oracle instead."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_lib', 'libtgcn.so')

TGCN_ABI_VERSION = 9
TGCN_COMM_ID_BYTES = 128
SPMM_AUTO, SPMM_WAVE_PER_ROW = 0, 1


class SplitPlanStruct(Structure):
    """mirror of tgcn_split_plan_t"""
    _fields_ = [('threshold', c_int32), ('n_chunks', c_int32), ('n_long', c_int32), ('_pad', c_int32),
                ('chunk_beg', c_void_p), ('chunk_end', c_void_p), ('long_rows', c_void_p),
                ('long_chunk_ptr', c_void_p), ('workspace', c_void_p)]


class SegmentPlanStruct(Structure):
    """mirror of tgcn_segment_plan_t"""
    _fields_ = [('n_tiles', c_int32), ('tile_entries', c_int32), ('n_seg_rows', c_int32), ('n_direct_rows', c_int32),
                ('n_slots', c_int32), ('_pad', c_int32), ('tile_meta', c_void_p), ('ent_col', c_void_p),
                ('ent_val', c_void_p), ('ent_flags', c_void_p), ('seg_rows', c_void_p), ('row_slot_ptr', c_void_p), ('row_slots', c_void_p),
                ('direct_rows', c_void_p), ('workspace', c_void_p), ('direct_groups', c_void_p), ('n_direct_groups', c_int32),
                ('_pad2', c_int32)]


_SIGNATURES = {
    'tgcn_abi_version': (ctypes.c_int, []),
    'tgcn_last_error': (c_char_p, []),
    'tgcn_spmm_csr_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p,
                                         c_void_p, c_void_p, c_float, POINTER(SplitPlanStruct), c_void_p, c_uint32, c_void_p]),
    'tgcn_spmm_groups_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p,
                                            c_void_p, c_void_p, c_float, POINTER(SplitPlanStruct), c_void_p, c_int64, c_uint32, c_void_p]),
    'tgcn_spmm_segmented_f32': (ctypes.c_int, [POINTER(SegmentPlanStruct), c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                               c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_uint32, c_void_p]),
    'tgcn_score_dense_f32': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                            c_void_p]),
    'tgcn_mask_f32': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'tgcn_topk_f32': (ctypes.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'tgcn_score_topk_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'tgcn_score_topk_f32': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                           c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'tgcn_score_topk_prefilter_f32': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                                     c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'tgcn_score_topk_fallback_count': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_int32), c_void_p]),
    'tgcn_score_topk_stats': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, POINTER(c_int64), c_void_p]),
    'tgcn_item_pack_bytes': (ctypes.c_int64, [c_int32, c_int32]),
    'tgcn_item_pack_bf16': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'tgcn_item_norms_f32': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    'tgcn_ltr_folded_width': (c_int32, [c_int32, c_int32]),
    'tgcn_ltr_fold_users_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                               POINTER(c_float), c_float, c_void_p, c_void_p]),
    'tgcn_ltr_pack_items_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'tgcn_ltr_pair_features_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                  c_int32, c_int32, c_void_p, c_void_p]),
    'tgcn_score_candidates_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                                 c_void_p, c_void_p, c_void_p]),
    'tgcn_dropout_values_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tgcn_bpr_pairs_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tgcn_reg_rows_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tgcn_comm_unique_id': (ctypes.c_int, [c_void_p]),
    'tgcn_comm_init_rank': (ctypes.c_int, [POINTER(c_void_p), c_int32, c_int32, c_char_p]),
    'tgcn_comm_destroy': (ctypes.c_int, [c_void_p]),
    'tgcn_allgather_rows': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    'tgcn_score_pairwise_f32': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                               c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load libtgcn.so once; loud failure if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -m textgcn_amd.build` (hipcc, gfx950). '
                'textgcn_amd has no CPU / torch fallback for its kernels.')
        import torch  # noqa: F401  -- first: libtgcn.so binds to the HIP runtime bundled with torch (build.py)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.tgcn_abi_version() != TGCN_ABI_VERSION:
            raise RuntimeError(f'libtgcn.so ABI {handle.tgcn_abi_version()} != binding {TGCN_ABI_VERSION}; rebuild')
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().tgcn_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (code {rc}): {msg}')


def resolve_device(device):
    """torch.device with its index filled in: `torch.device('cuda')` (what the reference's parser hands over,
    TextGCN/parser.py:174) compares unequal to the `cuda:0` every tensor reports, so it is resolved once, where a
    device enters the package."""
    import torch
    dev = torch.device(device)
    if dev.type == 'cuda' and dev.index is None and torch.cuda.is_available():
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)"""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


_RAW_STREAM = None


def raw_stream(device):
    """the current HIP stream of `device` as a plain integer (what a c_void_p parameter takes): torch's raw-stream getter where
    this build has it (one C call, no Stream object), else the public way"""
    global _RAW_STREAM
    if _RAW_STREAM is None:
        import torch
        get = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        _RAW_STREAM = get if get is not None else (lambda idx: torch.cuda.current_stream(idx).cuda_stream)
    return _RAW_STREAM(device.index)
