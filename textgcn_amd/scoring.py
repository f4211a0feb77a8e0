"""Scoring wrappers over the C ABI: dense user x item scores (fp32 MFMA), train-item mask, top-k, pairwise.

Replaces torch.matmul (TextGCN/base_model.py:179), the pandas explode + -inf scatter (:257-258),
torch.topk + round (:261-263) and torch.sum(u*i, 1) (:171).
"""
import torch

from . import _capi


def _dev(t):
    if t.device.type != 'cuda':
        raise RuntimeError('textgcn_amd kernels run on a ROCm GPU only (tensor is on %s)' % t.device)
    return t.device


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise TypeError(f'{name} must be a contiguous float32 tensor')
    return t


def score_dense(users_emb, items_emb, user_ids=None, out=None):
    """S[b, i] = <users_emb[user_ids[b]], items_emb[i]>  (user_ids None: rows of users_emb in order)."""
    dev = _dev(users_emb)
    _f32c(users_emb, 'users_emb'), _f32c(items_emb, 'items_emb')
    if users_emb.dim() != 2 or items_emb.dim() != 2 or users_emb.shape[1] != items_emb.shape[1]:
        raise ValueError('users_emb / items_emb must be [*, d] with equal d')
    if user_ids is not None:
        if user_ids.dtype != torch.int64 or not user_ids.is_contiguous() or user_ids.device != dev:
            raise TypeError('user_ids must be a contiguous int64 tensor on the same device')
        b = user_ids.numel()
    else:
        b = users_emb.shape[0]
    n_items, d = items_emb.shape
    if out is None:
        out = torch.empty((b, n_items), dtype=torch.float32, device=dev)
    elif out.shape != (b, n_items) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError('out must be float32 [B, I] with unit inner stride')
    rc = _capi.lib().tgcn_score_dense_f32(_capi.ptr(users_emb), _capi.ptr(user_ids), b, _capi.ptr(items_emb), n_items, d,
                                          _capi.ptr(out), out.stride(0), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_score_dense_f32')
    return out


def mask_train(scores, mask_rowptr, mask_items):
    """scores[b, mask_items[e]] = -inf for e in [mask_rowptr[b], mask_rowptr[b+1]) -- in place."""
    dev = _dev(scores)
    if mask_rowptr.dtype != torch.int32 or mask_items.dtype != torch.int32:
        raise TypeError('mask arrays must be int32')
    if mask_rowptr.numel() != scores.shape[0] + 1:
        raise ValueError('mask_rowptr must have B+1 entries')
    rc = _capi.lib().tgcn_mask_f32(_capi.ptr(scores), scores.stride(0), scores.shape[0], scores.shape[1],
                                   _capi.ptr(mask_rowptr), _capi.ptr(mask_items), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_mask_f32')
    return scores


MAX_K_PER_PASS = 64   # the kernels keep a user's list in one register per lane (include/tgcn.h)


def topk(scores, k, round4=False):
    """(values [B,k] fp32, indices [B,k] int64), ordered by (value desc, index asc).  k > 64 (the kernels' list
    length) takes ceil(k/64) passes over a copy in which the items already taken are set to -inf; on a row with
    fewer than k finite scores the positions past the last finite one hold -inf with unspecified item ids, as with
    torch.topk (base_model.py:261)."""
    dev = _dev(scores)
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise TypeError('scores must be float32 [B, I] with unit inner stride')
    if k > MAX_K_PER_PASS:
        if k > scores.shape[1]:
            raise ValueError('k exceeds the number of items')
        work, vals, idxs = scores.clone(), [], []
        for k0 in range(0, k, MAX_K_PER_PASS):
            v, i = topk(work, min(MAX_K_PER_PASS, k - k0), round4)
            vals.append(v), idxs.append(i)
            work.scatter_(1, retired_positions(i), float('-inf'))
        return torch.cat(vals, dim=1), torch.cat(idxs, dim=1)
    b = scores.shape[0]
    val = torch.empty((b, k), dtype=torch.float32, device=dev)
    idx = torch.empty((b, k), dtype=torch.int64, device=dev)
    rc = _capi.lib().tgcn_topk_f32(_capi.ptr(scores), scores.stride(0), b, scores.shape[1], int(k), 1 if round4 else 0,
                                   _capi.ptr(val), _capi.ptr(idx), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_topk_f32')
    return val, idx


NO_ITEM = 2 ** 31 - 1      # the id the kernels leave in a list position no score could fill (a row with fewer than k non-NaN scores)


def retired_positions(idx):
    """The ids of one pass as positions to retire before the next pass: NO_ITEM entries (nothing was taken there) are replaced by
    the row's first id -- already retired, so the duplicate changes nothing -- or by 0 on a row that took nothing at all.  (As
    raw positions they are out of range: a scatter with them faults the GPU.)"""
    first = torch.where(idx[:, :1] == NO_ITEM, torch.zeros_like(idx[:, :1]), idx[:, :1])
    return torch.where(idx == NO_ITEM, first.expand_as(idx), idx)


_WORKSPACE = {}
_PLAN_BYTES = {}    # (B, I, d, k) -> tgcn_score_topk_workspace_bytes
_PACK_BYTES = {}    # (I, d) -> tgcn_item_pack_bytes
_LAST_CALL = {}     # (device, slot) -> (B, I, d, k, prefilter) of the last score_topk call that used the slot's workspace


def last_call(dev, slot=0):
    """(B, I, d, k, prefilter) of the last score_topk call on (dev, slot), or None: what call_stats / fallback_count can be asked about"""
    return _LAST_CALL.get((_capi.resolve_device(dev), slot))


def _check_last_call(dev, slot, b, n_items, d, k, prefilter=None):
    got = _LAST_CALL.get((dev, slot))
    if got is None:
        raise RuntimeError('no score_topk call has used this slot')
    want = (int(b), int(n_items), int(d), int(min(k, MAX_K_PER_PASS)))
    if got[:4] != want or (prefilter is not None and bool(prefilter) != got[4]):
        raise ValueError(f'the last score_topk call on slot {slot} was (B, I, d, k, prefilter) = {got}, not {want + (prefilter,)}: its '
                         'workspace offsets belong to that shape (a partial tail chunk?)')


def _workspace(dev, nbytes, slot=0):
    """one growing scratch buffer per (device, slot) -- calls that run concurrently on different streams must use
    different slots (torch's caching allocator returns 256-byte aligned blocks)"""
    key = (dev, slot)
    buf = _WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _WORKSPACE[key] = buf
    return buf


def fallback_count(dev, b, n_items, d, k, slot=0):
    """Users of the last score_topk call on (dev, slot) that took the exact fallback (diagnostic; synchronises)."""
    import ctypes
    dev = _capi.resolve_device(dev)
    _check_last_call(dev, slot, b, n_items, d, k)
    ws = _WORKSPACE[(dev, slot)]
    out = ctypes.c_int32(0)
    rc = _capi.lib().tgcn_score_topk_fallback_count(_capi.ptr(ws), b, n_items, d, int(min(k, MAX_K_PER_PASS)), ctypes.byref(out),
                                                    _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_score_topk_fallback_count')
    return out.value


def call_stats(dev, b, n_items, d, k, prefilter, slot=0):
    """What the last score_topk call on (dev, slot) did, summed over its users (tgcn_score_topk_stats; diagnostic, synchronises):
    dict(fallback_users, kept_pairs, rescored_pairs, logged_pairs)."""
    import ctypes
    dev = _capi.resolve_device(dev)
    _check_last_call(dev, slot, b, n_items, d, k, prefilter)
    ws = _WORKSPACE[(dev, slot)]
    out = (ctypes.c_int64 * 4)()
    rc = _capi.lib().tgcn_score_topk_stats(_capi.ptr(ws), b, n_items, d, int(min(k, MAX_K_PER_PASS)), 1 if prefilter else 0, out,
                                           _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_score_topk_stats')
    return dict(zip(('fallback_users', 'kept_pairs', 'rescored_pairs', 'logged_pairs'), [int(x) for x in out]))


LONG_CALL_PAIRS = 1 << 33     # users x items of one call from which more than two calls in flight only queue


def calls_in_flight(users_per_call, n_items, streams):
    """How many consecutive score_topk calls a caller should keep in flight (on `streams` HIP streams of its own).  A short call
    leaves the chip idle between its small launches, and a neighbour's kernels fill the gaps: four in flight measured best on
    50 000 - 60 000 items.  A call whose filter alone runs for milliseconds (16 384 users x 2 M items: 6 ms) fills the chip by itself;
    further calls only queue behind it and their small launches delay its own: 5.31 / 5.20 / 4.49 T pairs/s with 1 / 2 / 4 calls in
    flight (tools/score_mode_bench.py --shapes c4 --users 16384) -- two, so that the host's work for the next call still overlaps."""
    return min(int(streams), 2) if int(users_per_call) * int(n_items) >= LONG_CALL_PAIRS else int(streams)


def item_pack(items_emb):
    """The item operand of the prefilter's bf16 pass (tgcn_item_pack_bf16): per row its bf16 image and the row's factors of the
    error bound, as a uint8 tensor.  Pack once per item table and hand it to score_topk(prefilter=True, item_pack=...).
    None for widths the bf16 pass does not take (score_topk then runs the fp32 path)."""
    dev = _dev(items_emb)
    _f32c(items_emb, 'items_emb')
    n_items, d = items_emb.shape
    need = _capi.lib().tgcn_item_pack_bytes(n_items, d)
    if need <= 0:
        return None
    out = torch.empty(need, dtype=torch.uint8, device=dev)
    rc = _capi.lib().tgcn_item_pack_bf16(_capi.ptr(items_emb), n_items, d, _capi.ptr(out), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_item_pack_bf16')
    return out


def item_norms(items_emb):
    """[I, 2] floats: the item factors (row norm, norm of the row's bf16 rounding residual) of the prefilter's error bound
    (tgcn_item_norms_f32; diagnostic -- the pack carries them, rounded up to bf16)."""
    dev = _dev(items_emb)
    _f32c(items_emb, 'items_emb')
    out = torch.empty((items_emb.shape[0], 2), dtype=torch.float32, device=dev)
    rc = _capi.lib().tgcn_item_norms_f32(_capi.ptr(items_emb), items_emb.shape[0], items_emb.shape[1], _capi.ptr(out),
                                         _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_item_norms_f32')
    return out


def score_topk(users_emb, items_emb, k, user_ids=None, mask_rowptr=None, mask_items=None, round4=False, slot=0, prefilter=False,
               item_pack=None):
    """Fused predict step: top-k over all items of the masked scores (tgcn_score_topk_f32).  Same result as
    score_dense -> mask_train -> topk, without the [B, I] matrix.  mask_* is a CSR over the batch rows.
    `slot` selects the scratch buffer: use distinct slots for calls issued on different streams.
    `prefilter`: find the candidates with the bf16 pass and rescore them in fp32 (tgcn_score_topk_prefilter_f32) -- the same
    result bit for bit; `item_pack` = item_pack(items_emb) when the table is shared by many calls."""
    dev = _dev(users_emb)
    k = int(k)
    _f32c(users_emb, 'users_emb'), _f32c(items_emb, 'items_emb')
    if users_emb.shape[1] != items_emb.shape[1]:
        raise ValueError('users_emb / items_emb differ in width')
    if user_ids is not None:
        if user_ids.dtype != torch.int64 or not user_ids.is_contiguous() or user_ids.device != dev:
            raise TypeError('user_ids must be a contiguous int64 tensor on the same device')
        b = user_ids.numel()
    else:
        b = users_emb.shape[0]
    if mask_rowptr is not None:
        if mask_rowptr.dtype != torch.int32 or mask_items.dtype != torch.int32 or mask_rowptr.numel() != b + 1:
            raise TypeError('mask arrays must be int32 with B+1 row pointers')
    n_items, d = items_emb.shape
    if k > MAX_K_PER_PASS:
        # ceil(k/64) passes; the items taken so far join the user's mask list for the next pass
        if k > n_items:
            raise ValueError('k exceeds the number of items')
        rows = torch.arange(b, device=dev, dtype=torch.int64)
        if mask_rowptr is None:
            rowptr, keys = torch.zeros(b + 1, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)
        else:
            rowptr = mask_rowptr.to(torch.int64)
            keys = torch.repeat_interleave(rows, rowptr[1:] - rowptr[:-1]) * n_items + mask_items.to(torch.int64)
        vals, idxs = [], []
        for k0 in range(0, k, MAX_K_PER_PASS):
            kk = min(MAX_K_PER_PASS, k - k0)
            v, i = score_topk(users_emb, items_emb, kk, user_ids, rowptr.to(torch.int32),
                              (keys % n_items).to(torch.int32) if keys.numel() else torch.zeros(1, dtype=torch.int32, device=dev),
                              round4, slot, prefilter, item_pack)
            vals.append(v), idxs.append(i)
            keys = torch.sort(torch.cat([keys, (rows[:, None] * n_items + retired_positions(i)).reshape(-1)]))[0]
            rowptr = rowptr + kk * torch.arange(b + 1, device=dev, dtype=torch.int64)
        return torch.cat(vals, dim=1), torch.cat(idxs, dim=1)
    val = torch.empty((b, k), dtype=torch.float32, device=dev)
    idx = torch.empty((b, k), dtype=torch.int64, device=dev)
    lib = _capi.lib()
    # the call's plan depends on its shape alone: asked of the library once per (B, I, d, k), as is the pack's size per (I, d)
    shape = (b, n_items, d, k)
    need = _PLAN_BYTES.get(shape)
    if need is None:
        need = _PLAN_BYTES[shape] = max(int(lib.tgcn_score_topk_workspace_bytes(b, n_items, d, int(k))), 256)
    ws = _WORKSPACE.get((dev, slot))
    if ws is None or ws.numel() < need:
        ws = _workspace(dev, need, slot)
    _LAST_CALL[(dev, slot)] = (b, n_items, d, k, bool(prefilter))
    # raw integers for the pointer arguments (argtypes are bound once, at load): no ctypes object per argument
    mrp = 0 if mask_rowptr is None else mask_rowptr.data_ptr()
    mit = 0 if mask_items is None else mask_items.data_ptr()
    uid = 0 if user_ids is None else user_ids.data_ptr()
    stream = _capi.raw_stream(dev)
    if prefilter:
        pk = 0
        if item_pack is not None:
            pb = _PACK_BYTES.get((n_items, d))
            if pb is None:
                pb = _PACK_BYTES[(n_items, d)] = int(lib.tgcn_item_pack_bytes(n_items, d))
            if item_pack.dtype != torch.uint8 or item_pack.numel() != pb or item_pack.device != dev or not item_pack.is_contiguous():
                raise TypeError('item_pack must be the uint8 tensor of item_pack(items_emb) on the same device')
            pk = item_pack.data_ptr()
        rc = lib.tgcn_score_topk_prefilter_f32(users_emb.data_ptr(), uid, b, items_emb.data_ptr(), n_items, d, mrp, mit, k,
                                               1 if round4 else 0, pk, val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        if rc:
            _capi.check(rc, 'tgcn_score_topk_prefilter_f32')
        return val, idx
    rc = lib.tgcn_score_topk_f32(users_emb.data_ptr(), uid, b, items_emb.data_ptr(), n_items, d, mrp, mit, k, 1 if round4 else 0,
                                 val.data_ptr(), idx.data_ptr(), ws.data_ptr(), ws.numel(), stream)
    if rc:
        _capi.check(rc, 'tgcn_score_topk_f32')
    return val, idx


def score_pairwise(u_table, v_table, users=None, items=None):
    """out[r] = <u_table[users[r]], v_table[items[r]]> (None: row r)."""
    dev = _dev(u_table)
    _f32c(u_table, 'u_table'), _f32c(v_table, 'v_table')
    n = users.numel() if users is not None else u_table.shape[0]
    if (items.numel() if items is not None else v_table.shape[0]) != n:
        raise ValueError('users / items differ in length')
    for t in (users, items):
        if t is not None and (t.dtype != torch.int64 or not t.is_contiguous()):
            raise TypeError('index tensors must be contiguous int64')
    out = torch.empty((n,), dtype=torch.float32, device=dev)
    rc = _capi.lib().tgcn_score_pairwise_f32(_capi.ptr(u_table), _capi.ptr(users), _capi.ptr(v_table), _capi.ptr(items), n,
                                             u_table.shape[1], _capi.ptr(out), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_score_pairwise_f32')
    return out


def score_candidates(u_table, users, items_table, cand, mask_rowptr=None, mask_items=None):
    """out[b, j] = <u_table[users[b]], items_table[cand[b, j]]>; train items of the user (mask CSR over ALL users)
    score -inf.  cand: int64 [B, m]."""
    dev = _dev(u_table)
    _f32c(u_table, 'u_table'), _f32c(items_table, 'items_table')
    if users.dtype != torch.int64 or cand.dtype != torch.int64 or not cand.is_contiguous() or cand.dim() != 2:
        raise TypeError('users / cand must be int64 (cand contiguous [B, m])')
    b, m = cand.shape
    if users.numel() != b:
        raise ValueError('users and cand differ in length')
    out = torch.empty((b, m), dtype=torch.float32, device=dev)
    rc = _capi.lib().tgcn_score_candidates_f32(_capi.ptr(u_table), _capi.ptr(users.contiguous()), _capi.ptr(items_table),
                                               _capi.ptr(cand), b, m, u_table.shape[1], _capi.ptr(mask_rowptr),
                                               _capi.ptr(mask_items), _capi.ptr(out), _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_score_candidates_f32')
    return out
