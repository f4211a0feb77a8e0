"""The hot path's input format: the symmetric-normalised user-item Laplacian, as CSR.

Restates what TextGCN/dataset.py:122-157 (_precalculate_normalization + _convert_sp_mat_to_sp_tensor)
produces -- a coalesced COO matrix sorted by (row, col) with fp32 values -- without the reference's DOK /
dgl detour (infeasible at nnz = 100M).  Index work is integer and bit-exact; values follow
the reference's float64 expression (d_r * a_rc) * d_c rounded once to fp32.

Host side is numpy (this is one-off graph construction, not the timed path); device side is three int32 /
fp32 torch tensors handed to the C ABI by pointer.
"""
from dataclasses import dataclass

import numpy as np

INT32_MAX = np.iinfo(np.int32).max


def _as_index(a, name):
    a = np.asarray(a)
    if a.dtype.kind not in 'iu':
        raise TypeError(f'{name} must be an integer array, got {a.dtype}')
    return a.astype(np.int64, copy=False).ravel()


@dataclass
class NormGraph:
    """Normalised Laplacian A (N x N, N = n_users + n_items), rows 0..U-1 users, U..N-1 items."""
    n_users: int
    n_items: int
    rowptr: np.ndarray   # int64 [N+1]
    colidx: np.ndarray   # int32 [nnz]   ascending inside a row
    vals: np.ndarray     # float32 [nnz]

    @property
    def n(self):
        return self.n_users + self.n_items

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_pairs(cls, train_u, train_i, n_users, n_items):
        """Build from the train interactions (internal ids), as dataset.py:122-138 does.

        A = R^ + R^T (duplicate train rows add up, dataset.py:132), deg = row sums (:133),
        d = deg^-0.5 in float64 with inf -> 0 (:134-135), value = (d_r * a_rc) * d_c (:136-137) -> fp32 (:156).
        """
        u = _as_index(train_u, 'train_u')
        i = _as_index(train_i, 'train_i')
        if u.shape != i.shape:
            raise ValueError('train_u and train_i differ in length')
        n_users, n_items = int(n_users), int(n_items)
        n = n_users + n_items
        if n >= INT32_MAX or 2 * len(u) >= INT32_MAX:
            raise ValueError('graph too large for int32 indices')
        if len(u) and (u.min() < 0 or u.max() >= n_users or i.min() < 0 or i.max() >= n_items):
            raise ValueError('interaction id out of range')
        # user rows: key = u * n_items + i ; item rows: key = i * n_users + u  (two half-size sorts)
        ku, mu = np.unique(u * np.int64(max(n_items, 1)) + i, return_counts=True)
        ki, mi = np.unique(i * np.int64(max(n_users, 1)) + u, return_counts=True)
        ru, cu = np.divmod(ku, max(n_items, 1))
        ri, ci = np.divmod(ki, max(n_users, 1))
        rows = np.concatenate([ru, ri + n_users])
        cols = np.concatenate([cu + n_users, ci])
        mult = np.concatenate([mu, mi]).astype(np.float64)
        deg = np.bincount(rows, weights=mult, minlength=n)
        with np.errstate(divide='ignore'):
            d_inv = np.power(deg, -0.5)
        d_inv[np.isinf(d_inv)] = 0.0
        vals = ((d_inv[rows] * mult) * d_inv[cols]).astype(np.float32)
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
        return cls(n_users, n_items, rowptr, cols.astype(np.int32), vals)

    @classmethod
    def from_coo(cls, idx, val, n_users, n_items):
        """Adopt an existing coalesced COO (e.g. a reference dataset's ``norm_matrix``): idx [2, nnz]
        sorted by (row, col), val fp32.  Accepts numpy arrays or a torch sparse COO tensor in `idx`."""
        if hasattr(idx, 'is_sparse') and idx.is_sparse:
            t = idx.coalesce()
            val = t.values().detach().cpu().numpy()
            idx = t.indices().detach().cpu().numpy()
        idx = np.asarray(idx)
        val = np.asarray(val, dtype=np.float32)
        n = int(n_users) + int(n_items)
        rows, cols = _as_index(idx[0], 'rows'), _as_index(idx[1], 'cols')
        if len(rows) >= INT32_MAX or n >= INT32_MAX:
            raise ValueError('graph too large for int32 indices')
        if len(rows):
            if rows.min() < 0 or rows.max() >= n or cols.min() < 0 or cols.max() >= n:
                raise ValueError('COO index out of range')
            key = rows * np.int64(n) + cols
            if np.any(key[1:] <= key[:-1]):
                raise ValueError('COO must be coalesced: sorted by (row, col) without duplicates')
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=rowptr[1:])
        return cls(int(n_users), int(n_items), rowptr, cols.astype(np.int32), val.copy())

    # ------------------------------------------------------------------ views
    def to_coo(self):
        """(idx int64 [2, nnz], val fp32) in the reference's coalesced order."""
        rows = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.rowptr))
        return np.stack([rows, self.colidx.astype(np.int64)]), self.vals

    def degrees(self):
        return np.diff(self.rowptr)

    def transpose_perm(self):
        """perm with (A^T).vals == vals[perm] on the same (symmetric) structure: entry e = (r, c) of the
        transpose takes the value stored at (c, r).  Used for dropped (non-symmetric) matrices."""
        idx, _ = self.to_coo()
        key_t = idx[1] * np.int64(self.n) + idx[0]
        perm = np.argsort(key_t, kind='stable')
        # structure is symmetric: sorted transposed keys == original keys
        return perm.astype(np.int64)

    # ------------------------------------------------------------------ 1-D row partition
    def partition(self, world):
        """nnz-balanced contiguous row blocks, separately for users and items.
        Returns (user_bounds [world+1], item_bounds [world+1]) in global row ids (items offset by U)."""
        def cut(lo, hi):
            if hi == lo:
                return np.full(world + 1, lo, dtype=np.int64)
            # weight = entries + 1 per row so that empty rows are spread too
            w = (self.rowptr[lo + 1:hi + 1] - self.rowptr[lo]) + np.arange(1, hi - lo + 1)
            targets = w[-1] * np.arange(1, world) / world
            inner = lo + 1 + np.searchsorted(w, targets, side='left')
            b = np.concatenate([[lo], np.minimum(inner, hi), [hi]]).astype(np.int64)
            return np.maximum.accumulate(b)
        return cut(0, self.n_users), cut(self.n_users, self.n)

    def row_block(self, r0, r1):
        """CSR of rows [r0, r1): (rowptr int64 rebased to 0, colidx, vals) -- views, no copy of the big arrays."""
        a, b = int(self.rowptr[r0]), int(self.rowptr[r1])
        return self.rowptr[r0:r1 + 1] - a, self.colidx[a:b], self.vals[a:b]


def split_plan_arrays(rowptr, threshold):
    """Host arrays of a tgcn_split_plan_t for a (local) rowptr: rows with more than `threshold` entries are
    cut into chunks of at most `threshold` entries.  Returns None when no row is long."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    lens = np.diff(rowptr)
    long_rows = np.nonzero(lens > threshold)[0]
    if len(long_rows) == 0:
        return None
    counts = -(-lens[long_rows] // threshold)
    long_chunk_ptr = np.zeros(len(long_rows) + 1, dtype=np.int64)
    np.cumsum(counts, out=long_chunk_ptr[1:])
    owner = np.repeat(np.arange(len(long_rows)), counts)
    within = np.arange(long_chunk_ptr[-1]) - long_chunk_ptr[owner]
    beg = rowptr[long_rows][owner] + within * threshold
    end = np.minimum(beg + threshold, rowptr[long_rows + 1][owner])
    return {
        'threshold': int(threshold),
        'chunk_beg': beg.astype(np.int32), 'chunk_end': end.astype(np.int32),
        'long_rows': long_rows.astype(np.int32), 'long_chunk_ptr': long_chunk_ptr.astype(np.int32),
    }


def block_plan_arrays(rowptr, colidx, row_begin, row_end, col_lo, col_hi, block_width, long_threshold=None, min_segment=4):
    """Host array of a tgcn_block_plan_t: rows [row_begin, row_end) of a (local) CSR whose entries lie in columns
    [col_lo, col_hi), cut into column blocks of `block_width` columns.  Returns (blkptr int32 [(n_blocks+1), n_rows],
    n_blocks).  Rows with more than `long_threshold` entries get empty segments (they go through the split plan).
    The block count is capped so that a row's mean segment keeps at least `min_segment` entries."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n_rows = int(row_end - row_begin)
    a, b = int(rowptr[row_begin]), int(rowptr[row_end])
    cols = np.asarray(colidx[a:b], dtype=np.int64)
    if len(cols) and (cols.min() < col_lo or cols.max() >= col_hi):
        raise ValueError('entries outside the declared column range')
    n_blocks = max(1, -(-(col_hi - col_lo) // int(block_width)))
    mean_deg = (b - a) / max(n_rows, 1)
    n_blocks = int(max(1, min(n_blocks, mean_deg // min_segment if mean_deg >= min_segment else 1)))
    width = -(-(col_hi - col_lo) // n_blocks)
    lens = np.diff(rowptr[row_begin:row_end + 1])
    rows = np.repeat(np.arange(n_rows, dtype=np.int64), lens)
    span = np.int64(col_hi - col_lo + 1)
    key = rows * span + (cols - col_lo)                      # ascending: CSR rows ascending, columns ascending inside
    blkptr = np.empty((n_blocks + 1, n_rows), dtype=np.int64)
    base = np.arange(n_rows, dtype=np.int64) * span
    for blk in range(n_blocks + 1):
        bound = min(blk * width, col_hi - col_lo)
        blkptr[blk] = a + np.searchsorted(key, base + bound, side='left')
    blkptr[n_blocks] = rowptr[row_begin + 1:row_end + 1]
    if long_threshold:
        is_long = lens > long_threshold
        blkptr[:, is_long] = rowptr[row_begin:row_end][is_long]
    return blkptr.astype(np.int32), n_blocks


def segment_plan_arrays(rowptr, colidx, phases, segs_per_wg, max_len=128, n_classes=8):
    """Host arrays of a tgcn_segment_plan_t (XCD-affine column blocking, include/tgcn.h).

    phases: list of (row_begin, row_end, col_lo, col_hi, n_blocks) -- row ranges whose entries fall in one column
    range, cut into n_blocks column blocks (a multiple or a divisor of n_classes).  A row's entries inside one block
    form a segment (pieces of at most max_len entries); every piece gets a workspace slot, a row's slots are
    contiguous and in entry order, so adding them in slot order is deterministic.  Pieces are laid out so that the
    workgroup that owns positions [g*segs_per_wg, (g+1)*segs_per_wg) only meets column blocks of class g % n_classes:
    workgroups are dealt round-robin over the 8 XCDs, so each XCD's L2 sees 1/8 of the gathered table (speed only --
    results never depend on the placement).  Rows outside every phase, and empty rows, are `direct` rows.
    """
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n_rows = len(rowptr) - 1
    direct = np.ones(n_rows, dtype=bool)
    S = int(segs_per_wg)
    pos_beg, pos_end, pos_slot, seg_rows, seg_row_slots = [], [], [], [], []
    slot_base = 0
    for (r0, r1, c0, c1, nb) in phases:
        nb = int(nb)
        if nb % n_classes and n_classes % nb:
            raise ValueError('n_blocks must be a multiple or a divisor of n_classes')
        n_r = int(r1 - r0)
        a, b = int(rowptr[r0]), int(rowptr[r1])
        cols = np.asarray(colidx[a:b], dtype=np.int64)
        if len(cols) and (cols.min() < c0 or cols.max() >= c1):
            raise ValueError('entries outside the declared column range')
        lens = np.diff(rowptr[r0:r1 + 1])
        width = -(-(c1 - c0) // nb)
        span = np.int64(c1 - c0 + 1)
        key = np.repeat(np.arange(n_r, dtype=np.int64), lens) * span + (cols - c0)
        base = np.arange(n_r, dtype=np.int64) * span
        blkptr = np.empty((nb + 1, n_r), dtype=np.int64)
        for blk in range(nb):
            blkptr[blk] = a + np.searchsorted(key, base + min(blk * width, c1 - c0), side='left')
        blkptr[nb] = rowptr[r0 + 1:r1 + 1]
        pieces = -(-(blkptr[1:] - blkptr[:-1]) // max_len)           # [nb, n_r]
        row_slots = pieces.sum(axis=0)
        row_base = slot_base + np.cumsum(row_slots) - row_slots
        blk_off = np.cumsum(pieces, axis=0) - pieces
        has = row_slots > 0
        direct[r0:r1] = ~has
        seg_rows.append(r0 + np.nonzero(has)[0])
        seg_row_slots.append(row_slots[has])
        slot_base += int(row_slots.sum())
        n_sub = max(1, nb // n_classes)
        for sub in range(n_sub):
            classes = [[] for _ in range(n_classes)]
            for x in range(n_classes):
                if nb >= n_classes:
                    blk, part, parts = sub * n_classes + x, 0, 1
                else:
                    blk, part, parts = x % nb, x // nb, n_classes // nb
                cnt = pieces[blk]
                rr = np.repeat(np.arange(n_r, dtype=np.int64), cnt)
                within = np.arange(int(cnt.sum()), dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
                beg = blkptr[blk][rr] + within * max_len
                end = np.minimum(beg + max_len, blkptr[blk + 1][rr])
                slot = row_base[rr] + blk_off[blk][rr] + within
                lo, hi = len(beg) * part // parts, len(beg) * (part + 1) // parts
                classes[x] = (beg[lo:hi], end[lo:hi], slot[lo:hi])
            L = max(len(c[0]) for c in classes)
            L = -(-L // S) * S
            if L == 0:
                continue
            tb = np.zeros((n_classes, L), dtype=np.int64)
            te = np.zeros((n_classes, L), dtype=np.int64)
            ts = np.zeros((n_classes, L), dtype=np.int64)
            for x, (cb, ce, cs) in enumerate(classes):
                tb[x, :len(cb)], te[x, :len(cb)], ts[x, :len(cb)] = cb, ce, cs
            inter = lambda t: t.reshape(n_classes, L // S, S).transpose(1, 0, 2).reshape(-1)   # noqa: E731
            pos_beg.append(inter(tb)), pos_end.append(inter(te)), pos_slot.append(inter(ts))
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64)   # noqa: E731
    slots = cat(seg_row_slots)
    seg_row_ptr = np.zeros(len(slots) + 1, dtype=np.int64)
    np.cumsum(slots, out=seg_row_ptr[1:])
    return {
        'seg_beg': cat(pos_beg).astype(np.int32), 'seg_end': cat(pos_end).astype(np.int32),
        'seg_slot': cat(pos_slot).astype(np.int32), 'seg_rows': cat(seg_rows).astype(np.int32),
        'seg_row_ptr': seg_row_ptr.astype(np.int32), 'direct_rows': np.nonzero(direct)[0].astype(np.int32),
        'n_slots': int(slot_base), 'segs_per_wg': S,
    }


def train_mask_csr(train_u, train_i, n_users):
    """CSR over all users of their train items (ascending): the device form of
    base_model.py:257 `train_user_dict[batch_users].explode()`.  Returns (rowptr int64 [U+1], items int32)."""
    u = _as_index(train_u, 'train_u')
    i = _as_index(train_i, 'train_i')
    order = np.lexsort((i, u))
    rowptr = np.zeros(int(n_users) + 1, dtype=np.int64)
    np.cumsum(np.bincount(u, minlength=int(n_users)), out=rowptr[1:])
    return rowptr, i[order].astype(np.int32)
