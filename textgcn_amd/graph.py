"""The hot path's input format: the symmetric-normalised user-item Laplacian, as CSR.

Restates what TextGCN/dataset.py:122-157 (_precalculate_normalization + _convert_sp_mat_to_sp_tensor)
produces -- a coalesced COO matrix sorted by (row, col) with fp32 values -- without the reference's DOK /
dgl detour (infeasible at nnz = 100M).  Index work is integer and bit-exact; values follow
the reference's float64 expression (d_r * a_rc) * d_c rounded once to fp32.

Host side is numpy (this is one-off graph construction, not the timed path: config 4's 100 M pairs take ~6 s on the GPU box's
host cores -- the same steps through torch on the GPU took 75 s, profiles/r04_experiments.md section 10); device side is three
int32 / fp32 torch tensors handed to the C ABI by pointer.
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

    def train_mask(self):
        """(rowptr int64 [U+1], items int32): every user's train items, ascending and distinct -- the user rows of A with the
        column offset taken off.  Equal to train_mask_csr(train_u, train_i, U) up to repeated pairs (a mask is idempotent)."""
        e = int(self.rowptr[self.n_users])
        return np.asarray(self.rowptr[:self.n_users + 1], dtype=np.int64), (np.asarray(self.colidx[:e]) - np.int32(self.n_users)).astype(np.int32)

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


def row_groups(rowptr, rows=None, threshold=None, max_rows=8, target_entries=64, longest_first=False, single_len=32, phases=None):
    """Groups of CONSECUTIVE rows for tgcn_spmm_groups_f32 (include/tgcn.h): int32 [n_groups, 2] = (first row, rows).

    rows: the ascending row ids to cover (None: every row); rows with more than `threshold` entries are left out (the split
    plan's chunk waves own them).  A group is a run of consecutive covered rows whose entries START inside one window of
    `target_entries` consecutive stored entries, at most `max_rows` of them -- so a group holds about target_entries entries
    (a row that starts in the window brings all of its entries along), one wave's worth of a few gather batches.  A row
    of `single_len` entries or more is a group of its own (the kernel walks it as one wave per row: its round trips are
    already amortised, config 2's 50-entry rows lose 5 % in shared groups).
    longest_first: groups sorted by entry count, descending (the launch's tail is its last long group).
    phases: optional row boundaries [r_0 < r_1 < ...]: no group crosses one, and with longest_first the groups are sorted INSIDE
    each range [r_j, r_j+1), the ranges handed out last to first -- row blocks that share a gather table (item rows, then user
    rows) stay together, so the waves in flight gather from ONE table."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = len(rowptr) - 1
    lens = np.diff(rowptr)
    if rows is None:
        cover = np.ones(n, dtype=bool)
    else:
        cover = np.zeros(n, dtype=bool)
        cover[np.asarray(rows, dtype=np.int64)] = True
    if threshold is not None:
        cover &= lens <= threshold
    idx = np.nonzero(cover)[0]
    if len(idx) == 0:
        return np.zeros((0, 2), dtype=np.int32)
    win = rowptr[idx] // int(target_entries)
    new = np.ones(len(idx), dtype=bool)
    new[1:] = (np.diff(idx) != 1) | (np.diff(win) != 0)       # a gap in the covered rows, or the next window
    ph = None
    if phases is not None and len(phases):
        ph = np.searchsorted(np.asarray(phases, dtype=np.int64), idx, side='right')
        new[1:] |= np.diff(ph) != 0
    run = np.cumsum(new) - 1
    pos = np.arange(len(idx)) - np.nonzero(new)[0][run]        # position inside the run
    new |= (pos % int(max_rows)) == 0
    if single_len:
        single = lens[idx] >= int(single_len)
        new |= single
        new[1:] |= single[:-1]
    first = idx[new]
    cnt = np.diff(np.append(np.nonzero(new)[0], len(idx)))
    groups = np.stack([first, cnt], axis=1)
    if longest_first:
        ent = rowptr[first + cnt] - rowptr[first]
        if ph is None:
            groups = groups[np.argsort(-ent, kind='stable')]
        else:
            groups = groups[np.lexsort((-ent, -ph[new]))]       # last phase first, longest first inside a phase
    return np.ascontiguousarray(groups.astype(np.int32))


def segment_plan_arrays(rowptr, colidx, vals, phases, tile_entries=256, n_classes=8, min_row_len=0):
    """Host arrays of a tgcn_segment_plan_t (XCD-affine column blocking, include/tgcn.h).

    phases: list of (row_begin, row_end, col_lo, col_hi, n_blocks[, n_classes[, min_row_len]]) -- row ranges whose entries fall in one
    column range (the optional seventh member overrides `min_row_len` for the phase), cut into n_blocks column blocks (a multiple of the phase's n_classes: 8 = one class per XCD, 4 = two XCDs share a
    class -- for a table of which a QUARTER fits an L2: half the pieces per row).  The entries of a phase are copied into
    n_classes streams: stream x holds the entries of blocks x, x + n_classes, ... ordered by (block, row, column).
    A row's run inside one block is a segment; streams are cut into tiles of `tile_entries` entries (one wavefront
    each) and a segment crossing a cut becomes two pieces.  The last entry of every piece has its bit set in
    `ent_flags` (one 64-bit word per 64 entries); pieces are numbered in stream order (workspace slots).  Tiles are laid out so that a workgroup (4
    consecutive tiles) stays inside one stream and workgroups g, g + n_classes, ... share a stream: workgroups are
    dealt round-robin over the 8 XCDs, so each XCD's L2 serves 1/n_classes of the gathered table (speed only --
    results never depend on the placement).  A row's value is the sum of its pieces in column order (`row_slots`).
    Rows outside every phase, rows with fewer than `min_row_len` entries and rows without entries are `direct` rows.
    """
    rowptr = np.asarray(rowptr, dtype=np.int64)
    colidx = np.asarray(colidx)
    vals = np.asarray(vals, dtype=np.float32)
    n_rows = len(rowptr) - 1
    T = int(tile_entries)
    if T <= 0 or T % 64:
        raise ValueError('tile_entries must be a positive multiple of 64')
    direct = np.ones(n_rows, dtype=bool)
    tiles_col, tiles_val, tiles_meta, tiles_src = [], [], [], []      # per phase, already in launch order
    flag_orig, flag_slot = [], []                      # piece ends: original entry offset, slot id
    slot_base = 0
    default_classes = n_classes
    for ph in phases:
        r0, r1, c0, c1, nb = ph[:5]
        n_classes = int(ph[5]) if len(ph) > 5 else default_classes
        if n_classes not in (1, 2, 4, 8):
            raise ValueError('n_classes must divide 8 (workgroups are dealt round-robin over the 8 XCDs)')
        nb = int(nb)
        if nb <= 0 or nb % n_classes:
            raise ValueError('n_blocks must be a positive multiple of n_classes')
        n_r = int(r1 - r0)
        a, b = int(rowptr[r0]), int(rowptr[r1])
        if b == a:
            continue
        cols = np.asarray(colidx[a:b], dtype=np.int64)
        if cols.min() < c0 or cols.max() >= c1:
            raise ValueError('entries outside the declared column range')
        lens = np.diff(rowptr[r0:r1 + 1])
        rows = np.repeat(np.arange(n_r, dtype=np.int64), lens)
        ent_off = np.arange(a, b, dtype=np.int64)
        mrl = int(ph[6]) if len(ph) > 6 else min_row_len
        if mrl > 0:               # short rows leave too few entries per block: they stay direct
            keep = (lens >= mrl)[rows]
            cols, rows, ent_off = cols[keep], rows[keep], ent_off[keep]
            if len(cols) == 0:
                continue
        width = -(-(c1 - c0) // nb)
        blk = (cols - c0) // width
        cls, sub = blk % n_classes, blk // n_classes
        n_sub = nb // n_classes
        order = np.argsort((cls * n_sub + sub) * n_r + rows, kind='stable')      # (class, block, row, column)
        o_cls, o_key = cls[order], (blk * n_r + rows)[order]
        cls_len = np.bincount(o_cls, minlength=n_classes)
        cls_start = np.cumsum(cls_len) - cls_len
        pos_in_cls = np.arange(len(order), dtype=np.int64) - cls_start[o_cls]
        last = np.ones(len(order), dtype=bool)
        last[:-1] = (o_key[1:] != o_key[:-1]) | (o_cls[1:] != o_cls[:-1])        # segment ends
        last |= (pos_in_cls % T) == T - 1                                         # tile cuts
        tiles_per_cls = -(-cls_len // T)
        n_t = int(-(-tiles_per_cls.max() // 4) * 4)                               # tiles per class, padded to workgroups
        while (n_t // 4 * n_classes) % 8:                                         # a phase is a whole number of 8-workgroup rounds, so
            n_t += 4                                                              # that the next phase's classes start on XCD class 0
        tile_in_cls = pos_in_cls // T
        # launch order: workgroup w = (tile_in_cls // 4) * n_classes + class, tile = w * 4 + tile_in_cls % 4
        tile_id = ((tile_in_cls // 4) * n_classes + o_cls) * 4 + tile_in_cls % 4
        dst = tile_id * T + pos_in_cls % T
        ent_col = np.zeros(n_t * n_classes * T, dtype=np.int64)
        ent_val = np.zeros(n_t * n_classes * T, dtype=np.float32)
        ent_col[dst] = cols[order] | (last.astype(np.int64) << 31)
        ent_val[dst] = vals[ent_off[order]]
        ent_src = np.zeros(n_t * n_classes * T, dtype=np.int64)
        ent_src[dst] = ent_off[order]
        slot = slot_base + np.cumsum(last) - 1                                    # slot of the piece an entry ends
        meta = np.zeros((n_t * n_classes, 2), dtype=np.int64)
        meta[:, 1] = np.bincount(tile_id, minlength=n_t * n_classes)
        first = np.ones(len(order), dtype=bool)
        first[1:] = tile_id[1:] != tile_id[:-1]
        # slot of a tile's first piece = pieces closed before its first entry
        meta[tile_id[first], 0] = slot_base + (np.cumsum(last) - last)[first]
        tiles_col.append(ent_col), tiles_val.append(ent_val), tiles_meta.append(meta), tiles_src.append(ent_src)
        flag_orig.append(ent_off[order][last]), flag_slot.append(slot[last])
        slot_base += int(last.sum())
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)   # noqa: E731
    f_orig, f_slot = cat(flag_orig, np.int64), cat(flag_slot, np.int64)
    by_entry = np.argsort(f_orig, kind='stable')               # row-major, ascending column: a row's pieces in order
    f_orig, f_slot = f_orig[by_entry], f_slot[by_entry]
    f_row = np.searchsorted(rowptr, f_orig, side='right') - 1
    cnt = np.bincount(f_row, minlength=n_rows) if len(f_row) else np.zeros(n_rows, dtype=np.int64)
    seg_rows = np.nonzero(cnt)[0]
    row_slot_ptr = np.zeros(len(seg_rows) + 1, dtype=np.int64)
    np.cumsum(cnt[seg_rows], out=row_slot_ptr[1:])
    direct[seg_rows] = False
    ent_col = cat(tiles_col, np.int64)
    return {
        'tile_entries': T, 'n_slots': int(slot_base),
        'tile_meta': np.ascontiguousarray(cat(tiles_meta, np.int64).reshape(-1, 2).astype(np.int32)),
        'ent_col': (ent_col & 0x7fffffff).astype(np.int32),
        'ent_flags': np.packbits((ent_col >> 31).astype(np.uint8).reshape(-1, 64), axis=1, bitorder='little').view(np.int64).reshape(-1),   # uint64 words, carried as int64
        'ent_val': cat(tiles_val, np.float32),
        'ent_src': cat(tiles_src, np.int32),      # offset in colidx/vals each stream entry was copied from (padding: 0)
        'seg_rows': seg_rows.astype(np.int32), 'row_slot_ptr': row_slot_ptr.astype(np.int32),
        'row_slots': f_slot.astype(np.int32), 'direct_rows': np.nonzero(direct)[0].astype(np.int32),
    }


def train_mask_csr(train_u, train_i, n_users):
    """CSR over all users of their train items (ascending): the device form of
    base_model.py:257 `train_user_dict[batch_users].explode()`.  Returns (rowptr int64 [U+1], items int32)."""
    u = _as_index(train_u, 'train_u')
    i = _as_index(train_i, 'train_i')
    rowptr = np.zeros(int(n_users) + 1, dtype=np.int64)
    np.cumsum(np.bincount(u, minlength=int(n_users)), out=rowptr[1:])
    # pairs that already arrive sorted by (user, item) -- the synthetic generator's and a sorted train frame's order -- need no
    # sort (a lexsort of config 4's 100 M pairs is most of a minute)
    if len(u) < 2 or (np.all(u[1:] >= u[:-1]) and np.all((i[1:] >= i[:-1]) | (u[1:] != u[:-1]))):
        return rowptr, i.astype(np.int32)
    order = np.lexsort((i, u))
    return rowptr, i[order].astype(np.int32)
