"""K-layer LightGCN propagation on the HIP path (single GPU, or one rank's row block of a sharded graph).

Replaces the loop of TextGCN/base_model.py:93-106 (representation): K x torch.sparse.mm (:148), the
torch.cat of the two embedding tables (:91) and torch.mean(torch.stack(...)) (:157).  Device memory is
plain torch tensors; the arithmetic is tgcn_spmm_csr_f32 / tgcn_spmm_segmented_f32 (include/tgcn.h).
"""
import ctypes

import numpy as np
import torch

from . import _capi
from .graph import NormGraph, row_groups, segment_plan_arrays, split_plan_arrays

DEFAULT_SPLIT_THRESHOLD = 1024
L2_SHARE_BYTES = 3 << 20        # of an XCD's 4 MB L2 that a gathered table can count on next to the streaming traffic
SEGMENT_CLASSES = 8             # one column-block class per XCD
SEGMENT_TILE_ENTRIES = 256      # entries per tile wave.  One launch for tiles + direct rows: 512..1024 within 1 %, 256 -8 %, 2048 -30 %;
                                # two launches (below): 192..256 best (0.555-0.557 ms per config-2 forward against 0.592), 128 / 320 +2 %
SEGMENT_MIN_ROW_LEN = 48        # shorter rows stay direct: < 6 entries per block do not pay for a workspace slot (0 / 16: +7 %; 32..64 within 1 %)
GROUP_TARGET_ENTRIES = 64       # stored entries per row group (one wave): four batches of 16 gathers
GROUP_SINGLE_LEN = 32           # rows of this many entries or more are a group of their own
SEGMENT_TWO_PHASE = True        # tiles alone (the XCDs' L2s hold only their block of the gathered table), then the piece reduce and the
                                # direct rows side by side in a second launch (tgcn_spmm_segmented_f32 flags bit 16): config 2 -6 %, same bits


HOT_MIN_TABLE_BYTES = 256 << 20  # hot-row mode applies to gather tables beyond the Infinity Cache
HOT_WINDOW_BYTES = 8 << 20      # hot-row mode: width of a column block of the gathered table (config 4: 104 .. 312 blocks within 1 %)
HOT_ENTRIES_PER_BLOCK = 4       # ... a row is cut when it leaves at least this many entries per block (config 4: 500 .. 800 entries best)


def segment_blocks_auto(rowptr, colidx, spec, d):
    """Column blocks for one (row_begin, row_end, col_lo, col_hi) row range: 0 = keep it on the row-group kernel; n = n XCD-affine
    blocks (one class per XCD) for every row of at least SEGMENT_MIN_ROW_LEN entries; (n, classes, min_row_len) = the hot-row mode
    of tables beyond the Infinity Cache (only rows of at least min_row_len entries are cut; see below).  Segmenting pays when the gathered table misses an XCD's L2 but an eighth of it fits,
    the hottest rows do not already serve most gathers from L2 (Zipf-popular items do), and a row leaves enough
    entries per block to amortise the extra workspace round trip (measured on BASELINE config 2: item rows, mean 100
    entries, +16 %; user rows, mean 50 entries over a Zipf table, -30 %)."""
    r0, r1, c0, c1 = spec
    row_bytes = 4 * d
    table = (c1 - c0) * row_bytes
    if d not in (64, 128, 256) or table <= L2_SHARE_BYTES:
        return 0
    a, b = int(rowptr[r0]), int(rowptr[r1])
    if table > SEGMENT_CLASSES * (L2_SHARE_BYTES + (L2_SHARE_BYTES >> 2)):
        if table <= HOT_MIN_TABLE_BYTES:      # between the two rules nothing measured pays (config 3: +-2 % whatever the blocks)
            return 0
        # HOT-ROW mode (round 4; config 4's item rows over the 1.28 GB user table).  No eighth of such a table fits an L2 -- but the
        # long rows of a Zipf graph are many: 22 k item rows of >= 640 entries hold 38 % of config 4's item-row entries, and inside a
        # window of ~32 k users those rows reference each user row ~5 times.  Cutting ONLY them at windows of 8 MB (pieces summed in
        # column order, as every segmented row) makes those re-references hits instead of 256-byte row fetches from HBM: 22.8 ->
        # 21.6 ms per config-4 forward (profiles/r04_experiments.md section 7).  Short rows stay one chain: a piece per window
        # would cost them more than their gathers.
        nb = min(1024, ((-(-table // HOT_WINDOW_BYTES) + 7) // 8) * 8)
        mrl = HOT_ENTRIES_PER_BLOCK * nb
        lens = np.diff(np.asarray(rowptr[r0:r1 + 1], dtype=np.int64))
        hot_entries = int(lens[lens >= mrl].sum())
        if hot_entries < max(1 << 20, 0.1 * (b - a)):
            return 0
        return (int(nb), SEGMENT_CLASSES, int(mrl))
    if b - a < (1 << 20) or (b - a) / max(r1 - r0, 1) < 8 * SEGMENT_CLASSES:
        return 0
    counts = np.bincount(np.asarray(colidx[a:b], dtype=np.int64) - c0, minlength=c1 - c0)
    hot = np.sort(counts)[::-1][:L2_SHARE_BYTES // row_bytes].sum()
    return SEGMENT_CLASSES if hot < 0.5 * (b - a) else 0


class EdgeValues:
    """A per-call replacement of a DeviceCSR's stored values on the same structure (edge dropout, transposed values),
    together with its copy in the segment plan's stream order.  The copy is made once -- by tgcn_dropout_values_f32 in the
    same launch that draws the mask, or by the first spmm call that needs it -- and then reused by every layer of the
    forward / backward that passes the same object."""

    def __init__(self, vals, seg_vals=None):
        self.vals = vals
        self.seg_vals = seg_vals


class DeviceCSR:
    """CSR of a contiguous block of rows of A on one device (+ optional long-row split plan and segment plan)."""

    def __init__(self, rowptr, colidx, vals, n_src_rows, device, split_threshold=None, block_specs=None,
                 order_rows=True, segment=None):
        """block_specs: optional list of (row_begin, row_end, col_lo, col_hi) covering all rows once -- row ranges
        whose entries fall in one column range (user rows x item columns, item rows x user columns).
        segment: None (off), 'auto' (segment_blocks_auto per spec) or a list of block counts per spec: enables the
        XCD-affine segmented kernel (tgcn_spmm_segmented_f32) for widths 64/128/256."""
        rowptr = np.asarray(rowptr, dtype=np.int64)
        if rowptr[-1] >= np.iinfo(np.int32).max:
            raise ValueError('row block has >= 2^31 entries')
        self.n_rows = len(rowptr) - 1
        self.n_src_rows = int(n_src_rows)
        self.nnz = int(rowptr[-1])
        self.device = _capi.resolve_device(device)
        self.rowptr = torch.from_numpy(rowptr.astype(np.int32)).to(self.device)
        if self.nnz == 0:   # a graph without interactions: keep non-NULL device arrays for the ABI's pointer checks
            colidx, vals = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.float32)
        self.colidx = torch.from_numpy(np.ascontiguousarray(colidx, dtype=np.int32)).to(self.device)
        self.vals = torch.from_numpy(np.ascontiguousarray(vals, dtype=np.float32)).to(self.device)
        # longest rows first: the launch's tail is its last long row (scheduling only; results do not depend on it)
        lens = np.diff(rowptr)
        self.row_order = None
        # Longest-first hand-out pays when the launch's tail -- one wave finishing its longest row -- is a visible
        # share of the launch: (longest work item) x (waves the chip runs at once) vs total entries.  On huge graphs
        # (config 4: 0.04) it only scatters the CSR reads (measured 5 % slower), so rows keep their natural order.
        longest = min(int(lens.max()) if len(lens) else 0, split_threshold or (1 << 30))
        self._rowptr_host = rowptr
        self._groups = {}
        self._longest_first = bool(order_rows and self.nnz and longest * 8192 > 0.1 * self.nnz)
        if self._longest_first:
            order = np.argsort(-lens, kind='stable')
            if block_specs:
                # one phase per row block that shares a gather table (item rows, then user rows), longest first inside
                # a phase: the rows in flight then gather from ONE table, not from both
                parts = []
                for (r0, r1, _c0, _c1) in sorted(block_specs, key=lambda sp: -sp[0]):
                    parts.append(r0 + np.argsort(-lens[r0:r1], kind='stable'))
                order = np.concatenate(parts)
            self.row_order = torch.from_numpy(order.astype(np.int32)).to(self.device)
        self._plan_host = split_plan_arrays(rowptr, split_threshold) if split_threshold else None
        self._plan_dev = None
        self._plan_struct = {}
        if self._plan_host is not None:
            self._plan_dev = {k: torch.from_numpy(v).to(self.device) for k, v in self._plan_host.items()
                              if k != 'threshold'}
        self._block_specs = block_specs
        self._split_threshold = split_threshold
        self._host = (rowptr, np.asarray(colidx), lens) if block_specs else None
        self._segment_mode = segment
        self.segment_blocks = None    # per (d): block counts per block_spec
        self.segment_tile = SEGMENT_TILE_ENTRIES
        self.segment_min_row_len = SEGMENT_MIN_ROW_LEN
        self._segment_plans = {}
        self.use_groups = True        # False: one wave per row everywhere (tools / A-B timing; same bits)

    @property
    def n_chunks(self):
        return 0 if self._plan_host is None else len(self._plan_host['chunk_beg'])

    def plan(self, d):
        """ctypes pointer to a tgcn_split_plan_t for embedding width d (workspace allocated once per d)."""
        if self._plan_host is None:
            return None
        if d not in self._plan_struct:
            ws = torch.empty((self.n_chunks, d), dtype=torch.float32, device=self.device)
            p = self._plan_dev
            st = _capi.SplitPlanStruct(self._plan_host['threshold'], self.n_chunks, len(self._plan_host['long_rows']), 0,
                                       p['chunk_beg'].data_ptr(), p['chunk_end'].data_ptr(), p['long_rows'].data_ptr(),
                                       p['long_chunk_ptr'].data_ptr(), ws.data_ptr())
            self._plan_struct[d] = (st, ws)
        return ctypes.byref(self._plan_struct[d][0])

    def groups(self, d, exact=False):
        """device int32 [n_groups, 2] row groups for tgcn_spmm_groups_f32 at width d (None: width / table size not supported).
        exact: no split plan -- every row, however long, is in a group (one chain per row)."""
        if d not in (64, 128, 256) or self.n_src_rows * d * 4 >= (1 << 32) or self.n_rows == 0:
            return None
        thr = None if (exact or self._plan_host is None) else self._plan_host['threshold']
        key = (4 if d == 256 else 8, thr)
        if key not in self._groups:
            phases = sorted({sp[0] for sp in self._block_specs} - {0}) if self._block_specs else None
            g = row_groups(self._rowptr_host, None, thr, key[0], GROUP_TARGET_ENTRIES, longest_first=self._longest_first,
                           single_len=GROUP_SINGLE_LEN, phases=phases)
            self._groups[key] = torch.from_numpy(g).to(self.device) if len(g) else torch.zeros((1, 2), dtype=torch.int32, device=self.device)
            self._groups[key, 'n'] = len(g)
        return self._groups[key], self._groups[key, 'n']

    def configure_segments(self, blocks_per_spec, tile_entries=SEGMENT_TILE_ENTRIES, min_row_len=SEGMENT_MIN_ROW_LEN):
        """Set the XCD-affine segmentation by hand: blocks_per_spec[i] column blocks (a multiple of 8) for
        block_specs[i]; 0 keeps that row range on the one-wave-per-row path.  ('auto' picks these per width.)"""
        if not self._block_specs or len(blocks_per_spec) != len(self._block_specs):
            raise ValueError('one block count per block_spec is required')
        self._segment_mode = [tuple(int(x) for x in b) if isinstance(b, (tuple, list)) else int(b) for b in blocks_per_spec]
        self.segment_tile = int(tile_entries)
        self.segment_min_row_len = int(min_row_len)
        self._segment_plans = {}

    def segment_ent_src(self, d):
        """int32 device tensor: the CSR entry each slot of the segment plan's streams copies (None: no plan at width d)."""
        if self.segment_plan(d) is None:
            return None
        return self._segment_plans[d][0][2]['ent_src']

    def segment_plan(self, d, vals=None):
        """ctypes pointer to a tgcn_segment_plan_t for width d, or None when no row range is segmented at this width.
        vals: per-call replacement of the stored values (a tensor laid out as self.vals, or an EdgeValues)."""
        if self._segment_mode is None or not self._block_specs or d not in (64, 128, 256) or self.nnz == 0:
            return None
        if d not in self._segment_plans:
            rowptr, colidx, lens = self._host
            if self._segment_mode == 'auto':
                blocks = [segment_blocks_auto(rowptr, colidx, sp, d) for sp in self._block_specs]
            else:
                blocks = list(self._segment_mode)
            entry = None
            if any(blocks):
                # a block count may be (n_blocks, n_classes): see segment_plan_arrays
                phases = [(r0, r1, c0, c1) + (tuple(nb) if isinstance(nb, (tuple, list)) else (nb,))
                          for (r0, r1, c0, c1), nb in zip(self._block_specs, blocks) if nb]
                h = segment_plan_arrays(rowptr, colidx, self.vals.cpu().numpy(), phases, self.segment_tile,
                                        min_row_len=self.segment_min_row_len)
                n_dg = 0
                if self.n_src_rows * d * 4 < (1 << 32) and len(h['direct_rows']):
                    # natural row order, as the one-wave-per-row form hands the direct rows out (the tiles' pieces are reduced
                    # beside them: nothing here is a tail)
                    h['direct_groups'] = row_groups(rowptr, h['direct_rows'], None, 4 if d == 256 else 8, GROUP_TARGET_ENTRIES,
                                                    single_len=GROUP_SINGLE_LEN, phases=[sp[0] for sp in self._block_specs if sp[0]])
                    n_dg = len(h['direct_groups'])
                    if n_dg == len(h['direct_rows']):      # every direct row a group of its own (config 2: 50-entry user rows):
                        n_dg = 0                           # the one-wave-per-row launch does the same work without the group list
                        del h['direct_groups']
                dv = {k: torch.from_numpy(v).to(self.device) for k, v in h.items() if isinstance(v, np.ndarray)}
                ws = torch.empty((max(h['n_slots'], 1), d), dtype=torch.float32, device=self.device)
                st = _capi.SegmentPlanStruct(len(h['tile_meta']), h['tile_entries'], len(h['seg_rows']), len(h['direct_rows']),
                                             h['n_slots'], 0, dv['tile_meta'].data_ptr(), dv['ent_col'].data_ptr(),
                                             dv['ent_val'].data_ptr(), dv['ent_flags'].data_ptr(), dv['seg_rows'].data_ptr(),
                                             dv['row_slot_ptr'].data_ptr(), dv['row_slots'].data_ptr(),
                                             dv['direct_rows'].data_ptr(), ws.data_ptr(),
                                             dv['direct_groups'].data_ptr() if n_dg else None, n_dg, 0)
                entry = (st, ws, dv, h)
            self._segment_plans[d] = (entry, blocks)
        entry, blocks = self._segment_plans[d]
        self.segment_blocks = blocks
        if entry is None:
            return None
        if vals is None and (self.use_groups or not entry[0].n_direct_groups):
            return ctypes.byref(entry[0])
        if vals is None:       # A/B switch: the plan without its direct-row groups
            tmp = _capi.SegmentPlanStruct.from_buffer_copy(entry[0])
            tmp.n_direct_groups = 0
            self._segment_keep = (tmp, None)
            return ctypes.byref(tmp)
        # per-call values (edge dropout, transposed values): the plan's streams get their own gathered copy, made once per
        # EdgeValues object (K forward + K backward launches share it)
        st, dv = entry[0], entry[2]
        if isinstance(vals, EdgeValues):
            if vals.seg_vals is None:
                vals.seg_vals = vals.vals.index_select(0, dv['ent_src'])
            ev = vals.seg_vals
        else:
            ev = vals.index_select(0, dv['ent_src'])
        tmp = _capi.SegmentPlanStruct.from_buffer_copy(st)
        tmp.ent_val = ev.data_ptr()
        if not self.use_groups:
            tmp.n_direct_groups = 0
        self._segment_keep = (tmp, ev)     # alive until the next call on this CSR (the launch is stream-ordered after the gather)
        return ctypes.byref(tmp)


def _check_dense(t, name, device, rows=None, d=None):
    if not isinstance(t, torch.Tensor) or t.dtype != torch.float32:
        raise TypeError(f'{name} must be a float32 torch tensor')
    if t.device != device:
        raise ValueError(f'{name} is on {t.device}, expected {device}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if t.dim() != 2 or (rows is not None and t.shape[0] != rows) or (d is not None and t.shape[1] != d):
        raise ValueError(f'{name} has shape {tuple(t.shape)}, expected ({rows}, {d})')


def spmm(csr, x, y=None, acc_in=None, acc_out=None, acc_div=1.0, exact=False, variant=_capi.SPMM_AUTO, unroll=0,
         vals=None, segmented=None):
    """One layer: y = A_block . x, optionally acc_out = (acc_in + y) / acc_div (see tgcn_spmm_csr_f32).

    x [n_src_rows, d]; y / acc_in / acc_out [n_rows, d] (y or acc_out may be None).  exact=True ignores the
    long-row and segment plans: every row is one sequential fmaf chain, bit-identical to the reference's CPU kernel.
    vals: optional replacement of the stored values on the same structure (edge dropout, transposed values).
    segmented: None = use the XCD-affine segmented kernel (tgcn_spmm_segmented_f32) when the CSR has a segment plan
    for this width (DeviceCSR(segment=...)), False = never, True = require it."""
    dev = csr.device
    if dev.type != 'cuda':
        raise RuntimeError('textgcn_amd kernels run on a ROCm GPU only (device is %s)' % dev)
    _check_dense(x, 'x', dev, csr.n_src_rows)
    d = x.shape[1]
    for t, name in ((y, 'y'), (acc_in, 'acc_in'), (acc_out, 'acc_out')):
        if t is not None:
            _check_dense(t, name, dev, csr.n_rows, d)
    if y is not None and y.data_ptr() == x.data_ptr():
        raise ValueError('y must not alias x')
    own_vals = vals is None
    bundle = vals if isinstance(vals, EdgeValues) else None
    if bundle is not None:
        vals = bundle.vals
    if vals is None:
        vals = csr.vals
    elif vals.dtype != torch.float32 or vals.numel() != max(csr.nnz, 1) or vals.device != dev or not vals.is_contiguous():
        raise ValueError('vals must be a contiguous float32 device tensor with one entry per stored element')
    # the segment plan carries its own copy of the stored values: a per-call `vals` (dropout) is gathered into it
    seg = None
    if segmented is not False and not exact and variant == _capi.SPMM_AUTO:
        seg = csr.segment_plan(d, None if own_vals else (bundle or vals))
    if segmented is True and seg is None:
        raise ValueError('segmented=True but this CSR has no segment plan for the call (width, exact or variant)')
    if seg is not None:
        rc = _capi.lib().tgcn_spmm_segmented_f32(
            seg, _capi.ptr(csr.rowptr), _capi.ptr(csr.colidx), _capi.ptr(vals), csr.n_rows, _capi.ptr(x), csr.n_src_rows, d,
            _capi.ptr(y), _capi.ptr(acc_in), _capi.ptr(acc_out), float(acc_div), ((unroll & 0xff) << 8) | ((1 << 16) if SEGMENT_TWO_PHASE else 0),
            _capi.current_stream(dev))
        _capi.check(rc, 'tgcn_spmm_segmented_f32')
        return y if y is not None else acc_out
    plan = None if exact else csr.plan(d)
    grp = csr.groups(d, exact) if (variant == _capi.SPMM_AUTO and csr.use_groups) else None
    if grp is not None:
        rc = _capi.lib().tgcn_spmm_groups_f32(
            _capi.ptr(csr.rowptr), _capi.ptr(csr.colidx), _capi.ptr(vals), csr.n_rows, _capi.ptr(x), csr.n_src_rows, d,
            _capi.ptr(y), _capi.ptr(acc_in), _capi.ptr(acc_out), float(acc_div), plan, _capi.ptr(grp[0]), grp[1],
            (unroll & 0xff) << 8, _capi.current_stream(dev))
        _capi.check(rc, 'tgcn_spmm_groups_f32')
        return y if y is not None else acc_out
    rc = _capi.lib().tgcn_spmm_csr_f32(
        _capi.ptr(csr.rowptr), _capi.ptr(csr.colidx), _capi.ptr(vals), csr.n_rows, _capi.ptr(x), csr.n_src_rows, d,
        _capi.ptr(y), _capi.ptr(acc_in), _capi.ptr(acc_out), float(acc_div), plan, _capi.ptr(csr.row_order),
        (variant & 0xff) | ((unroll & 0xff) << 8),
        _capi.current_stream(dev))
    _capi.check(rc, 'tgcn_spmm_csr_f32')
    return y if y is not None else acc_out


class Propagator:
    """Whole-graph K-layer forward on one GPU.

    forward(e0) -> combined [N, d]  ==  mean(E0..EK)  (or EK when single=True, base_model.py:159-164),
    E(k+1) = A . E(k).  Buffers are allocated once per embedding width and reused.
    """

    def __init__(self, graph: NormGraph, device, split_threshold=DEFAULT_SPLIT_THRESHOLD, segment='auto'):
        self.graph = graph
        self.device = _capi.resolve_device(device)
        u, n = graph.n_users, graph.n
        # A is bipartite: user rows hold item columns and vice versa
        specs = [(0, u, u, n), (u, n, 0, u)]
        self.csr = DeviceCSR(graph.rowptr, graph.colidx, graph.vals, graph.n, self.device, split_threshold,
                             block_specs=specs, segment=segment)
        self._buf = {}

    def buffers(self, d):
        if d not in self._buf:
            n = self.graph.n
            self._buf[d] = tuple(torch.empty((n, d), dtype=torch.float32, device=self.device) for _ in range(3))
        return self._buf[d]

    def forward(self, e0, n_layers, single=False, exact=False, out=None, keep_layers=False, variant=_capi.SPMM_AUTO,
                unroll=0, vals=None, segmented=None):
        n = self.graph.n
        _check_dense(e0, 'e0', self.device, n)
        d = e0.shape[1]
        ping, pong, acc = self.buffers(d)
        if out is None:
            out = torch.empty((n, d), dtype=torch.float32, device=self.device)
        layers = [e0] if keep_layers else None
        if n_layers == 0:
            out.copy_(e0)
            return (out, layers) if keep_layers else out

        x = e0
        for k in range(1, n_layers + 1):
            last = k == n_layers
            if keep_layers:
                y = torch.empty((n, d), dtype=torch.float32, device=self.device)
            else:
                y = ping if (k & 1) else pong
            if single:
                spmm(self.csr, x, y=out if (last and not keep_layers) else y, exact=exact, variant=variant, unroll=unroll,
                     vals=vals, segmented=segmented)
                if last and keep_layers:
                    out.copy_(y)
            else:
                # running layer sum: acc = E0 + E1 (k = 1), acc += Ek, divided by K+1 on the last layer;
                # the last layer's own Y is not needed (no store)
                spmm(self.csr, x, y=None if (last and not keep_layers) else y, acc_in=e0 if k == 1 else acc,
                     acc_out=out if last else acc, acc_div=float(n_layers + 1) if last else 1.0, exact=exact,
                     variant=variant, unroll=unroll, vals=vals, segmented=segmented)
            if keep_layers:
                layers.append(y)
            x = y
        return (out, layers) if keep_layers else out
