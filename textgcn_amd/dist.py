"""Row-sharded K-layer propagation over P ranks (one process per GPU) -- SURVEY.md §8(e).

The reference is single-device (no collective anywhere); this is new.  Y[r,:] depends on row r
of A and the whole previous layer, so rows are partitioned 1-D and each layer ends with an all-gather of the
freshly propagated blocks.  A is bipartite ([[0, R],[R^T, 0]]): user rows read only item rows of X and vice
versa, so a layer is two half-steps and the all-gather of the user block runs on RCCL's stream while the item
half-step computes.

Partition.  Users and items are cut separately into P contiguous blocks balanced by stored entries (item
degrees are Zipf-skewed: equal row counts would give uneven half-steps); `balance='rows'` gives equal row counts.
Every block is padded with empty rows to the largest block, so the collective stays an even all-gather.

Layout.  A rank's block is cut again into C row chunks of cb rows; the layer table is stored chunk-major,
[chunk][rank][cb rows], so chunk c of all ranks is ONE contiguous slab and `all_gather_into_tensor` of that chunk
writes straight into the table (no packing).  A half-step is C SpMM launches, and chunk c's all-gather starts as
soon as its launch is enqueued: the gather of chunk c runs under the SpMM of chunks c+1.. and under the whole
other half-step.  Column ids of the local CSR blocks are remapped to table rows once, on the host; the order of a
row's entries is kept, so a row's fmaf chain -- and with it every output bit -- is the single-GPU one whatever
P, C and the balance are.

Collectives.  backend "nccl" (= RCCL over xGMI) -> torch.distributed's communicator, or, with collective='capi',
libtgcn's own entry point tgcn_allgather_rows on a communicator made by tgcn_comm_init_rank (include/tgcn.h) on a
side stream.  Any other backend (gloo) is the CPU / one-GPU rehearsal used by the tests: CPU tensors are gathered
asynchronously by gloo itself, device tensors are staged through host memory.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _capi
from ._capi import resolve_device
from .graph import NormGraph
from .propagate import DEFAULT_SPLIT_THRESHOLD, DeviceCSR, spmm


def padded_layout(n_users, n_items, world):
    """Equal-row blocks (balance='rows', one chunk): (rows per user block, rows per item block, U_pad, I_pad)."""
    bu = -(-n_users // world)
    bi = -(-n_items // world)
    return bu, bi, bu * world, bi * world


def equal_row_bounds(n, world, offset=0):
    b = -(-n // world)
    return offset + np.minimum(np.arange(world + 1, dtype=np.int64) * b, n)


class BlockLayout:
    """Padded chunk-major placement of one node kind (users or items) in the layer table.

    bounds [P+1]: contiguous global id ranges per rank.  Rank p's row o (0-based inside its block) lives in chunk
    c = o // cb at table row  c * P * cb + p * cb + o % cb;  cb = ceil(max block / C), rows per rank b = C * cb."""

    def __init__(self, bounds, chunks=1):
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.world = len(self.bounds) - 1
        sizes = np.diff(self.bounds)
        if self.world < 1 or np.any(sizes < 0):
            raise ValueError('bounds must be ascending')
        self.chunks = max(1, int(chunks))
        self.cb = max(1, -(-int(sizes.max()) // self.chunks))
        self.b = self.cb * self.chunks
        self.n_pad = self.b * self.world
        self.n = int(self.bounds[-1] - self.bounds[0])

    def table_rows(self, ids):
        """global ids (same origin as bounds) -> table rows"""
        ids = np.asarray(ids, dtype=np.int64)
        p = np.searchsorted(self.bounds, ids, side='right') - 1
        p = np.minimum(p, self.world - 1)
        o = ids - self.bounds[p]
        c, w = np.divmod(o, self.cb)
        return c * (self.world * self.cb) + p * self.cb + w

    def size(self, rank):
        return int(self.bounds[rank + 1] - self.bounds[rank])

    def chunk_slab(self, c):
        """table rows of chunk c of all ranks (the all-gather's output)"""
        return slice(c * self.world * self.cb, (c + 1) * self.world * self.cb)

    def my_slab(self, rank, c):
        """table rows of this rank's chunk c (the all-gather's input, a view of chunk_slab(c))"""
        s = c * self.world * self.cb + rank * self.cb
        return slice(s, s + self.cb)


def _hip_spmm(csr, x, **kw):
    return spmm(csr, x, **kw)


class _Done:
    def wait(self):
        return None


class _EventWork:
    """completion of a collective enqueued on a side stream: wait() orders the CURRENT stream after it"""

    def __init__(self, event, device):
        self.event, self.device = event, device

    def wait(self):
        torch.cuda.current_stream(self.device).wait_event(self.event)


class CapiComm:
    """RCCL communicator owned by libtgcn (tgcn_comm_init_rank / tgcn_allgather_rows, include/tgcn.h).  The 128-byte id is
    made by rank 0 and handed to the other ranks by `broadcast` (any out-of-band channel; the default uses the
    torch.distributed group the process already has)."""

    def __init__(self, rank, world, device, group=None, broadcast=None):
        self.rank, self.world, self.device = rank, world, device
        lib = _capi.lib()
        buf = ctypes.create_string_buffer(_capi.TGCN_COMM_ID_BYTES)
        if rank == 0:
            _capi.check(lib.tgcn_comm_unique_id(buf), 'tgcn_comm_unique_id')
        ident = bytes(buf.raw)
        if world > 1:
            if broadcast is None:
                box = [ident]
                dist.broadcast_object_list(box, src=0, group=group)
                ident = box[0]
            else:
                ident = broadcast(ident)
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            _capi.check(lib.tgcn_comm_init_rank(ctypes.byref(self._comm), world, rank, ident), 'tgcn_comm_init_rank')
        self.stream = torch.cuda.Stream(device)

    def all_gather(self, full, local):
        """full [P*b, d] <- every rank's local [b, d]; enqueued on the communicator's side stream after the work already
        on the current stream.  Returns a handle whose wait() orders the current stream after the collective."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        rc = _capi.lib().tgcn_allgather_rows(self._comm, _capi.ptr(local), _capi.ptr(full), local.shape[0], local.shape[1],
                                             ctypes.c_void_p(self.stream.cuda_stream))
        _capi.check(rc, 'tgcn_allgather_rows')
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _EventWork(ev, self.device)

    def close(self):
        if self._comm:
            torch.cuda.synchronize(self.device)
            _capi.lib().tgcn_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()


class ShardedPropagator:
    """One rank's share of the K-layer forward.

    forward(e0_users_local, e0_items_local) -> (users_local [bu, d], items_table [I_pad, d]): the rank keeps its own
    users (they are scored where they live) and the gathered item table of the layer mean, in table order
    (`items_in_order` puts it back in item-id order).

    LIFETIME of the two results: both are views of this object's reusable buffers (`users_local` of the output block,
    `items_table` of the ping-pong layer table the last layer was written to) and are valid until the NEXT forward() on this
    object, which overwrites them from its first launch on.  A consumer that keeps a result across another forward -- scoring
    overlapped with the next step, a result held through a timing loop -- passes copy=True (or clones): the gathered table is
    then copied out once (I_pad x d x 4 bytes) and the buffers are free again.

    collective: 'torch' (default; 'auto' means the same) = torch.distributed's communicator, 'capi' = libtgcn's own
    (tgcn_comm_init_rank / tgcn_allgather_rows).  The choice is an argument only: no environment variable changes it.

    record_events = True makes forward() bracket every half-step's wait-for-gather and its SpMM launches with HIP events on the
    launch stream; `layer_times()` then returns, per layer, the milliseconds that stream spent computing and the milliseconds it
    sat waiting for an all-gathered block (bench.py prints them so that a multi-GPU number can be attributed).
    """

    def __init__(self, graph: NormGraph, rank, world, device, group=None, split_threshold=DEFAULT_SPLIT_THRESHOLD,
                 local_spmm=None, balance='nnz', chunks=1, collective='auto', force_collective=False, segment='auto'):
        """segment: 'auto' (default) = a row chunk whose gather table -- the other node kind's part of the layer table -- is a few
        L2 sizes runs the XCD-affine segmented kernels, by the rule a whole-graph Propagator uses (propagate.segment_blocks_auto:
        config 2 cut in two keeps its item rows segmented; config 4's tables are far too large either way); None = never.  Rows a
        segment plan cuts are summed piecewise (normwise 1e-6 against the one-chain result, as on one GPU), so with segments a
        P-rank result equals the 1-rank result to rounding, not bit for bit; exact=True ignores the plans and stays bit-identical."""
        self.rank, self.world = int(rank), int(world)
        self.device = resolve_device(device)
        self.group = group
        self.n_users, self.n_items = graph.n_users, graph.n_items
        if balance == 'nnz':
            ub, ib = graph.partition(self.world)
        elif balance == 'rows':
            ub = equal_row_bounds(graph.n_users, self.world)
            ib = equal_row_bounds(graph.n_items, self.world, graph.n_users)
        else:
            raise ValueError("balance must be 'nnz' or 'rows'")
        self.lay_u = BlockLayout(ub, chunks)
        self.lay_i = BlockLayout(np.asarray(ib) - graph.n_users, chunks)      # item ids 0..I-1
        self.bu, self.bi = self.lay_u.b, self.lay_i.b
        self.u_pad, self.i_pad = self.lay_u.n_pad, self.lay_i.n_pad
        self.n_pad = self.u_pad + self.i_pad
        self._spmm = local_spmm or _hip_spmm   # tests inject a CPU stand-in to rehearse the exchange logic
        self.uses_collective = self.world > 1 or force_collective
        self.backend = dist.get_backend(group) if self.uses_collective else 'none'
        self._capi_comm = None
        if self.uses_collective and self.backend == 'nccl':
            if collective == 'auto':
                collective = 'torch'
            if collective == 'capi':
                self._capi_comm = CapiComm(self.rank, self.world, self.device, group)
            elif collective != 'torch':
                raise ValueError("collective must be 'auto', 'torch' or 'capi'")
        # local CSR blocks, one per row chunk, column ids remapped to table rows
        self._segment = segment if self.device.type == 'cuda' else None
        self.csr_u = self._chunk_csrs(graph, self.lay_u, 0, lambda c: self.u_pad + self.lay_i.table_rows(c - graph.n_users),
                                      split_threshold, (self.u_pad, self.n_pad))
        self.csr_i = self._chunk_csrs(graph, self.lay_i, graph.n_users, self.lay_u.table_rows, split_threshold, (0, self.u_pad))
        self.nnz_local = sum(c.nnz for c in self.csr_u) + sum(c.nnz for c in self.csr_i)
        self._buf = {}
        self._item_order = None
        self.record_events = False
        self._events = None
        self._skip = None            # phase_times(): 'spmm' / 'gather' leaves that half of forward() out

    # ------------------------------------------------------------------ construction
    def segment_note(self):
        """which of this rank's row chunks run the segmented kernels (known once a forward has built the plans)"""
        blocks = [c.segment_blocks for c in self.csr_u + self.csr_i if c.segment_blocks]
        on = sum(1 for b in blocks if any(b))
        return 'none' if not on else f'{on} of {len(self.csr_u) + len(self.csr_i)} row chunks on this rank (tgcn_spmm_segmented_f32)'

    def _chunk_csrs(self, graph, lay, row_origin, remap, split_threshold, col_range):
        g0 = row_origin + int(lay.bounds[self.rank])
        g1 = row_origin + int(lay.bounds[self.rank + 1])
        out = []
        for c in range(lay.chunks):
            r0 = min(g0 + c * lay.cb, g1)
            r1 = min(r0 + lay.cb, g1)
            rp, ci, va = graph.row_block(r0, r1)
            rp = np.asarray(rp, dtype=np.int64)
            if len(rp) - 1 < lay.cb:   # pad rows: empty
                rp = np.concatenate([rp, np.full(lay.cb - (len(rp) - 1), rp[-1], dtype=np.int64)])
            cols = remap(np.asarray(ci, dtype=np.int64)) if len(ci) else np.zeros(0, dtype=np.int64)
            out.append(DeviceCSR(rp, cols, va, self.n_pad, self.device, split_threshold,
                                 block_specs=[(0, lay.cb, col_range[0], col_range[1])] if self._segment else None,
                                 segment=self._segment))
        return out

    def user_range(self, r=None):
        r = self.rank if r is None else r
        return int(self.lay_u.bounds[r]), int(self.lay_u.bounds[r + 1])

    def item_range(self, r=None):
        r = self.rank if r is None else r
        return int(self.lay_i.bounds[r]), int(self.lay_i.bounds[r + 1])

    # ------------------------------------------------------------------ helpers
    def local_e0(self, e0_full):
        """Cut this rank's padded E0 blocks out of a full [N, d] table (host or device tensor)."""
        d = e0_full.shape[1]
        u0, u1 = self.user_range()
        i0, i1 = self.item_range()
        eu = torch.zeros((self.bu, d), dtype=torch.float32, device=self.device)
        ei = torch.zeros((self.bi, d), dtype=torch.float32, device=self.device)
        eu[:u1 - u0] = e0_full[u0:u1].to(self.device)
        ei[:i1 - i0] = e0_full[self.n_users + i0:self.n_users + i1].to(self.device)
        return eu, ei

    def buffers(self, d):
        if d not in self._buf:
            mk = lambda n: torch.zeros((n, d), dtype=torch.float32, device=self.device)  # noqa: E731
            self._buf[d] = {'x': [mk(self.n_pad), mk(self.n_pad)], 'acc_u': mk(self.bu), 'acc_i': mk(self.bi),
                            'out_u': mk(self.bu)}
        return self._buf[d]

    def _all_gather(self, full, local):
        """full [P*cb, d] <- the [cb, d] blocks of every rank; `local` is the view of `full` at this rank's offset.
        Asynchronous where the backend allows; returns a handle with wait()."""
        if not self.uses_collective or self._skip == 'gather':
            return _Done()
        if self._capi_comm is not None:
            return self._capi_comm.all_gather(full, local)
        if self.backend == 'nccl' or full.device.type == 'cpu':
            return dist.all_gather_into_tensor(full, local, group=self.group, async_op=True)
        # one-GPU rehearsal (gloo, device tensors): staged through host memory, synchronous
        host_local = local.detach().cpu().contiguous()
        host_full = torch.empty((self.world * host_local.shape[0], host_local.shape[1]), dtype=host_local.dtype)
        dist.all_gather_into_tensor(host_full, host_local, group=self.group)
        full.copy_(host_full.to(full.device))
        return _Done()

    def _launch(self, *a, **kw):
        if self._skip != 'spmm':
            self._spmm(*a, **kw)

    def phase_times(self, e0_u, e0_i, n_layers, reps=3, exact=False):
        """The two halves of forward() ALONE on this rank, same buffers, same launches: per layer k (1..K) the milliseconds of its
        SpMM launches with every all-gather left out (`spmm_alone_ms`) and the milliseconds its all-gathers take with every SpMM
        left out (`allgather_alone_ms`: the gathers layer k issues, waited for right behind them), HIP events on the launch
        stream, mean of `reps` passes.  The tables hold stale rows meanwhile (timing only); the next forward() rewrites them.
        max(sum of the two) / the measured forward = how much of the shorter half the pipeline hides."""
        out = {}
        keep = self.record_events
        self.record_events = True
        try:
            for what, key, field in (('gather', 'spmm_alone_ms', 'compute_ms'), ('spmm', 'allgather_alone_ms', 'wait_on_gather_ms')):
                self._skip = what
                tot = [0.0] * (n_layers + 2)
                for _ in range(reps):
                    self.forward(e0_u, e0_i, n_layers, exact=exact)
                    for rec in self.layer_times():
                        tot[rec['layer']] += rec[field]
                if what == 'gather':
                    out[key] = [round(tot[k] / reps, 3) for k in range(1, n_layers + 1)]
                else:   # the gathers layer k issues are waited for at the start of layer k + 1 (or at the end of the forward)
                    out[key] = [round(tot[k + 1] / reps, 3) for k in range(1, n_layers + 1)]
        finally:
            self._skip = None
            self.record_events = keep
        return out

    @staticmethod
    def _wait(works):
        for w in works:
            w.wait()

    def _mark(self, layer, kind):
        """record_events: a timed HIP event on the launch stream, tagged (layer, 'wait_begin' | 'wait_end' | 'compute_end')"""
        if self._events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self._events.append((layer, kind, ev))

    def layer_times(self):
        """After a forward() with record_events = True (and a device synchronisation): [{layer, compute_ms, wait_on_gather_ms}]
        per layer (+ layer K + 1 = the final wait for the last item gather), summed over the two half-steps of the layer."""
        if not self._events:
            return []
        torch.cuda.synchronize(self.device)
        out = {}
        prev = None
        for layer, kind, ev in self._events:
            rec = out.setdefault(layer, {'layer': layer, 'compute_ms': 0.0, 'wait_on_gather_ms': 0.0})
            if kind == 'wait_end':
                rec['wait_on_gather_ms'] += prev.elapsed_time(ev)
            elif kind == 'compute_end':
                rec['compute_ms'] += prev.elapsed_time(ev)
            prev = ev
        return [out[k] for k in sorted(out)]

    def _chunks(self, table_part, lay):
        """[(gather target slab, this rank's view of it)] per chunk of one node kind's part of a layer table"""
        return [(table_part[lay.chunk_slab(c)], table_part[lay.my_slab(self.rank, c)]) for c in range(lay.chunks)]

    # ------------------------------------------------------------------ forward
    def forward(self, e0_u, e0_i, n_layers, single=False, exact=False, copy=False):
        """e0_u [bu, d], e0_i [bi, d]: this rank's (padded) rows of E0.  The results are views of reusable buffers, dead after
        the next forward() (class docstring); copy=True returns private copies."""
        self._events = [] if (self.record_events and self.device.type == 'cuda') else None
        d = e0_u.shape[1]
        b = self.buffers(d)
        x, y = b['x']
        lu, li = self.lay_u, self.lay_i
        C = lu.chunks
        rows_u = [slice(c * lu.cb, (c + 1) * lu.cb) for c in range(C)]      # local rows of chunk c
        rows_i = [slice(c * li.cb, (c + 1) * li.cb) for c in range(C)]
        # layer-0 table: every rank's E0 chunks
        pending = []
        for part, lay, e0, rows in ((x[:self.u_pad], lu, e0_u, rows_u), (x[self.u_pad:], li, e0_i, rows_i)):
            for (full, mine), r in zip(self._chunks(part, lay), rows):
                mine.copy_(e0[r])
                pending.append(self._all_gather(full, mine))
        self._wait(pending)
        if n_layers == 0:
            b['out_u'].copy_(e0_u)
            return (b['out_u'].clone(), x[self.u_pad:].clone()) if copy else (b['out_u'], x[self.u_pad:])
        acc_u, acc_i = b['acc_u'], b['acc_i']
        # Pending gathers of the table being READ (x): users feed the item half-step, items feed the user half-step.
        wait_users, wait_items = [], []
        for k in range(1, n_layers + 1):
            last = k == n_layers
            div = float(n_layers + 1) if last else 1.0
            yu_chunks = self._chunks(y[:self.u_pad], lu)
            yi_chunks = self._chunks(y[self.u_pad:], li)

            def user_half():   # reads the item rows of x
                self._mark(k, 'wait_begin')
                self._wait(wait_items)
                self._mark(k, 'wait_end')
                works = []
                for c in range(C):
                    full, mine = yu_chunks[c]
                    r = rows_u[c]
                    if single:
                        self._launch(self.csr_u[c], x, y=b['out_u'][r] if last else mine, exact=exact)
                    else:
                        self._launch(self.csr_u[c], x, y=None if last else mine, acc_in=e0_u[r] if k == 1 else acc_u[r],
                                   acc_out=b['out_u'][r] if last else acc_u[r], acc_div=div, exact=exact)
                    if not last:   # users stay where they live after the last layer
                        works.append(self._all_gather(full, mine))
                self._mark(k, 'compute_end')
                return works

            def item_half():   # reads the user rows of x
                self._mark(k, 'wait_begin')
                self._wait(wait_users)
                self._mark(k, 'wait_end')
                works = []
                for c in range(C):
                    full, mine = yi_chunks[c]
                    r = rows_i[c]
                    if single:
                        self._launch(self.csr_i[c], x, y=mine, exact=exact)
                    else:
                        # on the last layer the item block of the *mean* is what gets gathered: write it into the table
                        self._launch(self.csr_i[c], x, y=None if last else mine, acc_in=e0_i[r] if k == 1 else acc_i[r],
                                   acc_out=mine if last else acc_i[r], acc_div=div, exact=exact)
                    works.append(self._all_gather(full, mine))
                self._mark(k, 'compute_end')
                return works

            # Alternate the order so that EVERY all-gather overlaps a half-step: the block gathered last in layer
            # k-1 is consumed last in layer k (odd layers: users then items; even layers: items then users).
            if k & 1:
                nu = user_half()
                ni = item_half()
            else:
                ni = item_half()
                nu = user_half()
            wait_users, wait_items = nu, ni
            x, y = y, x
        self._mark(n_layers + 1, 'wait_begin')
        self._wait(wait_users)
        self._wait(wait_items)
        self._mark(n_layers + 1, 'wait_end')
        # after the swap, x holds the last layer: its item part is the gathered item table (a view, no copy)
        if copy:
            return b['out_u'].clone(), x[self.u_pad:].clone()
        return b['out_u'], x[self.u_pad:]

    # ------------------------------------------------------------------ consumers
    def items_in_order(self, items_table):
        """[I_pad, d] table order -> [I, d] item-id order (one row gather; scoring reads this)."""
        if self._item_order is None:
            self._item_order = torch.from_numpy(self.lay_i.table_rows(np.arange(self.n_items))).to(items_table.device)
        return items_table.index_select(0, self._item_order)

    def gather_users(self, users_local):
        """All ranks' user blocks -> [U, d] in user-id order (tests / single-process consumers)."""
        d = users_local.shape[1]
        full = torch.empty((self.world * self.bu, d), dtype=torch.float32, device=users_local.device)
        mine = full[self.rank * self.bu:(self.rank + 1) * self.bu]
        mine.copy_(users_local)
        self._all_gather(full, mine).wait()
        if full.device.type == 'cuda':
            torch.cuda.current_stream(full.device).synchronize()
        rows = np.concatenate([r * self.bu + np.arange(self.lay_u.size(r)) for r in range(self.world)])
        return full[torch.from_numpy(rows).to(full.device)]

    def close(self):
        if self._capi_comm is not None:
            self._capi_comm.close()
            self._capi_comm = None


class ColumnShardedPropagator:
    """The alternative partition: by FEATURE columns.  Every rank holds the whole graph and d / P columns of every table.

    The product is independent per column, so a layer needs no exchange at all and each column's fmaf chain is exactly the
    single-GPU one (bit-identical, whatever P); one all-gather of the combined table after layer K assembles the d columns.
    Against the row partition this trades the (P-1)/P x N x d x 4 bytes received per layer for P-fold replicated CSR reads
    and narrower row gathers (4 d / P bytes: one 128-byte line at d = 64, P = 2; below that the fabric over-fetches).  On
    xGMI (point-to-point links) it is the better split for small P, where a row partition's all-gather crosses one or three
    links; it is an option (`bench.py --shard features`), the row partition stays the default."""

    def __init__(self, graph: NormGraph, d, rank, world, device, group=None, split_threshold=DEFAULT_SPLIT_THRESHOLD,
                 force_collective=False):
        from .propagate import Propagator
        self.rank, self.world = int(rank), int(world)
        self.device = resolve_device(device)
        self.group = group
        if d % self.world or (d // self.world) % 4 or (d // self.world) not in (8, 16, 32, 64, 128, 256):
            raise ValueError(f'd = {d} cannot be split in {world} column blocks of a supported width (8, 16, 32, 64, 128, 256)')
        self.d, self.dl = int(d), d // self.world
        self.cols = slice(self.rank * self.dl, (self.rank + 1) * self.dl)
        self.n = graph.n
        self.n_users = graph.n_users
        # narrow tables are far beyond what an XCD-affine block could hold: the one-wave-per-row / group kernels
        self.prop = Propagator(graph, self.device, split_threshold=split_threshold, segment=None)
        self.uses_collective = self.world > 1 or force_collective     # the process group is only needed by assemble()

    def local_e0(self, e0_full):
        """this rank's columns of a full [N, d] table (host or device tensor), contiguous on the device"""
        return e0_full[:, self.cols].to(self.device).contiguous()

    def forward(self, e0_cols, n_layers, single=False, exact=False):
        """e0_cols [N, d / P] -> the combined table's same columns [N, d / P]; no communication."""
        return self.prop.forward(e0_cols, n_layers, single=single, exact=exact)

    def assemble(self, out_cols):
        """[N, d / P] of every rank -> [N, d] on every rank: ONE all-gather + a column interleave."""
        if not self.uses_collective:
            return out_cols if self.world == 1 else None
        blocks = torch.empty((self.world, self.n, self.dl), dtype=torch.float32, device=out_cols.device)
        if dist.get_backend(self.group) == 'nccl' or out_cols.device.type == 'cpu':
            dist.all_gather_into_tensor(blocks.view(self.world * self.n, self.dl), out_cols.contiguous(), group=self.group)
        else:   # one-GPU rehearsal (gloo, device tensors): staged through host memory
            host = torch.empty((self.world * self.n, self.dl), dtype=torch.float32)
            dist.all_gather_into_tensor(host, out_cols.detach().cpu().contiguous(), group=self.group)
            blocks.copy_(host.view(self.world, self.n, self.dl))
        return blocks.permute(1, 0, 2).reshape(self.n, self.d)

    def close(self):
        pass


class UserShardedPropagator:
    """A third partition (round 4; an OPTION beside the prescribed row partition): by USERS only, the item table replicated.

    Rank p owns a contiguous nnz-balanced block of users U_p -- their rows of A and the [|U_p|, d] user tables, which never leave the
    rank -- and holds the whole [I, d] item table.  A is bipartite, so a layer is
        Y_u[U_p]  = A[U_p, items] . X_i                    local: the single-GPU chain of every user row, on the local item table
        P_i       = A[items, U_p] . X_u[U_p]               the item rows restricted to the rank's own users: a PARTIAL sum
        Y_i       = sum over ranks of P_i                  ONE all-reduce of [I, d] per layer
    Against the row partition's per-layer all-gather of BOTH tables ((P-1)/P x (U + I) x d x 4 bytes received per rank: 1.57 GB at
    config 4, P = 8) the exchange is an all-reduce of the ITEM table alone (2 (P-1)/P x I x d x 4 = 0.9 GB sent and received), and
    the 1.28 GB user table -- 71 % of the rows -- is never communicated; the partial item product gathers from the rank's own user
    block (160 MB at P = 8: Infinity-Cache resident).  The all-reduce of chunk c of the item rows runs under the partial product of
    chunks c + 1.. and under the whole user half-step.
    Price: an item row is the sum of P partial chains instead of one chain, added in the collective's order: the result equals the
    one-GPU forward to rounding (normwise ~1e-7; the path's bar is 1e-4), not bit for bit, and is only as deterministic as the
    collective (RCCL's ring / tree order is fixed for a fixed world).  The reference is single-device: nothing to match."""

    def __init__(self, graph: NormGraph, rank, world, device, group=None, split_threshold=DEFAULT_SPLIT_THRESHOLD, local_spmm=None,
                 chunks=1, force_collective=False):
        self.rank, self.world = int(rank), int(world)
        self.device = resolve_device(device)
        self.group = group
        self.n_users, self.n_items = graph.n_users, graph.n_items
        ub, _ = graph.partition(self.world)
        self.bounds = np.asarray(ub, dtype=np.int64)
        u0, u1 = int(self.bounds[self.rank]), int(self.bounds[self.rank + 1])
        self.u0, self.u1 = u0, u1
        self.bu = max(u1 - u0, 1)           # (a rank without users keeps one empty row: the kernels want a non-empty table)
        self._spmm = local_spmm or _hip_spmm
        self.uses_collective = self.world > 1 or force_collective
        self.backend = dist.get_backend(group) if self.uses_collective else 'none'
        I = self.n_items
        # user rows: columns are item ids (global id - U): rows of the local item table
        rp, ci, va = graph.row_block(u0, u1)
        rp = np.asarray(rp, dtype=np.int64)
        if u1 == u0:
            rp = np.zeros(2, dtype=np.int64)
        self.csr_u = DeviceCSR(rp, np.asarray(ci, dtype=np.int64) - graph.n_users, va, max(I, 1), self.device, split_threshold)
        # item rows restricted to the rank's users, in C row chunks; columns = local user index
        a, b = int(graph.rowptr[graph.n_users]), int(graph.rowptr[graph.n])
        cols = np.asarray(graph.colidx[a:b])
        keep = (cols >= u0) & (cols < u1)
        rows = np.repeat(np.arange(I, dtype=np.int64), np.diff(np.asarray(graph.rowptr[graph.n_users:graph.n + 1], dtype=np.int64)))[keep]
        lcols = (cols[keep].astype(np.int64) - u0)
        lvals = np.asarray(graph.vals[a:b])[keep]
        cnt = np.bincount(rows, minlength=I)
        rp_i = np.zeros(I + 1, dtype=np.int64)
        np.cumsum(cnt, out=rp_i[1:])
        self.chunks = max(1, min(int(chunks), max(I, 1)))
        self.cb = -(-max(I, 1) // self.chunks)
        self.csr_i = []
        for c in range(self.chunks):
            r0, r1 = min(c * self.cb, I), min((c + 1) * self.cb, I)
            e0, e1 = int(rp_i[r0]), int(rp_i[r1])
            self.csr_i.append(DeviceCSR(rp_i[r0:r1 + 1] - e0, lcols[e0:e1], lvals[e0:e1], self.bu, self.device, split_threshold)
                              if r1 > r0 else None)
        self.nnz_local = int(rp[-1]) + int(rp_i[-1])
        self._buf = {}
        self.record_events = False
        self._events = None

    def user_range(self):
        return self.u0, self.u1

    def local_e0(self, e0_full):
        """(this rank's user rows [bu, d], the whole item table [I, d]) of a full [N, d] table (host or device tensor)"""
        d = e0_full.shape[1]
        eu = torch.zeros((self.bu, d), dtype=torch.float32, device=self.device)
        eu[:self.u1 - self.u0] = e0_full[self.u0:self.u1].to(self.device)
        return eu, e0_full[self.n_users:].to(self.device).contiguous()

    def buffers(self, d):
        if d not in self._buf:
            mk = lambda n: torch.zeros((max(n, 1), d), dtype=torch.float32, device=self.device)  # noqa: E731
            self._buf[d] = {'xu': [mk(self.bu), mk(self.bu)], 'xi': [mk(self.n_items), mk(self.n_items)], 'acc_u': mk(self.bu),
                            'acc_i': mk(self.n_items), 'out_u': mk(self.bu), 'out_i': mk(self.n_items)}
        return self._buf[d]

    def _all_reduce(self, t):
        if not self.uses_collective:
            return _Done()
        if self.backend == 'nccl' or t.device.type == 'cpu':
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        host = t.detach().cpu()                  # one-GPU rehearsal (gloo, device tensors): staged through host memory
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(host.to(t.device))
        return _Done()

    def _mark(self, layer, kind):
        if self._events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self._events.append((layer, kind, ev))

    def layer_times(self):
        """After a forward() with record_events = True: [{layer, compute_ms, wait_on_reduce_ms}] -- ms the launch stream spent in the
        layer's SpMM launches and its item epilogue, and ms it sat waiting for the all-reduced item rows."""
        if not self._events:
            return []
        torch.cuda.synchronize(self.device)
        out = {}
        prev = None
        for layer, kind, ev in self._events:
            rec = out.setdefault(layer, {'layer': layer, 'compute_ms': 0.0, 'wait_on_reduce_ms': 0.0})
            if prev is not None and kind in ('launched', 'done'):
                rec['compute_ms'] += prev.elapsed_time(ev)
            elif prev is not None and kind == 'reduced':
                rec['wait_on_reduce_ms'] += prev.elapsed_time(ev)
            prev = ev
        return [out[k] for k in sorted(out)]

    def forward(self, e0_u, e0_i, n_layers, single=False, exact=False):
        """e0_u [bu, d]: this rank's users; e0_i [I, d]: all items.  -> (users_local [bu, d], items [I, d] in item-id order): views
        of reusable buffers, valid until the next forward() on this object."""
        self._events = [] if (self.record_events and self.device.type == 'cuda') else None
        d = e0_u.shape[1]
        b = self.buffers(d)
        if n_layers == 0:
            b['out_u'].copy_(e0_u)
            b['out_i'].copy_(e0_i)
            return b['out_u'], b['out_i']
        xu, xi = e0_u, e0_i
        I = self.n_items
        for k in range(1, n_layers + 1):
            last = k == n_layers
            div = float(n_layers + 1) if last else 1.0
            yu, yi = b['xu'][k & 1], b['xi'][k & 1]
            self._mark(k, 'start')
            # item partials first: their all-reduce runs under the remaining chunks and under the user half-step
            works = []
            for c, csr in enumerate(self.csr_i):
                if csr is None:
                    continue
                r = slice(c * self.cb, min((c + 1) * self.cb, I))
                self._spmm(csr, xu, y=yi[r], exact=exact)
                works.append(self._all_reduce(yi[r]))
            # user half-step: the single-GPU chains, layer sum fused
            if single:
                self._spmm(self.csr_u, xi, y=b['out_u'] if last else yu, exact=exact)
            else:
                self._spmm(self.csr_u, xi, y=None if last else yu, acc_in=e0_u if k == 1 else b['acc_u'],
                           acc_out=b['out_u'] if last else b['acc_u'], acc_div=div, exact=exact)
            self._mark(k, 'launched')
            for w in works:
                w.wait()
            self._mark(k, 'reduced')
            # the items' layer sum, on the reduced rows (the epilogue's arithmetic: add, then one IEEE division on the last layer)
            if single:
                if last:
                    b['out_i'].copy_(yi)
            else:
                torch.add(e0_i if k == 1 else b['acc_i'], yi, out=b['out_i'] if last else b['acc_i'])
                if last:
                    b['out_i'].div_(div)
            self._mark(k, 'done')
            xu, xi = yu, yi
        return b['out_u'], b['out_i']

    def gather_users(self, users_local):
        """All ranks' user blocks -> [U, d] in user-id order (tests / single-process consumers)."""
        d = users_local.shape[1]
        bmax = int(np.diff(self.bounds).max()) if self.world > 0 else self.bu
        bmax = max(bmax, 1)
        mine = torch.zeros((bmax, d), dtype=torch.float32, device=users_local.device)
        mine[:self.u1 - self.u0] = users_local[:self.u1 - self.u0]
        full = torch.empty((self.world * bmax, d), dtype=torch.float32, device=users_local.device)
        if not self.uses_collective:
            full.copy_(mine)
        elif self.backend == 'nccl' or full.device.type == 'cpu':
            dist.all_gather_into_tensor(full, mine, group=self.group)
        else:
            host = torch.empty((self.world * bmax, d), dtype=torch.float32)
            dist.all_gather_into_tensor(host, mine.cpu(), group=self.group)
            full.copy_(host.to(full.device))
        rows = np.concatenate([r * bmax + np.arange(int(self.bounds[r + 1] - self.bounds[r])) for r in range(self.world)])
        return full[torch.from_numpy(rows).to(full.device)]

    def close(self):
        pass
