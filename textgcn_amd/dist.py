"""Row-sharded K-layer propagation over P ranks (one process per GPU) -- SURVEY.md §8(e).

The reference is single-device (no collective anywhere); this is new.  Y[r,:] depends on row r
of A and the whole previous layer, so rows are partitioned 1-D and each layer ends with an all-gather of the
freshly propagated blocks.  A is bipartite ([[0, R],[R^T, 0]]): user rows read only item rows of X and vice
versa, so a layer is two half-steps and the all-gather of the user block runs on RCCL's stream while the item
half-step computes.

Layout.  Users and items are each cut into P equal blocks of ceil(U/P) resp. ceil(I/P) rows (tables padded
with zero rows that no column index references), so every block of every rank has the same size and
`all_gather_into_tensor` writes straight into the layer buffer -- no packing, no uneven collective.  Padded
row ids: user u -> u, item i -> U_pad + i.  Row ownership never changes a row's summation order, so the
P-rank result is bit-identical to the 1-rank result.

Backends.  "nccl" (= RCCL over xGMI) gathers device buffers directly.  Any other backend (gloo in the CPU /
single-GPU rehearsals) is staged through host memory by this module; it exists for tests only.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._capi import resolve_device
from .graph import NormGraph
from .propagate import DEFAULT_SPLIT_THRESHOLD, DeviceCSR, spmm


def padded_layout(n_users, n_items, world):
    bu = -(-n_users // world)
    bi = -(-n_items // world)
    return bu, bi, bu * world, bi * world


def _hip_spmm(csr, x, **kw):
    return spmm(csr, x, **kw)


class ShardedPropagator:
    """One rank's share of the K-layer forward.

    forward(e0_users_local, e0_items_local) -> (users_local [bu, d], items_full [I_pad, d]): the rank keeps
    its own users (they are scored where they live) and the gathered item table of the layer mean.
    """

    def __init__(self, graph: NormGraph, rank, world, device, group=None, split_threshold=DEFAULT_SPLIT_THRESHOLD,
                 local_spmm=None):
        self.rank, self.world = int(rank), int(world)
        self.device = resolve_device(device)
        self.group = group
        self.n_users, self.n_items = graph.n_users, graph.n_items
        self.bu, self.bi, self.u_pad, self.i_pad = padded_layout(graph.n_users, graph.n_items, self.world)
        self.n_pad = self.u_pad + self.i_pad
        self._spmm = local_spmm or _hip_spmm   # tests inject a CPU stand-in to rehearse the exchange logic
        self.backend = dist.get_backend(group) if self.world > 1 else 'none'
        u0, u1 = self._user_range(self.rank)
        i0, i1 = self._item_range(self.rank)
        # local CSR blocks, column ids remapped to the padded layout
        rp, ci, va = graph.row_block(u0, u1)
        self.csr_u = self._make_csr(rp, ci.astype(np.int64) + (self.u_pad - graph.n_users), va, self.bu, split_threshold)
        rp, ci, va = graph.row_block(graph.n_users + i0, graph.n_users + i1)
        self.csr_i = self._make_csr(rp, ci, va, self.bi, split_threshold)
        self.nnz_local = self.csr_u.nnz + self.csr_i.nnz
        self._buf = {}

    def _user_range(self, r):
        return min(r * self.bu, self.n_users), min((r + 1) * self.bu, self.n_users)

    def _item_range(self, r):
        return min(r * self.bi, self.n_items), min((r + 1) * self.bi, self.n_items)

    def _make_csr(self, rowptr, colidx, vals, n_rows_padded, split_threshold):
        rowptr = np.asarray(rowptr, dtype=np.int64)
        if len(rowptr) - 1 < n_rows_padded:   # pad rows: empty
            rowptr = np.concatenate([rowptr, np.full(n_rows_padded - (len(rowptr) - 1), rowptr[-1], dtype=np.int64)])
        return DeviceCSR(rowptr, colidx, vals, self.n_pad, self.device, split_threshold)

    # ------------------------------------------------------------------ helpers
    def local_e0(self, e0_full):
        """Cut this rank's padded E0 blocks out of a full [N, d] table (host or device tensor)."""
        d = e0_full.shape[1]
        u0, u1 = self._user_range(self.rank)
        i0, i1 = self._item_range(self.rank)
        eu = torch.zeros((self.bu, d), dtype=torch.float32, device=self.device)
        ei = torch.zeros((self.bi, d), dtype=torch.float32, device=self.device)
        eu[:u1 - u0] = e0_full[u0:u1].to(self.device)
        ei[:i1 - i0] = e0_full[self.n_users + i0:self.n_users + i1].to(self.device)
        return eu, ei

    def buffers(self, d):
        if d not in self._buf:
            mk = lambda n: torch.zeros((n, d), dtype=torch.float32, device=self.device)  # noqa: E731
            self._buf[d] = {'x': [mk(self.n_pad), mk(self.n_pad)], 'acc_u': mk(self.bu), 'acc_i': mk(self.bi),
                            'out_u': mk(self.bu), 'out_i': mk(self.i_pad)}
        return self._buf[d]

    def _all_gather(self, full, local, async_op):
        """full [P*b, d] <- blocks of every rank (local is a view of full at this rank's offset, or a
        separate tensor).  Returns a work handle or None."""
        if self.world == 1:
            if local.data_ptr() != full[self.rank * local.shape[0]:].data_ptr():
                full[self.rank * local.shape[0]:(self.rank + 1) * local.shape[0]].copy_(local)
            return None
        if self.backend == 'nccl':
            return dist.all_gather_into_tensor(full, local, group=self.group, async_op=async_op)
        # rehearsal path (gloo): stage through host memory, synchronous
        host_local = local.detach().cpu().contiguous()
        host_full = torch.empty((self.world * host_local.shape[0], host_local.shape[1]), dtype=host_local.dtype)
        dist.all_gather_into_tensor(host_full, host_local, group=self.group)
        full.copy_(host_full.to(full.device))
        return None

    @staticmethod
    def _wait(work):
        if work is not None:
            work.wait()

    # ------------------------------------------------------------------ forward
    def forward(self, e0_u, e0_i, n_layers, single=False, exact=False):
        """e0_u [bu, d], e0_i [bi, d]: this rank's (padded) rows of E0."""
        d = e0_u.shape[1]
        b = self.buffers(d)
        x, y = b['x']
        xu, xi = x[:self.u_pad], x[self.u_pad:]
        my_u = slice(self.rank * self.bu, (self.rank + 1) * self.bu)
        my_i = slice(self.rank * self.bi, (self.rank + 1) * self.bi)
        # layer-0 table: gather E0 blocks
        xu[my_u].copy_(e0_u)
        xi[my_i].copy_(e0_i)
        self._wait(self._all_gather(xu, xu[my_u], False))
        self._wait(self._all_gather(xi, xi[my_i], False))
        if n_layers == 0:
            b['out_u'].copy_(e0_u)
            b['out_i'].copy_(xi)
            return b['out_u'], b['out_i']
        acc_u, acc_i = b['acc_u'], b['acc_i']
        # Pending gathers of the table being READ (x): users feed the item half-step, items feed the user half-step.
        wait_users = wait_items = None
        for k in range(1, n_layers + 1):
            last = k == n_layers
            yu, yi = y[:self.u_pad], y[self.u_pad:]
            div = float(n_layers + 1) if last else 1.0

            def user_half():   # reads the item rows of x
                self._wait(wait_items)
                if single:
                    self._spmm(self.csr_u, x, y=b['out_u'] if last else yu[my_u], exact=exact)
                else:
                    self._spmm(self.csr_u, x, y=None if last else yu[my_u], acc_in=e0_u if k == 1 else acc_u,
                               acc_out=b['out_u'] if last else acc_u, acc_div=div, exact=exact)
                return None if last else self._all_gather(yu, yu[my_u], True)

            def item_half():   # reads the user rows of x
                self._wait(wait_users)
                if single:
                    self._spmm(self.csr_i, x, y=yi[my_i], exact=exact)
                else:
                    # on the last layer the item block of the *mean* is what gets gathered: write it into yi
                    self._spmm(self.csr_i, x, y=None if last else yi[my_i], acc_in=e0_i if k == 1 else acc_i,
                               acc_out=yi[my_i] if last else acc_i, acc_div=div, exact=exact)
                return self._all_gather(yi, yi[my_i], True)

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
        self._wait(wait_users)
        self._wait(wait_items)
        # after the swap, x holds the last layer: its item part is the gathered item table
        b['out_i'].copy_(x[self.u_pad:])
        return b['out_u'], b['out_i']

    def gather_users(self, users_local):
        """All ranks' user blocks -> [U_pad, d] (for tests / single-process consumers)."""
        full = torch.empty((self.u_pad, users_local.shape[1]), dtype=torch.float32, device=self.device)
        self._wait(self._all_gather(full, users_local.contiguous(), False))
        return full
