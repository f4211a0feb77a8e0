"""LightGCN with the reference's BaseModel method surface, computed by the HIP path.

Drop-in for TextGCN/base_model.py (BaseModel): same constructor `(params, dataset)`, same hook methods
(`_copy_params`, `_copy_dataset_params`, `_init_embeddings`, `_add_vars`), same public members
(`representation`, `layer_aggregation`, `layer_combination`, `score_pairwise`, `score_batchwise`, `predict`,
`evaluate`, `fit`, `get_loss`, `bpr_loss`, `reg_loss`, `load_model`, `checkpoint`, `embedding_user`,
`embedding_item`, `k`, `batch_size`, `n_layers`, `emb_size`, `training`) and the same state_dict keys
(`embedding_user.weight`, `embedding_item.weight`), so `main.py`'s lgcn flow runs unchanged with this class in
its registry.

What changed underneath: the K x torch.sparse.mm + cat + stack/mean of `representation`
is tgcn_spmm_csr_f32 with the layer sum fused (K1-K4); matmul / mask / topk / round of `predict` are
tgcn_score_* / tgcn_mask / tgcn_topk (K5-K8) fed from a device CSR of the train items instead of a pandas
explode per batch; training differentiates through the same kernel (backward = transposed product) and edge
dropout keeps the CSR structure, zeroing values (K10).
"""
import logging
import os
import shutil
import weakref
from collections import defaultdict

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _capi, scoring
from ._capi import resolve_device
from .graph import NormGraph, train_mask_csr
from .metrics import METRICS, early_stop, ranking_metrics_device, true_lists_csr
from .propagate import DEFAULT_SPLIT_THRESHOLD, DeviceCSR, EdgeValues, Propagator, spmm


class _Propagate(torch.autograd.Function):
    """out = combine(E0, A E0, ..., A^K E0) with A given by (structure, values); backward is the same kernel on
    the transposed values: dE0 = sum_k (A^T)^k G / (K+1)  (Horner: g <- A^T g + G'), or (A^T)^K G for --single."""

    @staticmethod
    def forward(ctx, e0, model, vals, vals_t):
        ctx.model, ctx.vals_t = model, vals_t
        eng = model._engine
        out = torch.empty_like(e0)
        eng.forward(e0.detach(), model.n_layers, single=model._single, exact=model.exact, out=out, vals=vals)
        return out

    @staticmethod
    def backward(ctx, grad):
        model = ctx.model
        grad = grad.contiguous()
        if model.n_layers == 0:
            return grad, None, None, None
        g0 = grad if model._single else grad / float(model.n_layers + 1)
        return _backward_propagate(model, g0, ctx.vals_t), None, None, None


def _backward_propagate(model, grad, vals_t):
    """dE0 for d(out) = grad: sum_k (A^T)^k grad / (K + 1) in Horner form (g <- A^T g + g0), or (A^T)^K grad for --single.
    `grad` must already carry the 1 / (K + 1) of the layer mean when scaled=True."""
    eng = model._engine
    K = model.n_layers
    if K == 0:
        return grad
    ping, pong, _ = eng.buffers(grad.shape[1])
    x = grad
    for k in range(K):
        y = torch.empty_like(grad) if k == K - 1 else (ping if k % 2 == 0 else pong)
        if model._single:
            spmm(eng.csr, x, y=y, exact=model.exact, vals=vals_t)
        else:
            spmm(eng.csr, x, y=None, acc_in=grad, acc_out=y, exact=model.exact, vals=vals_t)
        x = y
    return x


class _MatrixHandle:
    """The model's own matrix as `representation` hands it to `layer_aggregation` when only `layer_combination` is overridden:
    the engine's CSR + the call's values (edge dropout) and their transpose.  Not a tensor: an override of
    `layer_aggregation` gets a torch sparse COO tensor instead, as in the reference."""

    def __init__(self, csr, vals=None, vals_t=None):
        self.csr, self.vals, self.vals_t = csr, vals, vals_t


class _Spmm(torch.autograd.Function):
    """y = A . emb for one `layer_aggregation` call (base_model.py:148) with autograd: backward = A^T . g -- the same kernel
    on the transposed values (the model's matrix: same structure) or on the transposed CSR (a matrix the caller brought)."""

    @staticmethod
    def forward(ctx, emb, model, csr, vals, transposed):
        ctx.model, ctx.transposed = model, transposed
        y = torch.empty((csr.n_rows, emb.shape[1]), dtype=torch.float32, device=emb.device)
        spmm(csr, emb.detach().contiguous(), y=y, exact=model.exact, vals=vals)
        return y

    @staticmethod
    def backward(ctx, grad):
        csr_t, vals_t = ctx.transposed()
        g = torch.empty((csr_t.n_rows, grad.shape[1]), dtype=torch.float32, device=grad.device)
        spmm(csr_t, grad.contiguous(), y=g, exact=ctx.model.exact, vals=vals_t)
        return g, None, None, None, None


class _BprStep(torch.autograd.Function):
    """get_loss (base_model.py:181-210) as ONE autograd node on the native path: K-layer propagation (tgcn_spmm_*), BPR
    pair terms (tgcn_bpr_pairs_f32), L2 terms (tgcn_reg_rows_f32); backward = gradient scatter of the pairs into a zeroed
    table, the transposed propagation of that table, and the L2 rows.  No torch gather / elementwise op touches a [*, d]
    tensor, and autograd's upstream factor reaches the kernels as a device scalar (no host round trip)."""

    @staticmethod
    def forward(ctx, wu, wi, model, cols, n_real=None):
        """cols [2 + m, b] int64 (users, positives, negatives...).  n_real: None, or a DEVICE scalar = the number of rows whose
        user id is not negative -- rows with a negative user id are padding (the kernels skip them) and the means run over the
        n_real real rows (AdvSamplModel keeps its ragged triple lists dense on the device this way)."""
        lib = _capi.lib()
        dev = model.device
        stream = _capi.current_stream(dev)
        e0 = model._e0_flat(wu, wi)
        n_u, d, K = model.n_users, e0.shape[1], model.n_layers
        drop = model._dropout_values() if (model.training and model.dropout > 0) else (None, None)
        out = torch.empty_like(e0)
        model._engine.forward(e0, K, single=model._single, exact=model.exact, out=out, vals=drop[0])
        b, m = cols.shape[1], cols.shape[0] - 2
        users, pos, negs = cols[0], cols[1], cols[2:]
        terms = torch.empty((m, b), dtype=torch.float32, device=dev)
        _capi.check(lib.tgcn_bpr_pairs_f32(_capi.ptr(out[:n_u]), _capi.ptr(out[n_u:]), _capi.ptr(users), _capi.ptr(pos), _capi.ptr(negs),
                                           b, m, d, 1.0, None, _capi.ptr(terms), None, None, stream), 'tgcn_bpr_pairs_f32')
        reg_terms = torch.empty((b,), dtype=torch.float32, device=dev)
        _capi.check(lib.tgcn_reg_rows_f32(_capi.ptr(e0[:n_u]), _capi.ptr(e0[n_u:]), _capi.ptr(users), _capi.ptr(pos), _capi.ptr(negs),
                                          b, m, d, 0.0, None, _capi.ptr(reg_terms), None, None, stream), 'tgcn_reg_rows_f32')
        if n_real is None:
            bpr = terms.sum() / float(b * m)
            reg = reg_terms.sum() * (model.reg_lambda / (2.0 * b))
            ctx.rescale = None
        else:
            n_real = n_real.to(torch.float32).clamp(min=1.0)
            bpr = terms.sum() / (n_real * float(m))
            reg = reg_terms.sum() * (model.reg_lambda / 2.0) / n_real
            ctx.rescale = float(b) / n_real          # the kernels divide their gradients by b (x m): corrected through *upstream
        ctx.model, ctx.out, ctx.vals_t, ctx.cols, ctx.e0 = model, out, drop[1], cols, e0
        return bpr, reg

    @staticmethod
    def backward(ctx, g_bpr, g_reg):
        model, out, cols, e0 = ctx.model, ctx.out, ctx.cols, ctx.e0
        lib = _capi.lib()
        stream = _capi.current_stream(model.device)
        n_u, d, K = model.n_users, e0.shape[1], model.n_layers
        b, m = cols.shape[1], cols.shape[0] - 2
        g_bpr = g_bpr.to(torch.float32).contiguous()
        g_reg = g_reg.to(torch.float32).contiguous()
        if ctx.rescale is not None:
            g_bpr, g_reg = (g_bpr * ctx.rescale).contiguous(), (g_reg * ctx.rescale).contiguous()
        grad = torch.zeros_like(e0)
        scale = 1.0 if (model._single or K == 0) else 1.0 / float(K + 1)     # the layer mean's factor, folded into the scatter
        _capi.check(lib.tgcn_bpr_pairs_f32(_capi.ptr(out[:n_u]), _capi.ptr(out[n_u:]), _capi.ptr(cols[0]), _capi.ptr(cols[1]),
                                           _capi.ptr(cols[2:]), b, m, d, scale, _capi.ptr(g_bpr), None, _capi.ptr(grad[:n_u]),
                                           _capi.ptr(grad[n_u:]), stream), 'tgcn_bpr_pairs_f32')
        de0 = _backward_propagate(model, grad, ctx.vals_t)
        _capi.check(lib.tgcn_reg_rows_f32(_capi.ptr(e0[:n_u]), _capi.ptr(e0[n_u:]), _capi.ptr(cols[0]), _capi.ptr(cols[1]),
                                          _capi.ptr(cols[2:]), b, m, d, model.reg_lambda / b, _capi.ptr(g_reg), None,
                                          _capi.ptr(de0[:n_u]), _capi.ptr(de0[n_u:]), stream), 'tgcn_reg_rows_f32')
        ctx.out = None
        return de0[:n_u], de0[n_u:], None, None, None


class LightGCN(nn.Module):
    """reference: TextGCN/base_model.py:17-299 (class BaseModel)."""

    predict_chunk = 16384   # users per fused scoring call
    predict_streams = 4     # chunks in flight (one HIP stream + scratch buffer each)
    score_prefilter = True  # predict: candidates from the bf16 matrix pass, scores from the fp32 chains (identical lists and
                            # scores, ~2x the throughput for d <= 128); False: the fp32 MFMA filter finds the candidates
    exact = False   # True: no long-row split -> every row is one fmaf chain (bit-identical to the CPU reference)
    fused_adam = True
    deferred_nan_check = False   # False: fit() reads the loss's NaN flag before backward(), as base_model.py:123 does; True: one
                                 # step later (no pipeline bubble; the weights have then taken two steps with the NaN)

    def __init__(self, params, dataset):
        super().__init__()
        self._copy_params(params)
        self._copy_dataset_params(dataset)
        self._init_embeddings(params.emb_size)
        self._add_vars(params)
        self.load_model(getattr(params, 'load', None))
        self.to(self.device)

    # ------------------------------------------------------------------ construction hooks (base_model.py:33-75)
    def _copy_params(self, params):
        g = lambda name, default=None: getattr(params, name, default)  # noqa: E731
        self.k = sorted(g('k', [20, 40]))
        self.lr = g('lr', 0.001)
        self.uid = g('uid', 'model')
        self.save = g('save', False)
        self.quiet = g('quiet', True)
        self.epochs = g('epochs', 1)
        self.logger = g('logger') or logging.getLogger('textgcn_amd')
        self.device = resolve_device(g('device', 'cuda' if torch.cuda.is_available() else 'cpu'))
        self.dropout = g('dropout', 0.4)
        self.emb_size = g('emb_size', 64)
        self.n_layers = g('n_layers', 3)
        self.save_path = g('save_path', '.')
        self.batch_size = g('batch_size', 2048)
        self.reg_lambda = g('reg_lambda', 1e-4)
        self.evaluate_every = g('evaluate_every', 25)
        self.neg_samples = g('neg_samples', 1)
        self.slurm = g('slurm', False) or self.quiet
        self._single = bool(g('single', False))
        if self._single:   # base_model.py:51-52
            self.layer_combination = self.layer_combination_single
        if g('exact') is not None:
            self.exact = bool(g('exact'))
        self._split_threshold = g('split_threshold', DEFAULT_SPLIT_THRESHOLD)
        # 'cpu': torch.rand(nnz) on the CPU generator, the reference's exact stream (base_model.py:82) -- 10 M draws
        # + a host-to-device copy per training step on config 2; 'device': same Bernoulli(1-p) law drawn on the GPU
        self.dropout_rng = g('dropout_rng', 'device')

    def _copy_dataset_params(self, dataset):
        self.n_users = dataset.n_users
        self.n_items = dataset.n_items
        # the reference's container (base_model.py:57): a torch sparse COO tensor.  A dataset of ours builds it lazily
        # (interactions.py), so it is only fetched when somebody asks for it (`norm_matrix` property below)
        self._norm_matrix = dataset.norm_matrix if not hasattr(dataset, 'graph') else None
        self._dataset = dataset
        self.true_test_lil = dataset.true_test_lil
        self.train_user_dict = dataset.train_user_dict
        self.test_users = np.sort(np.asarray(dataset.test_df.user_id.unique()))
        self.user_mapping_dict = dict(dataset.user_mapping[['remap_id', 'org_id']].values)
        self.item_mapping_dict = dict(dataset.item_mapping[['remap_id', 'org_id']].values)
        # hot-path inputs: CSR Laplacian + CSR of train items per user
        if hasattr(dataset, 'graph'):
            self.graph = dataset.graph
        else:   # a reference BaseDataset: adopt its coalesced COO (dataset.py:138)
            self.graph = NormGraph.from_coo(dataset.norm_matrix, None, self.n_users, self.n_items)
        if hasattr(dataset, 'mask_rowptr'):
            rp, items = dataset.mask_rowptr, dataset.mask_items
        else:
            tud = dataset.train_user_dict
            users = np.repeat(np.asarray(tud.index), [len(v) for v in tud.values])
            rp, items = train_mask_csr(users, np.concatenate([np.asarray(v) for v in tud.values]), self.n_users)
        self._mask_rowptr_host, self._mask_items_host = np.asarray(rp, dtype=np.int64), np.asarray(items, dtype=np.int32)
        self._true_csr_host = true_lists_csr(self.true_test_lil)   # relevant lists of the test users, in test_users order

    def _init_embeddings(self, emb_size):
        """base_model.py:64-69; the two tables are views of ONE contiguous [N, d] buffer so the layer-0 matrix
        needs no torch.cat (K2)."""
        n = self.n_users + self.n_items
        self._flat = torch.empty((n, emb_size), dtype=torch.float32, device=self.device)
        self.embedding_user = nn.Embedding(self.n_users, emb_size, _weight=self._flat[:self.n_users])
        self.embedding_item = nn.Embedding(self.n_items, emb_size, _weight=self._flat[self.n_users:])
        nn.init.normal_(self.embedding_user.weight, std=0.1)
        nn.init.normal_(self.embedding_item.weight, std=0.1)

    def _add_vars(self, params):
        self.metrics = list(METRICS)
        self.metrics_logger = {m: np.zeros((0, len(self.k))) for m in self.metrics}
        self.training = False
        self._engine_obj = None
        self._mask_dev = None
        self._true_dev = None
        self._drop = None

    # ------------------------------------------------------------------ engine (device state, built lazily)
    @property
    def _engine(self):
        if self._engine_obj is None:
            if self.device.type != 'cuda':
                raise RuntimeError('LightGCN computes on a ROCm GPU only; there is no CPU fallback '
                                   f'(model device is {self.device})')
            self._engine_obj = Propagator(self.graph, self.device, split_threshold=self._split_threshold)
        return self._engine_obj

    def _predict_streams(self):
        if getattr(self, '_streams', None) is None:
            self._streams = [torch.cuda.Stream(self.device) for _ in range(self.predict_streams)]
        return self._streams

    def _mask(self):
        if self._mask_dev is None:
            self._mask_dev = (torch.from_numpy(self._mask_rowptr_host).to(self.device),
                              torch.from_numpy(self._mask_items_host).to(self.device))
        return self._mask_dev

    def _e0_flat(self, wu, wi):
        """detached [N, d] view of the two tables when they still share one buffer (no copy), else their concatenation"""
        if (wu.is_contiguous() and wi.is_contiguous()
                and wi.data_ptr() == wu.data_ptr() + wu.numel() * 4 and wu.untyped_storage().data_ptr() == wi.untyped_storage().data_ptr()):
            return torch.as_strided(wu.detach(), (self.n_users + self.n_items, wu.shape[1]), (wu.shape[1], 1))
        return torch.cat([wu.detach(), wi.detach()])

    def _e0(self):
        """[N, d] layer-0 matrix (base_model.py:88-91) without a copy when the tables still share storage."""
        wu, wi = self.embedding_user.weight, self.embedding_item.weight
        if torch.is_grad_enabled() and (wu.requires_grad or wi.requires_grad):
            return torch.cat([wu, wi])   # autograd needs the graph edge; forward cost is one copy (get_loss avoids it)
        return self._e0_flat(wu, wi)

    @property
    def embedding_matrix(self):
        return self._e0()

    # ------------------------------------------------------------------ dropout (base_model.py:77-86)
    def _dropout_values(self):
        """Edge dropout as value masking on the fixed CSR (tgcn_dropout_values_f32): keep entry e iff u_e < 1 - p, kept
        values scaled by 1/(1-p).  With dropout_rng='cpu' u is torch.rand(nnz) on the CPU generator exactly as
        base_model.py:82, so a seeded run drops the same edges as the reference (tests); the default is a Philox draw on
        the device keyed by a seed taken from torch's generator.  ONE launch writes the values, the transposed values
        (the backward's matrix) and both copies in the segment plan's stream order.
        Returns (EdgeValues forward, EdgeValues backward)."""
        g = self.graph
        dev = self.device
        csr = self._engine.csr
        if self._drop is None:
            perm = g.transpose_perm()
            sym = bool(np.array_equal(g.vals[perm], g.vals))    # (d_r a) d_c == (d_c a) d_r bit for bit unless edges repeat
            self._drop = {'perm': torch.from_numpy(perm.astype(np.int32)).to(dev),
                          'stored_t': None if sym else torch.from_numpy(g.vals[perm]).to(dev), 'plans': {}}
        st = self._drop
        nnz = max(g.nnz, 1)
        rand_u, seed = None, 0
        if self.dropout_rng == 'cpu':
            rand_u = torch.rand(g.nnz).to(dev)
        else:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        d = self.emb_size
        ent_src = csr.segment_ent_src(d) if not self.exact else None
        n_stream = 0 if ent_src is None else ent_src.numel()
        ent_stored = ent_src_t = ent_stored_t = None
        if n_stream:
            if d not in st['plans']:    # static per plan: the transposed entry of every stream slot (+ its stored value)
                src_t = st['perm'].index_select(0, ent_src.to(torch.int64))
                st['plans'][d] = (src_t, None if st['stored_t'] is None else st['stored_t'].index_select(0, ent_src.to(torch.int64)))
            ent_src_t, ent_stored_t = st['plans'][d]
            ent_stored = csr._segment_plans[d][0][2]['ent_val']
        vals, vals_t = torch.empty(nnz, dtype=torch.float32, device=dev), torch.empty(nnz, dtype=torch.float32, device=dev)
        ev = ev_t = None
        if n_stream:
            ev, ev_t = torch.empty(n_stream, dtype=torch.float32, device=dev), torch.empty(n_stream, dtype=torch.float32, device=dev)
        rc = _capi.lib().tgcn_dropout_values_f32(_capi.ptr(csr.vals), _capi.ptr(st['stored_t']), _capi.ptr(rand_u), seed,
                                                 float(1 - self.dropout), _capi.ptr(st['perm']), _capi.ptr(ent_stored),
                                                 _capi.ptr(ent_stored_t), _capi.ptr(ent_src), _capi.ptr(ent_src_t), g.nnz, n_stream,
                                                 _capi.ptr(vals), _capi.ptr(vals_t), _capi.ptr(ev), _capi.ptr(ev_t),
                                                 _capi.current_stream(dev))
        _capi.check(rc, 'tgcn_dropout_values_f32')
        return EdgeValues(vals, ev), EdgeValues(vals_t, ev_t)

    # ------------------------------------------------------------------ forward (base_model.py:93-106)
    @property
    def norm_matrix(self):
        """base_model.py:57: the normalised Laplacian as a torch sparse COO tensor (the reference's container).  The HIP path
        reads its own CSR of the same matrix; this tensor exists for callers and overrides that want the reference's object.
        Built on first use from the dataset's tensor or from the graph, on the model's device; `layer_aggregation` recognises
        it and uses the engine's CSR without converting anything."""
        if self._norm_matrix is None:
            ds_nm = getattr(self._dataset, 'norm_matrix', None)
            if ds_nm is not None:
                self._norm_matrix = ds_nm
            else:
                idx, val = self.graph.to_coo()
                self._norm_matrix = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(np.array(val)),
                                                            (self.graph.n, self.graph.n), is_coalesced=True)
        return self._norm_matrix

    @norm_matrix.setter
    def norm_matrix(self, value):
        self._norm_matrix = value

    def _overridden(self, name):
        """True when `name` is not LightGCN's own member any more: replaced on the instance (the reference rebinds members that
        way, ltr_models.py:177-179) or overridden by a subclass (rejected_models.py:27-39).  `--single`'s own rebinding of
        layer_combination (base_model.py:51-52) is not an override."""
        own = self.__dict__.get(name)
        if own is not None:
            if name == 'layer_combination' and getattr(own, '__func__', None) is LightGCN.layer_combination_single \
                    and getattr(own, '__self__', None) is self:
                return type(self).layer_combination_single is not LightGCN.layer_combination_single
            return True
        return getattr(type(self), name) is not getattr(LightGCN, name)

    @property
    def representation(self):
        """base_model.py:93-106.  With LightGCN's own layer_aggregation and layer_combination the K products and the layer
        mean run fused (one engine call, one autograd node).  When either member is overridden -- on a subclass or on the
        instance -- the reference's loop runs THROUGH the overrides: layer_aggregation(norm_matrix, E^k) K times, then
        layer_combination([E^0..E^K]); an overridden layer_aggregation receives a torch sparse COO tensor (the dropped matrix
        in training), as the reference's would."""
        e0 = self._e0()
        if self.training and self.dropout > 0:
            vals, vals_t = self._dropout_values()
        else:
            vals = vals_t = None   # A is symmetric: A^T = A
        agg_custom, comb_custom = self._overridden('layer_aggregation'), self._overridden('layer_combination')
        if agg_custom or comb_custom:
            if agg_custom:
                matrix = self.norm_matrix if vals is None else self._dropped_coo(vals, vals_t)
                if matrix.device != self.device:
                    matrix = self._norm_matrix = self._register_matrix(matrix.to(self.device), _MatrixHandle(self._engine.csr))
            else:
                matrix = _MatrixHandle(self._engine.csr, vals, vals_t)
            current = e0
            cache = [current]
            for _ in range(self.n_layers):
                current = self.layer_aggregation(matrix, current)
                cache.append(current)
            return torch.split(self.layer_combination(cache), [self.n_users, self.n_items])
        if torch.is_grad_enabled() and e0.requires_grad:
            out = _Propagate.apply(e0, self, vals, vals_t)
        else:
            out = self._engine.forward(e0.detach(), self.n_layers, single=self._single, exact=self.exact, vals=vals)
        return torch.split(out, [self.n_users, self.n_items])

    def _coo_indices(self):
        if getattr(self, '_coo_idx_dev', None) is None:
            self._coo_idx_dev = torch.from_numpy(self.graph.to_coo()[0]).to(self.device)
        return self._coo_idx_dev

    def _dropped_coo(self, vals, vals_t):
        """base_model.py:77-86 as a tensor, for an overridden layer_aggregation: the kept entries of this step's mask (their
        values already divided by 1 - p), coalesced order.  Registered, so that a `super().layer_aggregation(...)` inside the
        override runs on the engine's CSR with the same values instead of converting the tensor."""
        keep = vals.vals != 0          # stored values are products of positive factors: a zero is a dropped entry
        idx = self._coo_indices()
        n = self.graph.n
        m = torch.sparse_coo_tensor(idx[:, keep], vals.vals[keep], (n, n), is_coalesced=True)
        return self._register_matrix(m, _MatrixHandle(self._engine.csr, vals, vals_t))

    def _register_matrix(self, tensor, resolved):
        cache = self.__dict__.setdefault('_matrix_cache', {})
        for key in [k for k, (ref, _) in cache.items() if ref() is None]:
            del cache[key]
        while len(cache) >= 8:
            del cache[next(iter(cache))]
        cache[id(tensor)] = (weakref.ref(tensor), resolved)
        return tensor

    def _resolve_matrix(self, norm_matrix):
        """what `layer_aggregation` multiplies by: (DeviceCSR, per-call values or None, callable -> (transposed CSR, values))."""
        if isinstance(norm_matrix, _MatrixHandle):
            h = norm_matrix
            return h.csr, h.vals, (lambda: (h.csr, h.vals_t))
        if isinstance(norm_matrix, DeviceCSR):
            if norm_matrix is self._engine.csr:
                return norm_matrix, None, (lambda: (norm_matrix, None))
            return norm_matrix, None, self._transposer(norm_matrix)
        if isinstance(norm_matrix, torch.Tensor) and norm_matrix.layout in (torch.sparse_coo, torch.sparse_csr):
            if norm_matrix is self._norm_matrix:       # the model's own matrix: the engine's CSR is built from it
                return self._engine.csr, None, (lambda: (self._engine.csr, None))
            hit = self.__dict__.get('_matrix_cache', {}).get(id(norm_matrix))
            if hit is not None and hit[0]() is norm_matrix:
                r = hit[1]
                return self._resolve_matrix(r) if isinstance(r, (_MatrixHandle, DeviceCSR)) else r
            if norm_matrix.dim() != 2 or norm_matrix.dtype != torch.float32:
                raise TypeError(f'layer_aggregation: norm_matrix must be a 2-D float32 sparse tensor, got {tuple(norm_matrix.shape)} '
                                f'{norm_matrix.dtype}')
            coo = (norm_matrix.detach() if norm_matrix.layout == torch.sparse_coo else norm_matrix.detach().to_sparse_coo()).coalesce()
            idx = coo.indices().cpu().numpy()
            val = coo.values().cpu().numpy()
            n_rows, n_src = int(norm_matrix.shape[0]), int(norm_matrix.shape[1])
            rowptr = np.zeros(n_rows + 1, dtype=np.int64)
            np.cumsum(np.bincount(idx[0], minlength=n_rows), out=rowptr[1:])
            csr = DeviceCSR(rowptr, idx[1], val, n_src, self.device, self._split_threshold)
            self._register_matrix(norm_matrix, csr)
            return csr, None, self._transposer(csr)
        raise TypeError('layer_aggregation: norm_matrix must be a torch sparse tensor (COO or CSR, float32) or a DeviceCSR, got '
                        f'{type(norm_matrix).__name__}')

    def _transposer(self, csr):
        """lazily built CSR of the transpose of a matrix the caller brought (only a backward pass needs it)"""
        def make():
            t = getattr(csr, '_transposed', None)
            if t is None:
                import scipy.sparse as sp
                m = sp.csr_matrix((csr.vals.cpu().numpy()[:csr.nnz], csr.colidx.cpu().numpy()[:csr.nnz].astype(np.int64),
                                   csr.rowptr.cpu().numpy().astype(np.int64)), shape=(csr.n_rows, csr.n_src_rows))
                mt = m.transpose().tocsr()
                mt.sort_indices()
                t = csr._transposed = DeviceCSR(mt.indptr.astype(np.int64), mt.indices, mt.data.astype(np.float32), csr.n_rows,
                                                self.device, self._split_threshold)
            return t, None
        return make

    def layer_aggregation(self, norm_matrix, emb_matrix):
        """base_model.py:141-148: `torch.sparse.mm(norm_matrix, emb_matrix)` on the HIP path, with the matrix the CALLER passes:
        a torch sparse tensor (COO as the reference builds it, or CSR; converted to a device CSR once and remembered by
        identity -- the model's own `norm_matrix` and the dropped matrix `representation` hands out need no conversion), or a
        DeviceCSR.  Anything else is a TypeError.  Differentiable in emb_matrix."""
        csr, vals, transposed = self._resolve_matrix(norm_matrix)
        if not isinstance(emb_matrix, torch.Tensor) or emb_matrix.dim() != 2 or emb_matrix.shape[0] != csr.n_src_rows:
            raise ValueError(f'layer_aggregation: emb_matrix must be [{csr.n_src_rows}, d], got '
                             f'{tuple(getattr(emb_matrix, "shape", ()))}')
        if torch.is_grad_enabled() and emb_matrix.requires_grad:
            return _Spmm.apply(emb_matrix, self, csr, vals, transposed)
        y = torch.empty((csr.n_rows, emb_matrix.shape[1]), dtype=torch.float32, device=emb_matrix.device)
        spmm(csr, emb_matrix.detach().contiguous(), y=y, exact=self.exact, vals=vals)
        return y

    def layer_combination(self, vectors):
        """base_model.py:150-157: mean over layers == sequential sum then one division (fused in the kernel on
        the `representation` path; this standalone form exists for callers that hold the list)."""
        s = vectors[0]
        for v in vectors[1:]:
            s = s + v
        return s / torch.tensor(float(len(vectors)), device=s.device)

    def layer_combination_single(self, vectors):
        return vectors[-1]

    # ------------------------------------------------------------------ scoring (base_model.py:166-179)
    def score_pairwise(self, users_emb, items_emb, users, items):
        if torch.is_grad_enabled() and (users_emb.requires_grad or items_emb.requires_grad):
            return torch.sum(users_emb * items_emb, dim=1)   # training: autograd through a tiny elementwise op
        return scoring.score_pairwise(users_emb.contiguous(), items_emb.contiguous())

    def score_batchwise(self, users_emb, items_emb, users):
        return scoring.score_dense(users_emb.contiguous(), items_emb.contiguous())

    # ------------------------------------------------------------------ losses (base_model.py:181-210)
    def get_loss(self, data):
        """data: [batch, 2 + n_neg] rows (user, positive, negatives...) -> BPR + L2 terms."""
        if not hasattr(self, '_loss_values'):
            self._loss_values = defaultdict(float)
        data = torch.as_tensor(data)
        if data.dim() != 2 or data.shape[1] < 3:
            raise ValueError(f'get_loss expects [batch, 2 + n_neg] rows (user, positive, negatives...), got {tuple(data.shape)}')
        if data.dtype.is_floating_point or data.dtype.is_complex or data.dtype == torch.bool:
            raise TypeError(f'batch ids must be an integer tensor, got {data.dtype}')
        if self._native_loss():
            bpr, reg = _BprStep.apply(self.embedding_user.weight, self.embedding_item.weight, self, self._checked_ids(data))
            self._loss_values['bpr'] += bpr.detach()
            self._loss_values['reg'] += reg.detach()
            return bpr + reg
        cols = data.to(self.device).t()
        users, pos, negs = cols[0], cols[1], list(cols[2:])
        return self.bpr_loss(users, pos, negs) + self.reg_loss(users, pos, negs)

    def _checked_ids(self, data):
        """[2 + n_neg, batch] contiguous int64 ids on the model's device, every id inside its table: the contract of
        tgcn_bpr_pairs_f32 / tgcn_reg_rows_f32 (include/tgcn.h), which read raw 8-byte ids and add gradient rows at the
        addresses they name.  The reference's `users_emb[users]` (base_model.py:189-193) takes any integer dtype and raises
        IndexError on an id outside the table; so does this: a host batch (what a DataLoader yields) is checked before it is
        copied, without touching the GPU; a device batch is checked by two reductions whose flag is read together with the
        step's NaN flag, and its ids are clamped meanwhile so that no kernel ever sees an address outside the tables."""
        lim = torch.tensor([self.n_users] + [self.n_items] * (data.shape[1] - 1), dtype=torch.int64)
        if data.device.type == 'cpu':
            d64 = data.to(torch.int64)
            if data.numel() and bool(((d64 < 0) | (d64 >= lim)).any()):
                raise IndexError('get_loss: a user or item id of the batch is outside its embedding table')
            return d64.to(self.device).t().contiguous()
        d64 = data.to(self.device, torch.int64)
        lim = lim.to(self.device)
        bad = ((d64 < 0) | (d64 >= lim)).any()
        if getattr(self, '_in_epoch', False):
            self._bad_ids = bad if getattr(self, '_bad_ids', None) is None else (self._bad_ids | bad)
        elif bool(bad):        # get_loss called outside fit(): nobody would read the flag later
            raise IndexError('get_loss: a user or item id of the batch is outside its embedding table')
        return torch.minimum(d64.clamp_min(0), lim - 1).t().contiguous()

    def _native_loss(self):
        """The fused step applies to the plain LightGCN loss: a subclass or instance that replaces a scoring / loss member
        (LTRLinear rebinds score_pairwise, AdvSamplModel overrides get_loss) keeps the generic torch composition below."""
        cls = type(self)
        plain = all(name not in self.__dict__ and getattr(cls, name) is getattr(LightGCN, name)
                    for name in ('score_pairwise', 'bpr_loss', 'reg_loss', 'representation'))
        plain = plain and not self._overridden('layer_aggregation') and not self._overridden('layer_combination')
        return (plain and self.device.type == 'cuda' and self.emb_size <= 512 and torch.is_grad_enabled()
                and self.embedding_user.weight.requires_grad and self.embedding_item.weight.requires_grad)

    def bpr_loss(self, users, pos, negs):
        """mean over negatives of mean_b selu(s(u, neg) - s(u, pos))  (base_model.py:186-198)"""
        all_users, all_items = self.representation
        u = all_users[users]
        s_pos = self.score_pairwise(u, all_items[pos], users, pos)
        terms = [F.selu(self.score_pairwise(u, all_items[n], users, n) - s_pos).mean() for n in negs]
        loss = torch.stack(terms).sum() / len(terms)
        self._loss_values['bpr'] += loss
        return loss

    def reg_loss(self, users, pos, negs):
        """lambda / (2 b) * (|E_u[users]|^2 + |E_i[pos]|^2 + mean_neg-free |E_i[negs]|^2)  (base_model.py:200-210):
        the negatives' squared norm is that of the stacked [n_neg, b, d] block (one number), as in the reference."""
        sq = lambda t: t.norm(2).pow(2)  # noqa: E731
        total = sq(self.embedding_user(users)) + sq(self.embedding_item(pos)) + sq(self.embedding_item(torch.stack(negs))).mean()
        res = total * (self.reg_lambda / (2 * len(users)))
        self._loss_values['reg'] += res
        return res

    # ------------------------------------------------------------------ training loop (base_model.py:108-139)
    def _train_epoch(self, batches, epoch):
        self.train()
        self.training = True
        self._loss_values = defaultdict(float)
        pending = None                     # deferred mode: the previous step's flags (device scalars)
        self._bad_ids = None
        self._in_epoch = True              # device-id range flags raised inside the epoch are read with the NaN flag below
        try:
            self._run_epoch(batches, epoch, pending)
        finally:
            self._in_epoch = False

    def _run_epoch(self, batches, epoch, pending):

        def stop_if(flags):
            nan, bad = flags
            if bad is not None and bool(bad):
                raise IndexError(f'get_loss: a user or item id of a batch was outside its embedding table (epoch {epoch})')
            if bool(nan):
                raise AssertionError(f'loss is NA at epoch {epoch}')
        for data in batches:
            self.optimizer.zero_grad()
            loss = self.get_loss(data)
            flags = (loss.isnan(), self._bad_ids)
            self._bad_ids = None
            if not self.deferred_nan_check:
                stop_if(flags)             # base_model.py:123: before backward(), so a NaN never reaches the weights or Adam's moments
            loss.backward()
            self.optimizer.step()
            if self.deferred_nan_check:
                # opt-in: the flag of step t - 1 is read once step t is enqueued, so the step's only host sync never leaves the
                # GPU without queued work (~0.1 ms of a 1.5 ms step on config 2).  A NaN stops the run with the same
                # AssertionError one step later -- AFTER two optimizer steps have seen it: only for runs that discard the
                # model on that error.
                if pending is not None:
                    stop_if(pending)
                pending = flags
        if pending is not None:
            stop_if(pending)

    def fit(self, batches):
        """Adam over all parameters; every `evaluate_every` epochs: log losses, evaluate, checkpoint, early stop.
        A run that is never stopped early writes a final checkpoint (the reference's for/else)."""
        # same update rule as the reference's torch.optim.Adam (base_model.py:110).  `fused` = one launch per step on the GPU
        # instead of the default implementation's per-tensor foreach ops: same formulas, but the fused kernel evaluates
        # them in a different operation order, so the weights differ from a foreach/for-loop Adam in the last bits (they
        # are not pinned by a golden; G8 pins loss and gradient).  `fused_adam = False` restores torch's default.
        self.optimizer = torch.optim.Adam(self.parameters(), lr=self.lr, fused=self.device.type == 'cuda' and self.fused_adam)
        stopped = False
        for epoch in range(1, self.epochs + 1):
            self._train_epoch(batches, epoch)
            if epoch % self.evaluate_every != 0:
                continue
            self.logger.info(f'Epoch {epoch}: ' + ' '.join(f'{name} = {float(val.detach() if torch.is_tensor(val) else val):.4f}' for name, val in self._loss_values.items()))
            self.evaluate(epoch)
            self.checkpoint(epoch)
            if early_stop(self.metrics_logger):
                self.logger.warning(f'Early stopping triggerred at epoch {epoch}')
                stopped = True
                break
        if not stopped:
            self.checkpoint(self.epochs)

    # ------------------------------------------------------------------ evaluate / predict (base_model.py:212-276)
    @torch.no_grad()
    def evaluate(self, epoch=None):
        """base_model.py:212-233 without the list round trip: the top-k tensor `predict_tensors` leaves on the GPU goes
        straight into the metric arithmetic (metrics.ranking_metrics_device); 5 x len(k) numbers cross to the host."""
        self.eval()
        self.training = False
        _, idx = self.predict_tensors(self.test_users)
        if self._true_dev is None:
            self._true_dev = tuple(torch.from_numpy(a).to(self.device) for a in self._true_csr_host)
        results = ranking_metrics_device(self._true_dev[0], self._true_dev[1], idx, self.k, self.n_items)
        self.logger.info(' ' * 11 + ''.join([f'@{i:<6}' for i in self.k]))
        for m in results:
            self.metrics_logger[m] = np.append(self.metrics_logger[m], [results[m]], axis=0)
            self.logger.info(f'{m:11}' + ' '.join([f'{j:.4f}' for j in results[m]]))
        return results

    def _chunk_masks(self, users, step):
        """GENERATOR of (ids int64 [b], mask rowptr int32 [b + 1], mask items int32) on the device for the consecutive `step`-user
        chunks of `users`, cut out of the device mask CSR.  The whole call costs ONE host-to-device copy (the ids), made up front
        with the two row-pointer gathers; each chunk's own tensors -- a device cumsum for its row pointers, and for a
        non-contiguous id list one device gather of its items -- are built when the chunk is asked for, i.e. under the
        previous chunks' kernels, so only one chunk's gather is alive at a time and the first scoring call is issued before the
        second chunk's mask exists.  The host only does O(len(users)) numpy arithmetic on its copy of the row pointers (for the
        sizes) and never waits for the GPU."""
        rp = self._mask_rowptr_host
        rp_dev, items_dev = self._mask()
        ids_all = torch.from_numpy(np.ascontiguousarray(users)).to(self.device)
        start_all = rp_dev[ids_all]
        cnt_all = rp_dev[ids_all + 1] - start_all
        cnt_host = rp[users + 1] - rp[users]
        for j in range(0, len(users), step):
            batch = users[j:j + step]
            ids, cnt = ids_all[j:j + step], cnt_all[j:j + step]
            rowptr = torch.zeros(len(batch) + 1, dtype=torch.int32, device=self.device)
            torch.cumsum(cnt, 0, out=rowptr[1:])
            total = int(cnt_host[j:j + step].sum())
            if total == 0:
                items = items_dev[:1] if items_dev.numel() else torch.zeros(1, dtype=torch.int32, device=self.device)
            elif np.all(np.diff(batch) == 1):
                items = items_dev[int(rp[batch[0]]):int(rp[batch[-1] + 1])]
            else:
                shift = torch.repeat_interleave(start_all[j:j + step] - rowptr[:-1].to(torch.int64), cnt, output_size=total)
                items = items_dev[shift + torch.arange(total, device=self.device)]
            yield ids, rowptr, items

    def _batch_mask(self, batch_users, ids=None):
        """(mask rowptr, mask items) of one batch (see _chunk_masks)"""
        batch_users = np.asarray(batch_users, dtype=np.int64)
        if len(batch_users) == 0:
            return torch.zeros(1, dtype=torch.int32, device=self.device), torch.zeros(1, dtype=torch.int32, device=self.device)
        _, rowptr, items = next(self._chunk_masks(batch_users, len(batch_users)))
        return rowptr, items

    @torch.no_grad()
    def predict(self, users, save: bool = False, with_scores: bool = False):
        users = np.asarray(list(users) if not isinstance(users, np.ndarray) else users, dtype=np.int64)
        val, idx = self.predict_tensors(users)
        predictions = idx.tolist() if idx is not None else []
        scores = val.tolist() if val is not None else []
        if save:
            self._save_predictions(users, predictions, scores)
        if with_scores:
            return predictions, scores
        return predictions

    @torch.no_grad()
    def predict_tensors(self, users):
        """The device part of `predict`: (scores [n, kmax] fp32, items [n, kmax] int64) on the GPU, or (None, None)
        for an empty user list.  `predict` adds the reference's list conversion (base_model.py:265-266)."""
        self.training = False
        users = np.asarray(users, dtype=np.int64)
        kmax = max(self.k)
        y_val, y_idx = [], []
        users_emb, items_emb = self.representation
        users_emb, items_emb = users_emb.contiguous(), items_emb.contiguous()
        custom = 'score_batchwise' in self.__dict__ or type(self).score_batchwise is not LightGCN.score_batchwise
        # the reference scores `batch_size` users per step (base_model.py:245) to bound its [B, I] matrix; the
        # fused path has no such matrix, so it takes larger chunks (same results, fewer launches)
        step = self.batch_size if custom else max(self.batch_size, self.predict_chunk)
        # consecutive chunks are independent: they are issued round-robin on a few HIP streams (own scratch each), so
        # one chunk's small selection kernels run under the next chunk's GEMM (+30-45 % measured)
        main = torch.cuda.current_stream(self.device)
        streams = self._predict_streams()
        if not custom:      # (calls that run for milliseconds gain nothing from a third and fourth in flight: scoring.calls_in_flight)
            streams = streams[:scoring.calls_in_flight(min(step, max(len(users), 1)), self.n_items, len(streams))]
        # candidates from the bf16 pass, scores and order from the fp32 chains: the same lists bit for bit (scoring.score_topk);
        # the item-side factor of its error bound is computed once for the whole predict call
        prefilter = bool(getattr(self, 'score_prefilter', True)) and not custom
        item_pack = scoring.item_pack(items_emb) if prefilter and len(users) else None
        for n, (ids, rp, it) in enumerate(self._chunk_masks(users, step)):
            slot = n % len(streams)
            side = streams[slot]
            side.wait_stream(main)          # inputs (and the representation) are produced on the main stream
            with torch.cuda.stream(side):
                if custom:   # an override (e.g. LTR) returns the [B, I] matrix; mask + top-k stay on the HIP path
                    rating = self.score_batchwise(users_emb[ids], items_emb, ids).contiguous()
                    scoring.mask_train(rating, rp, it)                   # base_model.py:257-258
                    v, i = scoring.topk(rating, kmax, round4=True)       # base_model.py:261-263
                else:        # base_model.py:254-263 in one fused pass: gather + GEMM + mask + top-k + round
                    v, i = scoring.score_topk(users_emb, items_emb, kmax, user_ids=ids, mask_rowptr=rp, mask_items=it,
                                              round4=True, slot=slot, prefilter=prefilter, item_pack=item_pack)
            for t in (ids, rp, it, v, i):
                t.record_stream(side)
            y_val.append(v)
            y_idx.append(i)
        for side in streams:
            main.wait_stream(side)
        if not y_idx:
            return None, None
        return torch.cat(y_val), torch.cat(y_idx)

    def _save_predictions(self, users, predictions, scores):
        """predictions.tsv in the reference's format (base_model.py:268-273): original ids, python-list cells."""
        import pandas as pd
        # a row with fewer than k rankable (non-NaN) scores ends in (-inf, TGCN_NO_ITEM) fillers (include/tgcn.h): not items --
        # they are left out of the row's lists (the reference's torch.topk would rank the NaN scores first there; its
        # NaN-free loss assertion, base_model.py:123, keeps such tables out of a real run)
        n_items = self.n_items
        keep = [[0 <= i < n_items for i in row] for row in predictions]
        pred_unmapped = [[self.item_mapping_dict[i] for i, k in zip(row, kp) if k] for row, kp in zip(predictions, keep)]
        if scores:
            scores = [[v for v, k in zip(row, kp) if k] for row, kp in zip(scores, keep)]
        users_unmapped = [self.user_mapping_dict[u] for u in users.tolist()]
        path = os.path.join(self.save_path, 'predictions.tsv')
        pd.DataFrame({'user_id': users_unmapped, 'y_pred': pred_unmapped, 'scores': scores}).to_csv(path, sep='\t', index=False)
        self.logger.info(f'Predictions are saved in `{path}`')

    # ------------------------------------------------------------------ checkpoints (base_model.py:278-299)
    def load_model(self, load_path):
        """`load_path`: a state_dict file, or a run directory (then its best.pkl).  The loaded model is evaluated
        once and the metric history reset, as the reference does."""
        if load_path is None:
            self.logger.info(f'Created model {self.uid}')
            return
        path = os.path.join(load_path, 'best.pkl') if os.path.isdir(load_path) else load_path
        self.logger.info(f'Loading model {path}')
        self.load_state_dict(torch.load(path, map_location=self.device))
        self.logger.info('Performance of the loaded model:')
        self.evaluate()
        self.metrics_logger = {m: np.zeros((0, len(self.k))) for m in self.metrics}

    def checkpoint(self, epoch):
        """latest_checkpoint.pkl every time; best.pkl when the newest recall@k[0] is the best seen so far."""
        if not self.save:
            return
        latest = os.path.join(self.save_path, 'latest_checkpoint.pkl')
        torch.save(self.state_dict(), latest)
        history = self.metrics_logger[self.metrics[0]]
        if len(history) and history[-1][0] >= history[:, 0].max():
            self.logger.info(f'Updating best model at epoch {epoch}')
            shutil.copyfile(latest, os.path.join(self.save_path, 'best.pkl'))


def get_class(name):
    """Registry in the shape of the reference's main.get_class (main.py:16-22): name -> [DatasetCls, ModelCls]."""
    from .interactions import InteractionData
    table = {'lgcn': [InteractionData, LightGCN]}
    from .adv_sampling import AdvSamplData, AdvSamplModel
    from .ltr import LTRData, LTRLinear, LTRLinearWPop
    table['adv_sampling'] = [AdvSamplData, AdvSamplModel]
    table['ltr_linear'] = [LTRData, LTRLinear]
    table['ltr_pop'] = [LTRData, LTRLinearWPop]
    return table[name]
