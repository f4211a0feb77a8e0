"""ltr_linear on the HIP path: a linear head over five dot-product features of frozen LightGCN embeddings and
384-d text vectors (BASELINE config 5).

Drop-in for TextGCN/ltr_models.py `LTRLinear` (member names, rebinding order and state_dict keys kept).  The
reference evaluates 5 GEMMs, concatenates a [B, I, 5] tensor and applies nn.Linear over its last dimension
(ltr_models.py:131-146,200-204); the layers carry no activation (ltr_models.py:186-190), so any depth is one affine
map and the whole batchwise score is ONE GEMM of width d + 2t + 1: tgcn_ltr_fold_users_f32 /
tgcn_ltr_pack_items_f32 build its operands, the ordinary scoring kernels run it.
"""
import ctypes

import numpy as np
import torch
from torch import nn

from . import _capi, scoring
from .interactions import InteractionData
from .model import LightGCN

FEATURE_NAMES = ['lightgcn score', 'reviews', 'desc', 'reviews-description', 'description-reviews']   # ltr_models.py:71-77


class LTRData(InteractionData):
    """InteractionData + the four text tables LTRBase reads from its dataset (ltr_models.py:49-55).

    The reference derives them with SBERT + pandas (reviews_models.py, kg_models.py, ltr_models.py:19-35) -- one-off
    host preprocessing that is out of this path's scope -- so they are taken as given: tensors passed in, or a
    `ltr_tables.pt` dict in the data folder with keys items_as_desc, items_as_avg_reviews [I, t],
    users_as_avg_reviews, users_as_avg_desc [U, t]."""
    TABLES = ('items_as_desc', 'items_as_avg_reviews', 'users_as_avg_reviews', 'users_as_avg_desc')

    def __init__(self, params=None, tables=None, **kw):
        super().__init__(params, **kw)
        if tables is None:
            import os
            path = os.path.join(self.path, 'ltr_tables.pt')
            if not os.path.exists(path):
                raise FileNotFoundError(f'{path}: LTRData needs the four precomputed text tables {self.TABLES}')
            tables = torch.load(path, map_location='cpu')
        for name in self.TABLES:
            t = torch.as_tensor(tables[name], dtype=torch.float32)
            want = self.n_items if name.startswith('items') else self.n_users
            if t.dim() != 2 or t.shape[0] != want:
                raise ValueError(f'{name} must be [{want}, t], got {tuple(t.shape)}')
            setattr(self, name, t)
        # ltr_pop (LTRLinearWPop) additionally reads the dataset's two popularity columns (reviews_models.py:100-113)
        for name, want in (('popularity_users', self.n_users), ('popularity_items', self.n_items)):
            if name in tables:
                t = torch.as_tensor(tables[name], dtype=torch.float32).reshape(-1, 1)
                if t.shape[0] != want:
                    raise ValueError(f'{name} must have {want} rows, got {t.shape[0]}')
                setattr(self, name, t)


class _PairFeatures(torch.autograd.Function):
    """[n, 5] pairwise features (tgcn_ltr_pair_features_f32).  Only the first feature depends on trainable tensors -- the gathered
    embedding rows, when the base model is not frozen: d f0 / d e_u = e_i and vice versa; the text tables are constants."""

    @staticmethod
    def forward(ctx, ue, ie, model, users, items):
        ue, ie = ue.contiguous(), ie.contiguous()
        n = users.numel()
        feats = torch.empty((n, 5), dtype=torch.float32, device=ue.device)
        rc = _capi.lib().tgcn_ltr_pair_features_f32(_capi.ptr(ue), _capi.ptr(ie), _capi.ptr(model.users_as_avg_reviews),
                                                    _capi.ptr(model.users_as_avg_desc), _capi.ptr(model.items_as_avg_reviews),
                                                    _capi.ptr(model.items_as_desc), _capi.ptr(users), _capi.ptr(items), n, ue.shape[1],
                                                    model.text_dim, _capi.ptr(feats), _capi.current_stream(ue.device))
        _capi.check(rc, 'tgcn_ltr_pair_features_f32')
        ctx.save_for_backward(ue, ie)
        return feats

    @staticmethod
    def backward(ctx, g):
        ue, ie = ctx.saved_tensors
        g0 = g[:, :1]
        return (g0 * ie if ctx.needs_input_grad[0] else None), (g0 * ue if ctx.needs_input_grad[1] else None), None, None, None


class LTRLinear(LightGCN):
    """reference: TextGCN/ltr_models.py:38-210 (LTRBase + LTRLinear)."""

    ltr_predict_chunk = 8192    # users per folded scoring call (their [B, K] operand is built per call)
    predict_streams = 2         # the wide filter owns its CUs: a third call in flight only queues (config 5: 33.1 ms against 34.9-36.4
                                # with four, profiles/r04_c5_call_sweep.jsonl)

    def __init__(self, params, dataset):
        super().__init__(params, dataset)
        # the reference rebinds on the INSTANCE after construction so that the base model loaded inside
        # __init__ is evaluated with the plain LightGCN scoring (ltr_models.py:175-179)
        self.evaluate = self.evaluate_ltr
        self.score_pairwise = self.score_pairwise_ltr
        self.score_batchwise = self.score_batchwise_ltr

    def _copy_params(self, params):
        super()._copy_params(params)
        self.load_base = getattr(params, 'load_base', None)
        self.freeze = getattr(params, 'freeze', False)
        self._ltr_layers = list(getattr(params, 'ltr_layers', []) or [])

    def _copy_dataset_params(self, dataset):
        super()._copy_dataset_params(dataset)
        to = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous()  # noqa: E731
        self.items_as_avg_reviews = to(dataset.items_as_avg_reviews)
        self.users_as_avg_reviews = to(dataset.users_as_avg_reviews)
        self.users_as_avg_desc = to(dataset.users_as_avg_desc)
        self.items_as_desc = to(dataset.items_as_desc)
        self.all_items = dataset.all_items
        self.text_dim = self.items_as_desc.shape[1]

    def _init_embeddings(self, emb_size):
        super()._init_embeddings(emb_size)
        if self.freeze:   # ltr_models.py:57-61
            self.embedding_user.requires_grad_(False)
            self.embedding_item.requires_grad_(False)

    def _add_vars(self, params):
        super()._add_vars(params)
        if self.load_base:   # ltr_models.py:66-68: before the scoring functions are rebound
            self.load_model(self.load_base)
        self.feature_names = list(FEATURE_NAMES)
        self._setup_layers(params)

    def _setup_layers(self, params):
        sizes = [len(self.feature_names)] + self._ltr_layers + [1]   # ltr_models.py:186-190
        self.layers = nn.Sequential(*[nn.Linear(i, j) for i, j in zip(sizes, sizes[1:])]).to(self.device)

    # ------------------------------------------------------------------ the affine map the layers amount to
    def effective_weights(self):
        """(w [5], b) with layers(f) == f . w + b.  Composes the activation-free Linear stack exactly as the
        forward pass would (W_n ... W_1, biases pushed through)."""
        w = None
        b = None
        for lin in self.layers:
            wt, bs = lin.weight.detach().double(), lin.bias.detach().double()
            if w is None:
                w, b = wt, bs
            else:
                w, b = wt @ w, wt @ b + bs
        return w.reshape(-1).float().cpu().numpy(), float(b.reshape(-1)[0])

    # ------------------------------------------------------------------ folded operands
    def _k(self):
        return _capi.lib().tgcn_ltr_folded_width(self.emb_size, self.text_dim)

    def _fold_users(self, users_emb, emb_ids, text_ids, wb=None):
        """[n, K] user operand of the folded GEMM.  wb = effective_weights(): a device-to-host read (a stream sync), so a
        caller that folds chunk after chunk reads it ONCE and passes it in (predict_tensors)."""
        w, b = wb if wb is not None else self.effective_weights()
        n = emb_ids.numel() if emb_ids is not None else users_emb.shape[0]
        out = torch.empty((n, self._k()), dtype=torch.float32, device=self.device)
        w5 = (ctypes.c_float * 5)(*[float(x) for x in w[:5]])   # the five dot-product features; subclasses add columns
        rc = _capi.lib().tgcn_ltr_fold_users_f32(_capi.ptr(users_emb), _capi.ptr(self.users_as_avg_reviews),
                                                 _capi.ptr(self.users_as_avg_desc), _capi.ptr(emb_ids), _capi.ptr(text_ids), n,
                                                 self.emb_size, self.text_dim, w5, b, _capi.ptr(out), _capi.current_stream(self.device))
        _capi.check(rc, 'tgcn_ltr_fold_users_f32')
        return out

    def _pack_items(self, items_emb):
        """[I, K] item operand of the folded GEMM.  Never cached across calls: the propagated tables are rewritten in
        place by HIP kernels (no torch version bump) at addresses the caching allocator reuses, so no key over
        (pointer, version) can tell a fresh table from a stale one; predict_tensors packs once per call and passes it on."""
        out = torch.empty((self.n_items, self._k()), dtype=torch.float32, device=self.device)
        rc = _capi.lib().tgcn_ltr_pack_items_f32(_capi.ptr(items_emb), _capi.ptr(self.items_as_avg_reviews),
                                                 _capi.ptr(self.items_as_desc), self.n_items, self.emb_size, self.text_dim,
                                                 _capi.ptr(out), _capi.current_stream(self.device))
        _capi.check(rc, 'tgcn_ltr_pack_items_f32')
        return out

    # ------------------------------------------------------------------ scoring (ltr_models.py:200-210)
    def score_batchwise_ltr(self, users_emb, items_emb, users):
        """[B, I] scores.  users_emb are the already gathered rows (reference calling convention)."""
        users = torch.as_tensor(users, dtype=torch.int64, device=self.device).contiguous()
        ua = self._fold_users(users_emb.contiguous(), None, users)
        return scoring.score_dense(ua, self._pack_items(items_emb.contiguous()))

    def _pair_ids(self, users, items, n_rows):
        """contiguous int64 device ids inside the tables (the kernel reads text rows by raw id): any integer dtype, any device;
        an id outside its table raises IndexError as the reference's tensor indexing would (checked on the host copy the ids
        arrive as, or -- device ids -- clamped here and reported by fit()'s flag read, like get_loss's; outside a training
        epoch the flag is read right here)."""
        out = []
        for ids, lim in ((users, self.n_users), (items, self.n_items)):
            ids = torch.as_tensor(ids)
            if ids.dtype.is_floating_point or ids.dtype == torch.bool or ids.numel() != n_rows:
                raise TypeError('users / items must be integer id tensors with one id per gathered row')
            if ids.device.type == 'cpu':
                if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= lim):
                    raise IndexError('score_pairwise: id outside its table')
                ids = ids.to(self.device, torch.int64)
            else:
                ids = ids.to(self.device, torch.int64)
                bad = ((ids < 0) | (ids >= lim)).any()
                if getattr(self, '_in_epoch', False):   # fit(): the flag is read with the step's NaN flag (no extra sync)
                    self._bad_ids = bad if getattr(self, '_bad_ids', None) is None else (self._bad_ids | bad)
                elif bool(bad):                          # anywhere else nobody would read it: one flag read, then raise
                    raise IndexError('score_pairwise: id outside its table')
                ids = ids.clamp(0, lim - 1)
            out.append(ids.reshape(-1).contiguous())
        return out

    def score_pairwise_ltr(self, users_emb, items_emb, users, items):
        """[n, 1] scores of gathered (user, item) rows (training batches; ltr_models.py:206-210): the five features by ONE kernel
        (tgcn_ltr_pair_features_f32: no [n, 384] text gathers, no five row-dot ops), nn.Linear and its autograd by torch."""
        users, items = self._pair_ids(users, items, users_emb.shape[0])
        feats = _PairFeatures.apply(users_emb, items_emb, self, users, items)                         # :148-166
        return self.layers(feats)

    def evaluate_ltr(self, *args, **kwargs):
        if len(self.layers) == 1:   # ltr_models.py:192-198
            self.logger.info('Feature weights from the top layer:')
            for f, w in zip(self.feature_names, self.layers[0].weight.tolist()[0]):
                self.logger.info(f'{f:<20} {w:.4}')
        return LightGCN.evaluate(self, *args, **kwargs)

    @torch.no_grad()
    def predict_tensors(self, users):
        """Device part of predict (see LightGCN.predict_tensors) with the folded LTR score."""
        users_np = np.asarray(users, dtype=np.int64)
        if 'score_batchwise' not in self.__dict__:   # still inside __init__: the loaded base model is evaluated as LightGCN
            return LightGCN.predict_tensors(self, users_np)
        self.training = False
        kmax = max(self.k)
        users_emb, items_emb = self.representation
        users_emb, items_emb = users_emb.contiguous(), items_emb.contiguous()
        ia = self._pack_items(items_emb)
        wb = self.effective_weights()      # the one device-to-host read of the call, before the first chunk is issued
        # the folded operands are 896 / 960 wide: the bf16 candidate pass takes them too (same lists as the fp32 path, bit for bit)
        prefilter = bool(self.score_prefilter) and len(users_np) > 0
        pack = scoring.item_pack(ia) if prefilter else None
        y_val, y_idx = [], []
        main = torch.cuda.current_stream(self.device)
        streams = self._predict_streams()       # chunks round-robin on a few streams, as in LightGCN.predict_tensors
        step = max(self.batch_size, self.ltr_predict_chunk)   # no [B, I] matrix to bound: larger calls, fewer launches
        for n, (ids, rp, it) in enumerate(self._chunk_masks(users_np, step)):
            slot = n % len(streams)
            side = streams[slot]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ua = self._fold_users(users_emb, ids, ids, wb)
                v, i = scoring.score_topk(ua, ia, kmax, mask_rowptr=rp, mask_items=it, round4=True, slot=slot, prefilter=prefilter,
                                          item_pack=pack)
            for t in (ids, rp, it, ua, v, i):
                t.record_stream(side)
            y_val.append(v)
            y_idx.append(i)
        for side in streams:
            main.wait_stream(side)
        if not y_idx:
            return None, None
        return torch.cat(y_val), torch.cat(y_idx)


class LTRLinearWPop(LTRLinear):
    """reference: TextGCN/ltr_models.py:213-241 (registry name `ltr_pop`): LTRLinear plus two scalar features, the
    user's and the item's popularity as the dataset provides them (`dataset.popularity_users [U, 1]`,
    `dataset.popularity_items [I, 1]`).  With layers(f) = f . w + b the two extra terms are
    (w5 * pop_u[u]) * 1 + w6 * pop_i[i]: two more columns of the folded GEMM -- the user
    operand carries (w5 * pop_u[u], w6), the item operand (1, pop_i[i]) -- in the zero padding the folded width leaves
    (or in a widened operand when fewer than two padding columns exist)."""

    def _copy_dataset_params(self, dataset):
        super()._copy_dataset_params(dataset)
        to = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.device).reshape(-1, 1).contiguous()  # noqa: E731
        self.popularity_users = to(dataset.popularity_users)   # ltr_models.py:216-219
        self.popularity_items = to(dataset.popularity_items)

    def _setup_layers(self, params):
        self.feature_names += ['user popularity', 'item popularity']   # ltr_models.py:221-223
        super()._setup_layers(params)

    def _pop_columns(self):
        """(first popularity column, operand width)"""
        base = self.emb_size + 2 * self.text_dim + 1
        k = self._k()
        return base, k if k >= base + 2 else k + 64

    def _widen(self, t):
        _, k2 = self._pop_columns()
        if t.shape[1] == k2:
            return t
        wide = torch.zeros((t.shape[0], k2), dtype=torch.float32, device=self.device)
        wide[:, :t.shape[1]] = t
        return wide

    def _fold_users(self, users_emb, emb_ids, text_ids, wb=None):
        wb = wb if wb is not None else self.effective_weights()
        out = self._widen(super()._fold_users(users_emb, emb_ids, text_ids, wb))
        w, _ = wb
        c, _ = self._pop_columns()
        pop = self.popularity_users[:, 0] if text_ids is None else self.popularity_users[text_ids, 0]
        out[:, c] = float(w[5]) * pop
        out[:, c + 1] = float(w[6])
        return out

    def _pack_items(self, items_emb):
        out = self._widen(super()._pack_items(items_emb)).clone()
        c, _ = self._pop_columns()
        out[:, c] = 1.0
        out[:, c + 1] = self.popularity_items[:, 0]
        return out

    def score_pairwise_ltr(self, users_emb, items_emb, users, items):
        users, items = self._pair_ids(users, items, users_emb.shape[0])
        feats = torch.cat([_PairFeatures.apply(users_emb, items_emb, self, users, items),
                           self.popularity_users[users], self.popularity_items[items]], dim=1)   # ltr_models.py:233-241
        return self.layers(feats)
