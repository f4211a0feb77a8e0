#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference on CPU.

Runs only in the build container (needs /root/reference); the produced ``*.npz`` files are
pure data (inputs + expected outputs) and are what travels to the GPU box.  Nothing of the
reference's source is copied: it is imported from where it lies.

Recipe: plain ``import TextGCN`` fails with ModuleNotFoundError for two
packages that are not on the LightGCN arithmetic path:
  * ``sentence_transformers`` (TextGCN/utils.py:8) - only used to embed raw text
    (utils.py:109), never reached here because cached ``.torch`` tensors exist
    (utils.py:102-103) or the model is ``lgcn``.
  * ``dgl`` (TextGCN/dataset.py:6) - used only at dataset.py:142-149 to turn the train
    (user, item) pairs into a 0/1 scipy COO matrix U x I.  The stand-in below returns exactly
    that matrix: rows = users, cols = items, one unit entry per train row, input order.
Everything numerically relevant downstream (scipy normalisation, float32 conversion,
torch coalesce, torch.sparse.mm, mean/stack, matmul, -inf mask, topk, round, nn.Linear) is
the reference's own code running on this container's torch CPU build.
Also: ``np.NINF`` was removed in numpy 2 (base_model.py:258) -> shim.

Usage:  python tests/golden/make_golden.py        (writes tests/golden/*.npz)
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp
import torch

REF = '/root/reference'
OUT = os.environ.get('GOLDEN_OUT') or os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stand-ins
def _install_stubs():
    st = types.ModuleType('sentence_transformers')

    class SentenceTransformer:  # never instantiated on the paths we run
        def __init__(self, *a, **k):
            raise RuntimeError('SentenceTransformer stand-in must not be reached')

    st.SentenceTransformer = SentenceTransformer
    sys.modules['sentence_transformers'] = st

    dgl = types.ModuleType('dgl')

    class _G:
        def __init__(self, data):
            self._data = data
            self.ndata = {}

        def adj_external(self, etype=None, scipy_fmt='coo', ctx=None):
            assert etype == 'bought' and scipy_fmt == 'coo'
            src, dst = self._data[('user', 'bought', 'item')]
            src = np.asarray(src, dtype=np.int64)
            dst = np.asarray(dst, dtype=np.int64)
            n_u = int(src.max()) + 1
            n_i = int(dst.max()) + 1
            return sp.coo_matrix((np.ones(len(src), dtype=np.int64), (src, dst)), shape=(n_u, n_i))

    dgl.heterograph = lambda data, device=None: _G(data)
    sys.modules['dgl'] = dgl
    if not hasattr(np, 'NINF'):
        np.NINF = -np.inf
    sys.path.insert(0, REF)


_install_stubs()
import TextGCN  # noqa: E402  (the reference, imported from /root/reference)
from TextGCN.parser import parse_args  # noqa: E402


# --------------------------------------------------------------------------- helpers
def exact_embedding(n, d, salt):
    """Deterministic fp32 table, exactly representable, recomputable anywhere (no RNG state).

    value = ((h mod 2^16) - 2^15) / 2^18  in (-0.125, 0.125), h = integer hash of (row, col, salt).
    """
    r = np.arange(n, dtype=np.uint64)[:, None]
    c = np.arange(d, dtype=np.uint64)[None, :]
    h = (r * np.uint64(2654435761) + c * np.uint64(40503) + np.uint64(salt) * np.uint64(97531)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    v = (h & np.uint64(0xFFFF)).astype(np.int64) - 32768
    return (v.astype(np.float32) / np.float32(262144.0)).astype(np.float32)


def write_tsvs(folder, train_pairs, test_pairs, uw=6, iw=6):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, 'train.tsv'), 'w') as f:
        f.write('user_id\tasin\n')
        for u, i in train_pairs:
            f.write(f'u{u:0{uw}d}\ti{i:0{iw}d}\n')
    with open(os.path.join(folder, 'test.tsv'), 'w') as f:
        f.write('user_id\tasin\n')
        for u, i in test_pairs:
            f.write(f'u{u:0{uw}d}\ti{i:0{iw}d}\n')


def run_args(extra, data, workdir):
    os.chdir(workdir)  # the parser creates runs/<data>/<uid>/ relative to cwd (parser.py:169-170)
    argv = ['--data', data, '--gpu', '', '--quiet', '--slurm', '--uid', 'golden'] + extra
    return parse_args(argv)


def coo_of(norm_matrix):
    idx = norm_matrix._indices().numpy().astype(np.int64)
    val = norm_matrix._values().numpy().astype(np.float32)
    return idx, val


def layers_of(model):
    """E^0..E^K exactly as base_model.py:93-105 computes them (eval mode -> undropped matrix)."""
    with torch.no_grad():
        cur = model.embedding_matrix
        out = [cur.numpy().copy()]
        for _ in range(model.n_layers):
            cur = model.layer_aggregation(model.norm_matrix, cur)
            out.append(cur.numpy().copy())
    return out


def forward_bundle(model, dataset, users, prefix=''):
    """Everything the parity tests compare: per-layer, representation, rating, masked, topk."""
    b = {}
    layers = layers_of(model)
    for k, e in enumerate(layers):
        b[f'{prefix}layer{k}'] = e
    with torch.no_grad():
        ue, ie = model.representation
        b[f'{prefix}users_emb'] = ue.numpy().copy()
        b[f'{prefix}items_emb'] = ie.numpy().copy()
        users = np.asarray(users)
        rating = model.score_batchwise(ue[users], ie, users)
        b[f'{prefix}rating'] = rating.numpy().copy()
        masked = rating.clone()
        exploded = model.train_user_dict[users].reset_index(drop=True).explode()
        masked[exploded.index, exploded.tolist()] = -np.inf
        b[f'{prefix}masked'] = masked.numpy().copy()
        pred, scores = model.predict(users, with_scores=True)
        b[f'{prefix}topk_idx'] = np.asarray(pred, dtype=np.int64)
        b[f'{prefix}topk_val'] = np.asarray(scores, dtype=np.float32)
    return b


def dataset_bundle(dataset):
    idx, val = coo_of(dataset.norm_matrix)
    tu = dataset.train_df.user_id.values.astype(np.int64)
    ti = dataset.train_df.asin.values.astype(np.int64)
    return {
        'n_users': np.int64(dataset.n_users), 'n_items': np.int64(dataset.n_items),
        'train_u': tu, 'train_i': ti,
        'test_u': dataset.test_df.user_id.values.astype(np.int64),
        'test_i': dataset.test_df.asin.values.astype(np.int64),
        'norm_idx': idx, 'norm_val': val,
    }


def set_weights(model, eu, ei):
    with torch.no_grad():
        model.embedding_user.weight.copy_(torch.from_numpy(eu))
        model.embedding_item.weight.copy_(torch.from_numpy(ei))


ONLY = os.environ.get('GOLDEN_ONLY')   # 'g4' / 'g7': regenerate just that fixture (the synth-60x40 data is rebuilt either way)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f'wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)')


# --------------------------------------------------------------------------- G1: data/dummy
def g1_dummy(work):
    data = os.path.join(work, 'dummy')
    shutil.copytree(os.path.join(REF, 'data/dummy'), data)
    # the two TSVs are data (16 rows); keep them as a fixture for the end-to-end test
    os.makedirs(os.path.join(OUT, 'dummy'), exist_ok=True)
    for f in ('train.tsv', 'test.tsv'):
        shutil.copyfile(os.path.join(data, f), os.path.join(OUT, 'dummy', f))
    from transformers import set_seed
    args = run_args(['--model', 'lgcn', '--no_train', '--predict', '-k', '1', '2', '3'], data, work)
    set_seed(args.seed)  # main.py:28
    ds = TextGCN.BaseDataset(args)
    model = TextGCN.BaseModel(args, ds)
    b = dataset_bundle(ds)
    b['emb_user'] = model.embedding_user.weight.detach().numpy().copy()
    b['emb_item'] = model.embedding_item.weight.detach().numpy().copy()
    b.update(forward_bundle(model, ds, np.arange(ds.n_users)))
    res = model.evaluate()
    for m, v in res.items():
        b[f'metric_{m}'] = np.asarray(v, dtype=np.float64)
    b['k'] = np.asarray(args.k)
    b['test_users'] = model.test_users.astype(np.int64)
    # predictions.tsv as main.py:43 writes it
    model.predict(range(ds.n_users), with_scores=True, save=True)
    with open(os.path.join(args.save_path, 'predictions.tsv')) as f:
        b['predictions_tsv'] = np.frombuffer(f.read().encode(), dtype=np.uint8)
    b['user_org'] = ds.user_mapping.org_id.values.astype('U')
    b['item_org'] = ds.item_mapping.org_id.values.astype('U')
    save('g1_dummy.npz', **b)


# --------------------------------------------------------------------------- G2: synth 60x40
def synth_pairs(rng, n_u, n_i, per_user_lo, per_user_hi, n_test_users):
    train, test = [], []
    for u in range(n_u):
        k = int(rng.integers(per_user_lo, per_user_hi + 1))
        items = rng.choice(n_i, size=k, replace=False)
        train += [(u, int(i)) for i in items]
    # make sure every item occurs
    seen = {i for _, i in train}
    for i in range(n_i):
        if i not in seen:
            train.append((int(rng.integers(0, n_u)), i))
    tset = set(train)
    for u in rng.choice(n_u, size=n_test_users, replace=False):
        for _ in range(2):
            i = int(rng.integers(0, n_i))
            if (int(u), i) not in tset:
                test.append((int(u), i))
                tset.add((int(u), i))
    return sorted(set(train)), sorted(set(test))


def g2_synth(work, write=True):
    rng = np.random.default_rng(2)
    train, test = synth_pairs(rng, 60, 40, 3, 9, 30)
    data = os.path.join(work, 'synth60')
    write_tsvs(data, train, test)
    if not write:
        return data
    out = {}
    variants = {
        'a': (['--emb_size', '64', '--n_layers', '3'], 64),
        'single': (['--emb_size', '64', '--n_layers', '3', '--single'], 64),
        'k4d128': (['--emb_size', '128', '--n_layers', '4'], 128),
        'd48': (['--emb_size', '48', '--n_layers', '2'], 48),
    }
    ds = None
    for name, (extra, d) in variants.items():
        args = run_args(['--model', 'lgcn', '--no_train', '-k', '5', '10', '--batch_size', '32'] + extra, data, work)
        ds = TextGCN.BaseDataset(args)
        model = TextGCN.BaseModel(args, ds)
        eu, ei = exact_embedding(ds.n_users, d, 11), exact_embedding(ds.n_items, d, 12)
        set_weights(model, eu, ei)
        fb = forward_bundle(model, ds, np.arange(ds.n_users), prefix=f'{name}_')
        out.update(fb)
        res = model.evaluate()
        for m, v in res.items():
            out[f'{name}_metric_{m}'] = np.asarray(v, dtype=np.float64)
        out[f'{name}_d'] = np.int64(d)
        out[f'{name}_n_layers'] = np.int64(args.n_layers)
    out.update(dataset_bundle(ds))
    out['test_users'] = np.sort(ds.test_df.user_id.unique()).astype(np.int64)
    save('g2_synth60.npz', **out)
    return data


# --------------------------------------------------------------------------- G3: dropout
def g3_dropout(work, data):
    args = run_args(['--model', 'lgcn', '--no_train', '-k', '5', '--dropout', '0.4'], data, work)
    ds = TextGCN.BaseDataset(args)
    model = TextGCN.BaseModel(args, ds)
    eu, ei = exact_embedding(ds.n_users, 64, 11), exact_embedding(ds.n_items, 64, 12)
    set_weights(model, eu, ei)
    nnz = ds.norm_matrix._values().shape[0]
    torch.manual_seed(123)
    u = torch.rand(nnz)  # what base_model.py:82 draws
    torch.manual_seed(123)
    dropped = model._dropout_norm_matrix  # base_model.py:77-86
    didx, dval = coo_of(dropped)
    # one training-mode forward with the same mask (representation re-draws -> reseed)
    torch.manual_seed(123)
    model.training = True
    with torch.no_grad():
        ue, ie = model.representation
    model.training = False
    save('g3_dropout.npz', rand=u.numpy(), p=np.float64(0.4), drop_idx=didx, drop_val=dval,
         users_emb=ue.numpy(), items_emb=ie.numpy())


# --------------------------------------------------------------------------- G5: medium graph
def g5_medium(work):
    rng = np.random.default_rng(5)
    n_u, n_i, nnz = 1500, 700, 30000
    # zipf-ish items so that a few rows are long
    p = (np.arange(n_i) + 1.0) ** -0.8
    p /= p.sum()
    pairs = set()
    while len(pairs) < nnz:
        u = rng.integers(0, n_u, size=nnz)
        i = rng.choice(n_i, size=nnz, p=p)
        for a, b in zip(u.tolist(), i.tolist()):
            if len(pairs) < nnz:
                pairs.add((a, b))
    seen_u = {a for a, _ in pairs}
    seen_i = {b for _, b in pairs}
    for a in range(n_u):
        if a not in seen_u:
            pairs.add((a, int(rng.integers(0, n_i))))
    for b in range(n_i):
        if b not in seen_i:
            pairs.add((int(rng.integers(0, n_u)), b))
    train = sorted(pairs)
    test = [(u, i) for (u, i) in [(int(rng.integers(0, n_u)), int(rng.integers(0, n_i))) for _ in range(400)]
            if (u, i) not in pairs]
    data = os.path.join(work, 'medium')
    write_tsvs(data, train, sorted(set(test)))
    out = {}
    ds = None
    for d, K in ((64, 3), (128, 4)):
        args = run_args(['--model', 'lgcn', '--no_train', '-k', '20', '40', '--emb_size', str(d),
                         '--n_layers', str(K), '--batch_size', '512'], data, work)
        ds = TextGCN.BaseDataset(args)
        model = TextGCN.BaseModel(args, ds)
        eu, ei = exact_embedding(ds.n_users, d, 21), exact_embedding(ds.n_items, d, 22)
        set_weights(model, eu, ei)
        layers = layers_of(model)
        with torch.no_grad():
            ue, ie = model.representation
        full = np.concatenate([ue.numpy(), ie.numpy()])
        rows = np.sort(rng.choice(full.shape[0], size=192, replace=False))
        # always include the longest rows (items) so split-row paths are pinned too
        deg = np.bincount(ds.norm_matrix._indices()[0].numpy(), minlength=full.shape[0])
        rows = np.unique(np.concatenate([rows, np.argsort(-deg)[:16]]))
        pfx = f'd{d}_'
        out[pfx + 'rows'] = rows.astype(np.int64)
        out[pfx + 'repr_rows'] = full[rows]
        out[pfx + 'lastlayer_rows'] = layers[-1][rows]
        out[pfx + 'repr_bits_sum'] = np.uint64(full.view(np.uint32).astype(np.uint64).sum())
        out[pfx + 'lastlayer_bits_sum'] = np.uint64(layers[-1].view(np.uint32).astype(np.uint64).sum())
        out[pfx + 'repr_absmax'] = np.float64(np.abs(full).max())
        users = np.arange(0, ds.n_users, 7)
        pred, scores = model.predict(users, with_scores=True)
        out[pfx + 'pred_users'] = users.astype(np.int64)
        out[pfx + 'topk_idx'] = np.asarray(pred, dtype=np.int64)
        out[pfx + 'topk_val'] = np.asarray(scores, dtype=np.float32)
        res = model.evaluate()
        for m, v in res.items():
            out[pfx + f'metric_{m}'] = np.asarray(v, dtype=np.float64)
    b = dataset_bundle(ds)
    # the 60k-entry matrix is reproducible from train pairs; keep a checksum + the values only
    out['n_users'], out['n_items'] = b['n_users'], b['n_items']
    out['train_u'], out['train_i'] = b['train_u'].astype(np.int32), b['train_i'].astype(np.int32)
    out['test_u'], out['test_i'] = b['test_u'].astype(np.int32), b['test_i'].astype(np.int32)
    out['norm_row_sum'] = np.int64(b['norm_idx'][0].sum())
    out['norm_col_weighted'] = np.int64((b['norm_idx'][1] * (np.arange(b['norm_idx'].shape[1]) % 1009)).sum())
    out['norm_val'] = b['norm_val']
    save('g5_medium.npz', **out)


# --------------------------------------------------------------------------- G6: builder corner cases
def g6_builder(work):
    """A0 alone: isolated nodes cannot occur through the TSV path (ids come from train rows), but a
    duplicate edge can; and the float64 normalisation is pinned on awkward degrees."""
    out = {}
    rng = np.random.default_rng(6)
    cases = {}
    # c0: duplicate train row (a_rc = 2)
    cases['dup'] = ([(0, 0), (0, 1), (0, 1), (1, 1), (2, 0), (2, 2), (3, 2)], 4, 3)
    # c1: star (one item connected to everybody) + chain
    cases['star'] = ([(u, 0) for u in range(50)] + [(u, 1 + u % 7) for u in range(50)], 50, 8)
    # c2: random with many distinct degrees
    pr = set()
    while len(pr) < 900:
        pr.add((int(rng.integers(0, 97)), int(rng.zipf(1.6) % 61)))
    pr |= {(u, int(rng.integers(0, 61))) for u in range(97)}
    pr |= {(int(rng.integers(0, 97)), i) for i in range(61)}
    cases['rand'] = (sorted(pr), 97, 61)
    for name, (pairs, n_u, n_i) in cases.items():
        data = os.path.join(work, f'g6_{name}')
        # test set: first train pair's user with some other item keeps the loader happy
        write_tsvs(data, pairs, [pairs[0]])
        args = run_args(['--model', 'lgcn', '--no_train', '-k', '1'], data, work)
        ds = TextGCN.BaseDataset(args)
        b = dataset_bundle(ds)
        for k in ('n_users', 'n_items', 'train_u', 'train_i', 'norm_idx', 'norm_val'):
            out[f'{name}_{k}'] = b[k]
    save('g6_builder.npz', **out)


def ltr_files(data):
    """What TextGCN.LTRDataset reads beside the two TSVs, written where its cache lookup finds them: reviews_text.tsv and
    the two cached 384-d tensors (synthetic, exact_embedding).  Returns (the rng after its draws, text width)."""
    import pandas as pd
    train = pd.read_table(os.path.join(data, 'train.tsv'), dtype=str)
    rng = np.random.default_rng(4)
    rev = train.copy()
    rev['review'] = [f'review {k}' for k in range(len(rev))]
    rev['time'] = rng.integers(0, 1000, size=len(rev))
    rev.to_csv(os.path.join(data, 'reviews_text.tsv'), sep='\t', index=False)
    emb_dir = os.path.join(data, 'embeddings')
    os.makedirs(emb_dir, exist_ok=True)
    t = 384
    torch.save(torch.from_numpy(exact_embedding(len(rev), t, 41) * 8), os.path.join(emb_dir, 'item_full_reviews_loss_repr_all-MiniLM-L6-v2_0-seed.torch'))
    n_items = train.asin.nunique()
    torch.save(torch.from_numpy(exact_embedding(n_items, t, 42) * 8), os.path.join(emb_dir, 'item_kg_repr_all-MiniLM-L6-v2_0-seed.torch'))
    return rng, t


# --------------------------------------------------------------------------- G4: LTRLinear on frozen LightGCN
def g4_ltr(work, data):
    """ltr_linear on the synth-60x40 data (needs reviews_text.tsv + cached 384-d tensors).
    The text tensors are synthetic (exact_embedding), written where the reference's cache lookup finds them
    (reviews_models.py:37-53, kg_models.py:24-31), so no SBERT model is involved."""
    rng, t = ltr_files(data)
    # base checkpoint with known weights
    args = run_args(['--model', 'lgcn', '--no_train', '-k', '5', '10', '--batch_size', '32'], data, work)
    ds0 = TextGCN.BaseDataset(args)
    base = TextGCN.BaseModel(args, ds0)
    set_weights(base, exact_embedding(ds0.n_users, 64, 11), exact_embedding(ds0.n_items, 64, 12))
    ckpt = os.path.join(work, 'base.pkl')
    torch.save(base.state_dict(), ckpt)
    args = run_args(['--model', 'ltr_linear', '--no_train', '-k', '5', '10', '--batch_size', '32', '--load_base', ckpt, '--freeze'], data, work)
    ds = TextGCN.LTRDataset(args)
    model = TextGCN.LTRLinear(args, ds)
    with torch.no_grad():
        model.layers[0].weight.copy_(torch.tensor([[0.75, -0.5, 0.25, 0.125, -0.375]]))
        model.layers[0].bias.copy_(torch.tensor([0.0625]))
    out = {'n_users': np.int64(ds.n_users), 'n_items': np.int64(ds.n_items), 't': np.int64(t),
           'items_as_desc': ds.items_as_desc.numpy(), 'items_as_avg_reviews': ds.items_as_avg_reviews.numpy(),
           'users_as_avg_reviews': ds.users_as_avg_reviews.numpy(), 'users_as_avg_desc': ds.users_as_avg_desc.numpy(),
           'w': model.layers[0].weight.detach().numpy().copy(), 'b': model.layers[0].bias.detach().numpy().copy(),
           'state_keys': np.array(sorted(model.state_dict().keys()))}
    with torch.no_grad():
        ue, ie = model.representation
        out['users_emb'], out['items_emb'] = ue.numpy().copy(), ie.numpy().copy()
        users = np.arange(ds.n_users)
        u_vecs = model.get_user_vectors(ue[users], users)
        i_vecs = model.get_item_vectors(ie, model.all_items)
        out['features'] = model.get_features_batchwise(u_vecs, i_vecs).numpy().copy()
        out['scores'] = model.score_batchwise(ue[users], ie, users).numpy().copy()
        pairs_u = rng.integers(0, ds.n_users, 200)
        pairs_i = rng.integers(0, ds.n_items, 200)
        out['pairs_u'], out['pairs_i'] = pairs_u, pairs_i
        out['pair_scores'] = model.score_pairwise(ue[pairs_u], ie[pairs_i], torch.from_numpy(pairs_u), torch.from_numpy(pairs_i)).numpy().copy()
        pred, sc = model.predict(users, with_scores=True)
        out['topk_idx'], out['topk_val'] = np.asarray(pred, dtype=np.int64), np.asarray(sc, dtype=np.float32)
    res = model.evaluate()
    for m, v in res.items():
        out[f'metric_{m}'] = np.asarray(v, dtype=np.float64)
    b = dataset_bundle(ds)
    for k in ('train_u', 'train_i', 'test_u', 'test_i'):
        out[k] = b[k]
    if ONLY in (None, 'g4'):
        save('g4_ltr.npz', **out)
    if ONLY in (None, 'g7'):
        g7_ltr_pop(work, data, ckpt, ds, rng)


# --------------------------------------------------------------------------- G7: LTRLinearWPop (registry name ltr_pop)
def g7_ltr_pop(work, data, ckpt, ds, rng):
    """ltr_models.py:213-241 on the same data: the five text/graph features plus user and item popularity as the
    dataset computes them (reviews_models.py:100-113), through a 7-input Linear."""
    args = run_args(['--model', 'ltr_pop', '--no_train', '-k', '5', '10', '--batch_size', '32', '--load_base', ckpt, '--freeze'], data, work)
    model = TextGCN.LTRLinearWPop(args, ds)
    with torch.no_grad():
        model.layers[0].weight.copy_(torch.tensor([[0.75, -0.5, 0.25, 0.125, -0.375, 1.5, -2.25]]))
        model.layers[0].bias.copy_(torch.tensor([0.0625]))
    out = {'n_users': np.int64(ds.n_users), 'n_items': np.int64(ds.n_items),
           'popularity_users': ds.popularity_users.numpy().copy(), 'popularity_items': ds.popularity_items.numpy().copy(),
           'w': model.layers[0].weight.detach().numpy().copy(), 'b': model.layers[0].bias.detach().numpy().copy(),
           'feature_names': np.array(model.feature_names), 'state_keys': np.array(sorted(model.state_dict().keys()))}
    with torch.no_grad():
        ue, ie = model.representation
        users = np.arange(ds.n_users)
        out['scores'] = model.score_batchwise(ue[users], ie, users).numpy().copy()
        pairs_u = rng.integers(0, ds.n_users, 200)
        pairs_i = rng.integers(0, ds.n_items, 200)
        out['pairs_u'], out['pairs_i'] = pairs_u, pairs_i
        out['pair_scores'] = model.score_pairwise(ue[pairs_u], ie[pairs_i], torch.from_numpy(pairs_u), torch.from_numpy(pairs_i)).numpy().copy()
        pred, sc = model.predict(users, with_scores=True)
        out['topk_idx'], out['topk_val'] = np.asarray(pred, dtype=np.int64), np.asarray(sc, dtype=np.float32)
    save('g7_ltr_pop.npz', **out)


# --------------------------------------------------------------------------- G8: training loss + gradient
def g8_loss(work, data):
    """get_loss (base_model.py:181-210) on a fixed batch: BPR + L2 values and dE0 by the reference's autograd, without
    dropout and with the captured dropout mask of seed 123 (the G3 mask), for 1 and 2 negatives per row."""
    from collections import defaultdict
    out = {}
    rng = np.random.default_rng(8)
    for name, p, n_neg in (('nodrop', 0.0, 1), ('drop', 0.4, 1), ('drop2', 0.4, 2)):
        args = run_args(['--model', 'lgcn', '--no_train', '-k', '5', '--dropout', str(p), '--neg_samples', str(n_neg)], data, work)
        ds = TextGCN.BaseDataset(args)
        model = TextGCN.BaseModel(args, ds)
        set_weights(model, exact_embedding(ds.n_users, 64, 11), exact_embedding(ds.n_items, 64, 12))
        b = 96
        users = rng.integers(0, ds.n_users, b)
        pos = np.array([rng.choice(ds.train_user_dict[u]) for u in users])
        negs = rng.integers(0, ds.n_items, (b, n_neg))
        batch = torch.from_numpy(np.concatenate([users[:, None], pos[:, None], negs], axis=1).astype(np.int64))
        model._loss_values = defaultdict(float)
        model.train()
        model.training = True
        model.zero_grad()
        torch.manual_seed(123)          # the dropout draw of this step (base_model.py:82)
        loss = model.get_loss(batch)
        loss.backward()
        out[f'{name}_batch'] = batch.numpy()
        out[f'{name}_p'] = np.float64(p)
        out[f'{name}_loss'] = np.float64(loss.item())
        out[f'{name}_bpr'] = np.float64(float(model._loss_values['bpr']))
        out[f'{name}_reg'] = np.float64(float(model._loss_values['reg']))
        out[f'{name}_grad_user'] = model.embedding_user.weight.grad.numpy().copy()
        out[f'{name}_grad_item'] = model.embedding_item.weight.grad.numpy().copy()
        out[f'{name}_reg_lambda'] = np.float64(args.reg_lambda)
    save('g8_loss.npz', **out)


# --------------------------------------------------------------------------- G9: metrics with duplicate test rows
def g9_metrics(work):
    """utils.calculate_metrics (utils.py:36-63) on hand-made lists: duplicate items inside y_true (duplicate test rows),
    lists of different lengths, users without a hit."""
    import pandas as pd
    from TextGCN.utils import calculate_metrics
    rng = np.random.default_rng(9)
    n, kmax, n_items = 40, 10, 60
    y_true, y_pred = [], []
    for u in range(n):
        t = rng.integers(0, n_items, rng.integers(1, 9)).tolist()
        if u % 3 == 0:
            t = t + t[:2]                      # duplicate test rows of this user
        y_true.append(t)
        y_pred.append(rng.permutation(n_items)[:kmax].tolist())
    ks = [1, 5, 10]
    df = pd.DataFrame.from_dict({'user_id': list(range(n)), 'y_true': y_true, 'y_pred': y_pred, 'scores': [[0.0] * kmax] * n})
    res = calculate_metrics(df, ['recall', 'precision', 'hit', 'ndcg', 'f1'], ks)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(t) for t in y_true], out=ptr[1:])
    save('g9_metrics.npz', true_ptr=ptr, true_items=np.concatenate(y_true).astype(np.int64), y_pred=np.asarray(y_pred, dtype=np.int64),
         ks=np.asarray(ks, dtype=np.int64), **{f'metric_{m}': np.asarray(v, dtype=np.float64) for m, v in res.items()})


# --------------------------------------------------------------------------- G10: --reshuffle
def g10_reshuffle(work, data):
    """BaseDataset with --reshuffle (dataset.py:40-44,63-87) on the synth-60x40 files: the re-split train / test rows."""
    import pandas as pd
    args = run_args(['--model', 'lgcn', '--no_train', '-k', '5', '--reshuffle', '--seed', '0'], data, work)
    ds = TextGCN.BaseDataset(args)
    folder = os.path.join(data, 'reshuffle_0')
    tr = pd.read_table(os.path.join(folder, 'train.tsv'), dtype=str)
    te = pd.read_table(os.path.join(folder, 'test.tsv'), dtype=str)
    src_tr = pd.read_table(os.path.join(data, 'train.tsv'), dtype=str)
    src_te = pd.read_table(os.path.join(data, 'test.tsv'), dtype=str)
    save('g10_reshuffle.npz', in_train_user=src_tr['user_id'].values.astype('U16'), in_train_item=src_tr['asin'].values.astype('U16'),
         in_test_user=src_te['user_id'].values.astype('U16'), in_test_item=src_te['asin'].values.astype('U16'),
         train_user=tr['user_id'].values.astype('U16'), train_item=tr['asin'].values.astype('U16'),
         test_user=te['user_id'].values.astype('U16'), test_item=te['asin'].values.astype('U16'),
         n_users=np.int64(ds.n_users), n_items=np.int64(ds.n_items))
    shutil.rmtree(folder, ignore_errors=True)


# --------------------------------------------------------------------------- G11: dynamic negative sampling loss
def g11_adv_loss(work, data):
    """AdvSamplModel.get_loss (advanced_sampling.py:46-69) on the synth-60x40 data, no dropout: per row a user and 20 distinct
    candidate items; the reference ranks them, drops the user's positives, keeps max(k) = 5 hard negatives, pairs them with up
    to 5 random positives (random.sample -- captured here, it cannot be reproduced elsewhere) and takes BaseModel's loss over
    the triples.  Stored: the batch, the captured positives, the negatives in the reference's order, the triples, loss + dE0."""
    import random
    from collections import defaultdict
    import TextGCN.advanced_sampling as adv
    args = run_args(['--model', 'adv_sampling', '--no_train', '-k', '3', '5', '--dropout', '0.0'], data, work)
    ds = TextGCN.AdvSamplDataset(args)
    model = TextGCN.AdvSamplModel(args, ds)
    set_weights(model, exact_embedding(ds.n_users, 64, 11), exact_embedding(ds.n_items, 64, 12))
    rng = np.random.default_rng(11)
    b, m = 24, 20
    users = rng.integers(0, ds.n_users, b)
    cand = np.stack([rng.choice(ds.n_items, size=m, replace=False) for _ in users])
    batch = torch.from_numpy(np.concatenate([users[:, None], cand], axis=1).astype(np.int64))
    positives, survivors, triples = [], [], []
    real_sample, real_sub, real_loss = random.sample, adv.subtract_tensor_as_set, TextGCN.BaseModel.get_loss

    def rec_sample(pop, k):
        out = real_sample(pop, k)
        positives.append(list(out))
        return out

    def rec_sub(t1, t2):
        out = real_sub(t1, t2)
        survivors.append(out.clone())
        return out

    def rec_loss(self, d):
        triples.append(d.clone())
        return real_loss(self, d)
    model._loss_values = defaultdict(float)
    model.train()
    model.training = True
    model.zero_grad()
    random.seed(1111)
    random.sample, adv.subtract_tensor_as_set, TextGCN.BaseModel.get_loss = rec_sample, rec_sub, rec_loss
    threads = torch.get_num_threads()
    torch.set_num_threads(1)        # the 560-row index backward adds duplicates in thread order: one thread = one order, so
    try:                            # that the fixture regenerates byte for byte (the test's bar is normwise 1e-4 anyway)
        loss = model.get_loss(batch)
        loss.backward()
    finally:
        random.sample, adv.subtract_tensor_as_set, TextGCN.BaseModel.get_loss = real_sample, real_sub, real_loss
        torch.set_num_threads(threads)
    kmax = max(args.k)
    pos = np.full((b, ds.pos_samples), -1, dtype=np.int64)
    neg = np.full((b, kmax), -1, dtype=np.int64)
    for r in range(b):
        pos[r, :len(positives[r])] = positives[r]
        keep = survivors[r][:kmax].numpy()
        neg[r, :len(keep)] = keep
    save('g11_adv_loss.npz', batch=batch.numpy(), positives=pos, negatives=neg, triples=triples[0].numpy(),
         loss=np.float64(loss.detach().item()), bpr=np.float64(float(model._loss_values['bpr'])), reg=np.float64(float(model._loss_values['reg'])),
         grad_user=model.embedding_user.weight.grad.numpy().copy(), grad_item=model.embedding_item.weight.grad.numpy().copy(),
         k=np.asarray(args.k), pos_samples=np.int64(ds.pos_samples), reg_lambda=np.float64(args.reg_lambda))


def main():
    work = tempfile.mkdtemp(prefix='tgcn_golden_')
    try:
        if ONLY is None:
            g1_dummy(work)
        data60 = g2_synth(work, write=ONLY is None)
        if ONLY is None:
            g3_dropout(work, data60)
        if ONLY in (None, 'g4', 'g7'):
            g4_ltr(work, data60)
        if ONLY in (None, 'g8'):
            g8_loss(work, data60)
        if ONLY in (None, 'g9'):
            g9_metrics(work)
        if ONLY in (None, 'g10'):
            g10_reshuffle(work, data60)
        if ONLY in (None, 'g11'):
            g11_adv_loss(work, data60)
        if ONLY is None:
            g5_medium(work)
            g6_builder(work)
    finally:
        os.chdir('/')
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()
