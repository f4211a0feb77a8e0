"""Host-side interaction data for the drop-in model: TSV -> internal ids -> normalised graph, plus the BPR
triple sampler `fit()` iterates over.

The attribute names are the ones the reference's models read from their dataset object (SURVEY.md §8b:
n_users, n_items, norm_matrix, train_user_dict, true_test_lil, test_df, user_mapping, item_mapping,
all_items ...; TextGCN/base_model.py:54-62), so `LightGCN(params, InteractionData(params))` and
`LightGCN(params, <a reference BaseDataset>)` are interchangeable.  Id assignment follows
TextGCN/dataset.py:45-54,89-98: rows sorted by (user_id, asin) as strings, ids in order of first appearance.
The Laplacian is textgcn_amd.graph.NormGraph (CSR) -- no dgl / scipy-DOK detour.
"""
import os

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset

from .graph import NormGraph, train_mask_csr


def _read_pairs(path):
    df = pd.read_table(path, dtype=str)
    if not {'user_id', 'asin'} <= set(df.columns):
        raise ValueError(f'{path}: expected tab-separated columns user_id, asin')
    return df.sort_values(by=['user_id', 'asin'], kind='stable').reset_index(drop=True)


def reshuffle_train_test(path, seed, train_size=0.8, logger=None):
    """`--reshuffle` (TextGCN/dataset.py:63-87): pool train.tsv + test.tsv, keep users with at least 3 rows, split 80/20
    stratified by user (sklearn train_test_split, random_state = seed), sort, drop test rows whose item never occurs in the new
    train split, and write both files to <path>/reshuffle_<seed>/ (reused by later runs, dataset.py:41-44).  Returns the
    folder."""
    from sklearn.model_selection import train_test_split
    folder = os.path.join(path, f'reshuffle_{seed}')
    if logger:
        logger.info('reshuffling train-test')
    os.makedirs(folder, exist_ok=True)
    df = pd.concat([pd.read_table(os.path.join(path, 'train.tsv'), dtype=str), pd.read_table(os.path.join(path, 'test.tsv'), dtype=str)])
    sizes = df.groupby('user_id').size()
    df = df[df['user_id'].isin(sizes[sizes >= 3].index)]
    train, test = train_test_split(df, stratify=df['user_id'], train_size=train_size, random_state=seed)
    train = train.sort_values(by=['user_id', 'asin']).reset_index(drop=True)
    test = test.sort_values(by=['user_id', 'asin']).reset_index(drop=True)
    test = test[test['asin'].isin(train['asin'].unique())]
    train.to_csv(os.path.join(folder, 'train.tsv'), sep='\t', index=False)
    test.to_csv(os.path.join(folder, 'test.tsv'), sep='\t', index=False)
    return folder


class InteractionData(Dataset):
    def __init__(self, params=None, folder=None, k=None, neg_samples=1, seed=0, logger=None, reshuffle=None):
        folder = folder if folder is not None else params.data
        self.path = folder
        self.logger = logger if logger is not None else getattr(params, 'logger', None)
        self.neg_samples = getattr(params, 'neg_samples', neg_samples)
        self.seed = getattr(params, 'seed', seed)
        k = k if k is not None else getattr(params, 'k', None)
        if reshuffle if reshuffle is not None else getattr(params, 'reshuffle', False):   # dataset.py:40-44
            sub = os.path.join(folder, f'reshuffle_{self.seed}')
            folder = sub if os.path.exists(os.path.join(sub, 'train.tsv')) else reshuffle_train_test(folder, self.seed, logger=self.logger)

        train = _read_pairs(os.path.join(folder, 'train.tsv'))
        test = _read_pairs(os.path.join(folder, 'test.tsv'))
        # ids by first appearance in the sorted train frame (dataset.py:89-98)
        u_codes, u_names = pd.factorize(train['user_id'])
        i_codes, i_names = pd.factorize(train['asin'])
        self.user_mapping = pd.DataFrame({'remap_id': np.arange(len(u_names)), 'org_id': np.asarray(u_names, dtype=str)})
        self.item_mapping = pd.DataFrame({'remap_id': np.arange(len(i_names)), 'org_id': np.asarray(i_names, dtype=str)})
        u_of = pd.Series(np.arange(len(u_names)), index=u_names)
        i_of = pd.Series(np.arange(len(i_names)), index=i_names)
        unknown_users = sorted(set(test['user_id']) - set(u_names))
        if unknown_users:   # dataset.py:56-57
            raise AssertionError(f"users {set(unknown_users)} from test set doesn't appear in train set")
        cold = ~test['asin'].isin(i_names)
        if cold.any():      # dataset.py:58-61: test items never seen in train are dropped
            if self.logger:
                self.logger.warning(f"items {set(test.loc[cold, 'asin'])} from test set don't appear in train set, removing them")
            test = test[~cold]
        self.train_df = pd.DataFrame({'user_id': u_codes.astype(np.int64), 'asin': i_codes.astype(np.int64)})
        self.test_df = pd.DataFrame({'user_id': u_of[test['user_id']].values.astype(np.int64),
                                     'asin': i_of[test['asin']].values.astype(np.int64)})

        self.n_users, self.n_items = len(u_names), len(i_names)
        self.n_train, self.n_test = len(self.train_df), len(self.test_df)
        if k and not self.n_items > max(k):   # dataset.py:25
            raise AssertionError(f'all k must be less than number of items ({self.n_items}), got k={k}')
        self.all_items = range(self.n_items)
        self.train_user_dict = self.train_df.groupby('user_id')['asin'].aggregate(list)            # dataset.py:111
        self.true_test_lil = self.test_df.groupby('user_id')['asin'].aggregate(list).values.tolist()  # dataset.py:120
        self.bucket_len = self.n_train // self.n_users       # samples per user and epoch, dataset.py:105
        self.iterable_len = self.bucket_len * self.n_users

        tu, ti = self.train_df.user_id.values, self.train_df.asin.values
        self.graph = NormGraph.from_pairs(tu, ti, self.n_users, self.n_items)
        self.mask_rowptr, self.mask_items = train_mask_csr(tu, ti, self.n_users)
        self._norm_matrix = None
        self._rng = np.random.default_rng(self.seed)
        self._epoch = None

    @property
    def norm_matrix(self):
        """The reference's container (coalesced torch sparse COO fp32, dataset.py:138), built on demand."""
        if self._norm_matrix is None:
            idx, val = self.graph.to_coo()
            self._norm_matrix = torch.sparse_coo_tensor(torch.from_numpy(idx), torch.from_numpy(val.copy()),
                                                        (self.graph.n, self.graph.n)).coalesce()
        return self._norm_matrix

    # ------------------------------------------------------------------ BPR triples (dataset.py:167-193)
    # Same contract as the reference sampler -- per epoch every user contributes bucket_len rows
    # [user, positive, neg_1..neg_m], positives drawn with replacement from the user's train items, negatives
    # distinct non-positives -- but drawn for all users at once with numpy instead of a python loop per user.
    def _draw_epoch(self):
        b, m = self.bucket_len, self.neg_samples
        rp, items = self.mask_rowptr, self.mask_items
        deg = np.diff(rp)
        if np.any(self.n_items - deg < b * m):
            bad = int(np.argmax(self.n_items - deg < b * m))
            # the reference spins forever in this situation
            raise ValueError(f'user {bad} has fewer than {b * m} non-positive items; cannot sample negatives')
        users = np.repeat(np.arange(self.n_users), b)
        pos = items[rp[users] + (self._rng.random(len(users)) * deg[users]).astype(np.int64)]
        negs = np.empty((len(users), m), dtype=np.int64)
        key_pos = np.sort(self.train_df.user_id.values * np.int64(self.n_items) + self.train_df.asin.values)
        # a user's b*m negatives must be distinct: draw row-wise, redraw collisions
        block = self._rng.integers(0, self.n_items, size=(self.n_users, b * m))
        for _ in range(1000):
            k = np.arange(self.n_users)[:, None] * np.int64(self.n_items) + block
            p = np.searchsorted(key_pos, k)
            is_pos = key_pos[np.minimum(p, len(key_pos) - 1)] == k
            srt = np.sort(block, axis=1)
            order = np.argsort(block, axis=1, kind='stable')
            dup_sorted = np.concatenate([np.zeros((self.n_users, 1), bool), srt[:, 1:] == srt[:, :-1]], axis=1)
            dup = np.zeros_like(dup_sorted)
            np.put_along_axis(dup, order, dup_sorted, axis=1)
            bad = is_pos | dup
            if not bad.any():
                break
            block[bad] = self._rng.integers(0, self.n_items, size=int(bad.sum()))
        else:
            raise RuntimeError('negative sampling did not converge')
        negs[:] = block.reshape(self.n_users * b, m)
        self._epoch = np.concatenate([users[:, None], pos[:, None].astype(np.int64), negs], axis=1)
        self._served = 0

    def __len__(self):
        return self.iterable_len

    def __getitem__(self, idx):
        if self._epoch is None or self._served >= self.iterable_len:
            self._draw_epoch()
        self._served += 1
        return torch.from_numpy(self._epoch[idx])

    def batches(self, batch_size, shuffle=True):
        """What `fit()` iterates over (main.py:35 builds `DataLoader(dataset, batch_size, shuffle=True)` and hands it to
        `model.fit`, main.py:40) without the per-sample Python round trip of a DataLoader over `__getitem__`: a re-iterable
        whose every pass draws a fresh epoch of triples (same sampler contract) and yields [batch, 2 + neg_samples] int64
        tensors cut from it in one piece.  On config 2 a DataLoader batch costs ~10 ms of host time against a 1.4 ms
        training step."""
        return _EpochBatches(self, int(batch_size), bool(shuffle))


class _EpochBatches:
    def __init__(self, data, batch_size, shuffle):
        self.data, self.batch_size, self.shuffle = data, batch_size, shuffle

    def __len__(self):
        return -(-self.data.iterable_len // self.batch_size)

    def __iter__(self):
        d = self.data
        d._draw_epoch()
        rows = d._epoch
        d._served = d.iterable_len          # the epoch is consumed here, not through __getitem__
        if self.shuffle:
            rows = rows[d._rng.permutation(len(rows))]
        table = torch.from_numpy(np.ascontiguousarray(rows))
        for j in range(0, len(table), self.batch_size):
            yield table[j:j + self.batch_size]
