"""Dynamic negative sampling on the HIP path.

Mirrors TextGCN/advanced_sampling.py: each training row carries a user and `max_neg_samples` random candidate
items; the model ranks the candidates with the current embeddings, keeps the `max(k)` best non-positives as hard
negatives and pairs them with up to `pos_samples` random positives of the user; the BPR loss of LightGCN is then
taken over all (user, positive, negative) triples.  The reference does this with a batched matmul and a Python loop
per user (advanced_sampling.py:55-69); here the candidate scores and the positives filter are one kernel
(tgcn_score_candidates_f32), the selection is tgcn_topk_f32, and the pairing is tensor indexing.
"""
import numpy as np
import torch

from . import scoring
from .interactions import InteractionData
from .model import LightGCN


class AdvSamplData(InteractionData):
    pos_samples = 5            # advanced_sampling.py:12
    max_neg_samples = 1000     # advanced_sampling.py:13

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.n_candidates = min(self.n_items, self.max_neg_samples)

    def __getitem__(self, idx):
        """[user, c_1 .. c_m]: m distinct random items (advanced_sampling.py:21-22)."""
        user = idx // self.bucket_len
        cand = self._rng.choice(self.n_items, size=self.n_candidates, replace=False)
        return torch.from_numpy(np.concatenate([[user], cand]).astype(np.int64))


class AdvSamplModel(LightGCN):
    def _copy_dataset_params(self, dataset):
        super()._copy_dataset_params(dataset)
        self.pos_samples = getattr(dataset, 'pos_samples', 5)
        self._pos_rng = np.random.default_rng(getattr(dataset, 'seed', 0))

    @torch.no_grad()
    def hard_negatives(self, users, cand):
        """[B, max(k)] item ids: the best-scored candidates that are not train items of the user, best first
        (advanced_sampling.py:61-64).  A user with fewer than max(k) non-positive candidates gets -1 padding."""
        users_emb, items_emb = self.representation
        rp, it = self._mask_full()
        s = scoring.score_candidates(users_emb.contiguous(), users, items_emb.contiguous(), cand, rp, it)
        kk = min(max(self.k), cand.shape[1])
        val, pos = scoring.topk(s, kk)
        neg = torch.gather(cand, 1, scoring.retired_positions(pos))      # (NO_ITEM positions: value -inf, replaced by -1 below)
        return torch.where(torch.isneginf(val), torch.full_like(neg, -1), neg)

    def _mask_full(self):
        if getattr(self, '_mask_full_dev', None) is None:
            self._mask_full_dev = (torch.from_numpy(self._mask_rowptr_host.astype(np.int32)).to(self.device),
                                   torch.from_numpy(self._mask_items_host).to(self.device))
        return self._mask_full_dev

    def sample_positives(self, users_np):
        """up to pos_samples distinct random train items per user (advanced_sampling.py:62-63): [B, pos_samples], -1 padded"""
        rp, it = self._mask_rowptr_host, self._mask_items_host
        out = np.full((len(users_np), self.pos_samples), -1, dtype=np.int64)
        cnt = rp[users_np + 1] - rp[users_np]
        rows = np.repeat(np.arange(len(users_np)), cnt)
        flat = np.concatenate([it[rp[u]:rp[u + 1]] for u in users_np]) if len(users_np) else it[:0]
        order = np.lexsort((self._pos_rng.random(len(flat)), rows))       # random order inside each user
        rows, flat = rows[order], flat[order]
        start = np.concatenate([[0], np.cumsum(cnt)[:-1]])
        rank = np.arange(len(flat)) - start[rows]
        keep = rank < self.pos_samples
        out[rows[keep], rank[keep]] = flat[keep]
        return out

    def get_loss(self, data):
        """data: [B, 1 + m] rows (user, candidates...) -> BPR + L2 over positives x hard negatives (advanced_sampling.py:46-69)."""
        data = data.to(self.device)
        users, cand = data[:, 0].contiguous(), data[:, 1:].contiguous()
        neg = self.hard_negatives(users, cand)                                   # [B, kmax]
        pos = torch.from_numpy(self.sample_positives(users.cpu().numpy())).to(self.device)   # [B, P]
        b, p, n = users.numel(), pos.shape[1], neg.shape[1]
        uu = users[:, None, None].expand(b, p, n)
        pp = pos[:, :, None].expand(b, p, n)
        nn_ = neg[:, None, :].expand(b, p, n)
        ok = (pp >= 0) & (nn_ >= 0)
        triples = torch.stack([uu[ok], pp[ok], nn_[ok]], dim=1)                # cartesian product per user
        return super().get_loss(triples)
