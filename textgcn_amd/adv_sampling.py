"""Dynamic negative sampling on the HIP path.

Mirrors TextGCN/advanced_sampling.py: each training row carries a user and `max_neg_samples` random candidate
items; the model ranks the candidates with the current embeddings, keeps the `max(k)` best non-positives as hard
negatives and pairs them with up to `pos_samples` random positives of the user; the BPR loss of LightGCN is then
taken over all (user, positive, negative) triples.  The reference does this with a batched matmul and a Python loop
per user (advanced_sampling.py:55-69); here the candidate scores and the positives filter are one kernel
(tgcn_score_candidates_f32), the selection is tgcn_topk_f32, the positives are drawn on the device from the device mask CSR and
the pairing is tensor arithmetic: get_loss never waits for the GPU.
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

    def sample_positives(self, users):
        """Up to pos_samples DISTINCT random train items per user (advanced_sampling.py:62-63, `random.sample`): int64
        [B, pos_samples] on the device, -1 padded.  Drawn on the device from the device mask CSR with torch's device generator:
        draw j picks uniformly among the cnt - j items not yet taken (an index into the remaining set, shifted past the
        earlier picks in ascending order) -- a uniform sample without replacement, no host round trip, no per-user loop."""
        rp, it = self._mask_full()
        u = torch.as_tensor(users).to(self.device, torch.int64)
        start = rp[u].to(torch.int64)
        cnt = rp[u + 1].to(torch.int64) - start
        p = self.pos_samples
        rnd = torch.rand((u.numel(), p), device=self.device, dtype=torch.float64)
        picks = []
        for j in range(p):
            span = (cnt - j).clamp(min=1)
            x = torch.minimum((rnd[:, j] * span).to(torch.int64), span - 1)
            if picks:
                prev = torch.sort(torch.stack(picks, dim=1), dim=1)[0]
                for c in range(j):
                    x = x + (x >= prev[:, c]).to(torch.int64)
            picks.append(x)
        idx = torch.stack(picks, dim=1)
        valid = torch.arange(p, device=self.device)[None, :] < cnt[:, None]
        flat = (start[:, None] + idx).clamp(max=max(it.numel() - 1, 0))
        return torch.where(valid, it[flat].to(torch.int64), torch.full_like(idx, -1))

    def get_loss(self, data):
        """data: [B, 1 + m] rows (user, candidates...) -> BPR + L2 over positives x hard negatives (advanced_sampling.py:46-69).
        Nothing in here waits for the GPU: the candidate ranking, the positives and the pairing are device work, and the ragged
        per-user triple lists stay one dense [B x P x N] block whose empty places are padding rows the loss kernels skip
        (tgcn_bpr_pairs_f32: users[r] < 0), the means running over a device-side count of the real triples."""
        data = torch.as_tensor(data)
        if data.dim() != 2 or data.shape[1] < 2 or data.dtype.is_floating_point or data.dtype == torch.bool:
            raise TypeError('get_loss expects integer rows [user, candidate items...]')
        if data.device.type == 'cpu' and data.numel():     # what a DataLoader yields: range-checked before any kernel reads an id
            if int(data[:, 0].min()) < 0 or int(data[:, 0].max()) >= self.n_users or int(data[:, 1:].min()) < 0 \
                    or int(data[:, 1:].max()) >= self.n_items:
                raise IndexError('get_loss: a user or candidate id of the batch is outside its embedding table')
        data = data.to(self.device, torch.int64)
        users, cand = data[:, 0].contiguous(), data[:, 1:].contiguous()
        neg = self.hard_negatives(users, cand)                                   # [B, kmax], -1 padded
        pos = torch.as_tensor(self.sample_positives(users)).to(self.device, torch.int64)     # [B, P], -1 padded
        b, p, n = users.numel(), pos.shape[1], neg.shape[1]
        uu = users[:, None, None].expand(b, p, n)
        pp = pos[:, :, None].expand(b, p, n)
        nn_ = neg[:, None, :].expand(b, p, n)
        ok = (pp >= 0) & (nn_ >= 0)
        if self._native_loss():
            from collections import defaultdict
            from .model import _BprStep
            if not hasattr(self, '_loss_values'):
                self._loss_values = defaultdict(float)
            zero = torch.zeros((), dtype=torch.int64, device=self.device)
            cols = torch.stack([torch.where(ok, uu, zero - 1).reshape(-1), torch.where(ok, pp, zero).reshape(-1),
                                torch.where(ok, nn_, zero).reshape(-1)])           # [3, B P N]: padding rows carry user -1
            bpr, reg = _BprStep.apply(self.embedding_user.weight, self.embedding_item.weight, self, cols.contiguous(), ok.sum())
            self._loss_values['bpr'] += bpr.detach()
            self._loss_values['reg'] += reg.detach()
            return bpr + reg
        triples = torch.stack([uu[ok], pp[ok], nn_[ok]], dim=1)                # cartesian product per user (generic path: syncs)
        return super().get_loss(triples)
