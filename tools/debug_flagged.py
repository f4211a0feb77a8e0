#!/usr/bin/env python3
"""Dev tool: how many users of a fused-scoring call take the exact fallback, and why (c2, first B users)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import _capi, scoring, synth  # noqa: E402
from textgcn_amd.graph import NormGraph, train_mask_csr  # noqa: E402
from textgcn_amd.propagate import Propagator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n_u, n_i, nnz, d, K = synth.CONFIGS['c2']
u, i = synth.interactions(n_u, n_i, nnz, seed=0)
g = NormGraph.from_pairs(u, i, n_u, n_i)
dev = torch.device('cuda:0')
out = Propagator(g, dev).forward(synth.embeddings(g.n, d).to(dev), K)
ue, ie = out[:n_u].contiguous(), out[n_u:].contiguous()
mrp, mit = train_mask_csr(u, i, n_u)
users = np.arange(B)
rowptr = np.zeros(B + 1, dtype=np.int32)
np.cumsum(mrp[users + 1] - mrp[users], out=rowptr[1:])
items = mit[mrp[0]:mrp[B]]
v, idx = scoring.score_topk(ue, ie, 40, user_ids=torch.from_numpy(users).to(dev), mask_rowptr=torch.from_numpy(rowptr).to(dev),
                            mask_items=torch.from_numpy(np.ascontiguousarray(items)).to(dev), round4=True)
torch.cuda.synchronize()
ws = scoring._WORKSPACE[(dev, 0)]
total = _capi.lib().tgcn_score_topk_workspace_bytes(B, n_i, d, 40)
off = total - (((B + 1) * 4 + 255) // 256) * 256
flagged = ws[off:off + 4 * (B + 1)].view(torch.int32).cpu().numpy()
n = int(flagged[0])
print('flagged', n, 'of', B, 'users:', flagged[1:1 + min(n, 20)])
s = scoring.score_dense(ue, ie, user_ids=torch.from_numpy(users[:256]).to(dev))
print('score stats: max', float(s.max()), 'mean', float(s.mean()), 'std', float(s.std()))
if n:
    fu = flagged[1:1 + n]
    sf = scoring.score_dense(ue, ie, user_ids=torch.from_numpy(fu.astype(np.int64)).to(dev))
    srt, _ = torch.sort(sf, dim=1, descending=True)
    print('flagged users: top scores', srt[:3, :5].cpu().numpy(), 'train items', (mrp[fu + 1] - mrp[fu])[:10])
    print('ties at rank 40..45:', srt[:3, 38:46].cpu().numpy())
