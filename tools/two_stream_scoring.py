#!/usr/bin/env python3
"""Does issuing consecutive scoring calls on two HIP streams (own workspace each) hide the small kernels?"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import scoring, synth  # noqa: E402
from textgcn_amd.graph import NormGraph, train_mask_csr  # noqa: E402
from textgcn_amd.propagate import Propagator  # noqa: E402

n_u, n_i, nnz, d, K = synth.CONFIGS['c2']
u, i = synth.interactions(n_u, n_i, nnz, seed=0)
g = NormGraph.from_pairs(u, i, n_u, n_i)
dev = torch.device('cuda:0')
out = Propagator(g, dev).forward(synth.embeddings(g.n, d).to(dev), K)
ue, ie = out[:n_u].contiguous(), out[n_u:].contiguous()
mrp, mit = train_mask_csr(u, i, n_u)
res = {}
for B in (2048, 16384):
    batches = []
    for b in range(min(24, n_u // B)):
        users = np.arange(b * B, (b + 1) * B)
        rp = np.zeros(B + 1, dtype=np.int32)
        np.cumsum(mrp[users + 1] - mrp[users], out=rp[1:])
        batches.append((torch.from_numpy(users).to(dev), torch.from_numpy(rp).to(dev),
                        torch.from_numpy(np.ascontiguousarray(mit[mrp[users[0]]:mrp[users[-1] + 1]])).to(dev)))
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    for n_streams in (1, 2, 3):
        def run():
            for j, (ids, rp, it) in enumerate(batches):
                s = streams[j % n_streams]
                with torch.cuda.stream(s):
                    scoring.score_topk(ue, ie, 40, user_ids=ids, mask_rowptr=rp, mask_items=it, round4=True, slot=j % n_streams)
        torch.cuda.synchronize()
        run()
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        run()
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        b_.record()
        b_.synchronize()
        t = a.elapsed_time(b_) / 1e3
        res[f'B{B}_streams{n_streams}_Gpairs_per_s'] = round(len(batches) * B * n_i / t / 1e9, 1)
print(json.dumps(res))
