#!/usr/bin/env python3
"""BASELINE configs 3 and 5 end to end on one MI355X (the bench.py line is config 2; these are the other
single-GPU configs, run through the MODEL CLASSES exactly as a user would):

  c3: U=180k, I=60k, nnz=1.6M, d=128, K=4 -- `LightGCN.representation` + full-catalogue `predict` for all users
  c5: ltr_linear on the frozen c3 embeddings + 4 text tables of width 384 -- `LTRLinear.predict` for all users

Prints one JSON object per config (device time by HIP events; predict's D->H list conversion reported separately).
"""
import json
import os
import sys
import time
import types

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import synth  # noqa: E402
from textgcn_amd.graph import NormGraph, train_mask_csr  # noqa: E402


def dataset(u, i, n_u, n_i, graph, text=None):
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(
        n_users=n_u, n_items=n_i, graph=graph, norm_matrix=None, mask_rowptr=rp, mask_items=items,
        true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
        user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u0']}), item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i0']}),
        all_items=range(n_i))
    if text:
        ds.__dict__.update(text)
    return ds


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps / 1e3


def main():
    from textgcn_amd import scoring
    from textgcn_amd.ltr import LTRLinear
    from textgcn_amd.model import LightGCN
    dev = torch.device('cuda:0')
    n_u, n_i, nnz, d, K = synth.CONFIGS['c3']
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    build_s = time.time() - t0
    p = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device='cuda:0', load=None, batch_size=2048, quiet=True)
    m = LightGCN(p, dataset(u, i, n_u, n_i, g))
    users = np.arange(n_u)

    def fwd():
        with torch.no_grad():
            return m.representation
    t_fwd = timed(fwd, reps=10)

    t_all = timed(lambda: m.predict_tensors(users), reps=2)    # representation + fused scoring of every user
    t1 = time.time()
    m.predict(users, with_scores=True)
    t_predict_wall = time.time() - t1
    pairs = n_u * n_i
    print(json.dumps({'config': 'c3', 'U': n_u, 'I': n_i, 'nnz_A': g.nnz, 'd': d, 'K': K, 'graph_build_s': round(build_s, 1),
                      'forward_ms': t_fwd * 1e3, 'propagated_edges_per_s': K * g.nnz / t_fwd,
                      'full_catalogue_scoring_ms': (t_all - t_fwd) * 1e3, 'scored_pairs_per_s': pairs / (t_all - t_fwd),
                      'mfma_frac': 2.0 * d * pairs / (t_all - t_fwd) / 157.3e12,
                      'predict_wall_s_incl_tolist': round(t_predict_wall, 2)}))

    # ---- c5
    gen = torch.Generator().manual_seed(5)
    t = 384
    text = {'items_as_desc': torch.randn((n_i, t), generator=gen), 'items_as_avg_reviews': torch.randn((n_i, t), generator=gen),
            'users_as_avg_reviews': torch.randn((n_u, t), generator=gen), 'users_as_avg_desc': torch.randn((n_u, t), generator=gen)}
    p5 = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device='cuda:0', load=None, load_base=None, freeze=True,
                               batch_size=2048, quiet=True, ltr_layers=[])
    ltr = LTRLinear(p5, dataset(u, i, n_u, n_i, g, text))
    ltr.predict_chunk = 2048     # K = 960: the [B, I] matrix is materialised (d > 256), keep the reference batch

    def ltr_all():
        ltr.predict_tensors(users)       # representation + fold/pack + K = 960 GEMM + mask + top-k for every user
    t_ltr = timed(ltr_all, reps=1) - t_fwd
    kf = d + 2 * t
    print(json.dumps({'config': 'c5', 'U': n_u, 'I': n_i, 'd': d, 'text_dim': t, 'folded_K': int(ltr._k()),
                      'full_catalogue_ltr_scoring_ms': t_ltr * 1e3, 'scored_pairs_per_s': pairs / t_ltr,
                      'algorithmic_flops': 2.0 * kf * pairs, 'mfma_frac': 2.0 * kf * pairs / t_ltr / 157.3e12}))


if __name__ == '__main__':
    main()
