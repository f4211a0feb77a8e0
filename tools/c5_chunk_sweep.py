#!/usr/bin/env python3
"""Config 5 (LTRLinear.predict_tensors over every user, default candidate path; --model lgcn: config 3): users per scoring call x streams.

  python tools/c5_chunk_sweep.py [--chunks 4096,8192,16384,32768] [--streams 2,4] [--reps 3]

One JSON line per (chunk, streams): ms for the whole catalogue less the propagation, and whether the lists equal the first arm's.
A sweep tool: nothing here is the product path's configuration (that is ltr.LTRLinear.ltr_predict_chunk / predict_streams)."""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', default='4096,8192,16384,32768')
    ap.add_argument('--streams', default='2,4')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--model', default='ltr', choices=['ltr', 'lgcn'], help='lgcn: config 3 (LightGCN.predict_tensors, d = 128) instead')
    args = ap.parse_args()
    import bench
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.ltr import LTRLinear
    dev = torch.device('cuda:0')
    n_u, n_i, nnz, d, K = synth.CONFIGS['c3']
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    gen = torch.Generator().manual_seed(5)
    t = 384
    text = {'items_as_desc': torch.randn((n_i, t), generator=gen), 'items_as_avg_reviews': torch.randn((n_i, t), generator=gen),
            'users_as_avg_reviews': torch.randn((n_u, t), generator=gen), 'users_as_avg_desc': torch.randn((n_u, t), generator=gen)}
    p5 = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device=dev, load=None, load_base=None, freeze=True,
                               batch_size=2048, quiet=True, ltr_layers=[])
    if args.model == 'lgcn':
        from textgcn_amd.model import LightGCN
        ltr = LightGCN(p5, bench._model_dataset(u, i, n_u, n_i, g))
    else:
        ltr = LTRLinear(p5, bench._model_dataset(u, i, n_u, n_i, g, text))
    e0 = synth.embeddings(g.n, d, seed=0)
    with torch.no_grad():
        ltr.embedding_user.weight.copy_(e0[:n_u])
        ltr.embedding_item.weight.copy_(e0[n_u:])
    users = np.arange(n_u)

    def fwd():
        with torch.no_grad():
            return ltr.representation
    t_fwd = timed(fwd, 5)
    first = None
    for ns in (int(x) for x in args.streams.split(',')):
        ltr.predict_streams = ns
        ltr._streams = None
        for chunk in (int(x) for x in args.chunks.split(',')):
            ltr.ltr_predict_chunk = ltr.predict_chunk = chunk
            v, ix = ltr.predict_tensors(users)
            if first is None:
                first = (v.clone(), ix.clone())
            same = bool(torch.equal(v, first[0]) and torch.equal(ix, first[1]))
            del v, ix
            tt = timed(lambda: ltr.predict_tensors(users), args.reps) - t_fwd
            print(json.dumps({'users_per_call': chunk, 'streams': ns, 'ms_total': round(tt * 1e3, 2), 'calls': -(-n_u // chunk),
                              'pairs_per_s': round(n_u * n_i / tt / 1e9, 1), 'unit': 'G pairs/s', 'same_lists_as_first_arm': same}),
                  flush=True)


if __name__ == '__main__':
    main()
