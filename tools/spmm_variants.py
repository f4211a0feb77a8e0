#!/usr/bin/env python3
"""Dev tool: time every SpMM kernel variant on a synthetic config (interleaved rounds, one process)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import synth  # noqa: E402
from textgcn_amd.graph import NormGraph  # noqa: E402
from textgcn_amd.propagate import Propagator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2')
    ap.add_argument('--zipf', type=float, default=0.8)
    ap.add_argument('--rounds', type=int, default=10)
    ap.add_argument('--thresholds', type=int, nargs='*', default=[0, 512, 2048, 8192])
    ap.add_argument('--variants', type=str, nargs='*', default=['1:4', '1:8', '1:16'])
    args = ap.parse_args()
    n_u, n_i, nnz, d, K = synth.CONFIGS[args.config]
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0, zipf=args.zipf)
    gr = NormGraph.from_pairs(u, i, n_u, n_i)
    print(f'graph {args.config}: N={gr.n} nnz(A)={gr.nnz} max_deg={gr.degrees().max()} build {time.time() - t0:.1f}s', flush=True)
    dev = torch.device('cuda:0')
    e0 = synth.embeddings(gr.n, d).to(dev)
    results = []
    props = {t: Propagator(gr, dev, split_threshold=(t or None)) for t in args.thresholds}
    cases = [(t, v) for t in args.thresholds for v in args.variants]
    times = {c: [] for c in cases}
    for c in cases:  # warm-up
        t, v = c
        var, unr = map(int, v.split(':'))
        props[t].forward(e0, K, exact=(t == 0), variant=var, unroll=unr)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for c in cases:
            t, v = c
            var, unr = map(int, v.split(':'))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            props[t].forward(e0, K, exact=(t == 0), variant=var, unroll=unr)
            b.record()
            b.synchronize()
            times[c].append(a.elapsed_time(b))
    for c in cases:
        ms = float(np.median(times[c]))
        results.append({'threshold': c[0], 'variant': c[1], 'ms_fwd': ms, 'ms_min': float(np.min(times[c])),
                        'gedges_per_s': K * gr.nnz / ms / 1e6})
        print(f'T={c[0]:6d} variant={c[1]:5s} fwd {ms:8.3f} ms (min {np.min(times[c]):8.3f})  {K * gr.nnz / ms / 1e6:8.2f} Gedge/s', flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    with open(f'gpurun_out/spmm_variants_{args.config}.json', 'w') as f:
        json.dump(results, f, indent=1)


if __name__ == '__main__':
    main()
