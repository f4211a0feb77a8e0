#!/usr/bin/env python3
"""Per-rank compute of the feature (column) partition on ONE GPU: every rank of a P-way split runs the whole graph on d / P
columns with no per-layer exchange, so its forward time can be measured here -- only the final all-gather needs the other GPUs.

    python tools/column_split_bench.py --workload c4 --parts 1 2 4 8
Prints ms per K-layer forward on d / P columns, the gather bytes per layer and their rate."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c4')
    ap.add_argument('--parts', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    from textgcn_amd import synth
    from textgcn_amd.dist import ColumnShardedPropagator
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, K = synth.CONFIGS[args.workload]
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(g.n, d, seed=0)
    print(json.dumps({'workload': args.workload, 'graph_build_s': round(time.time() - t0, 1)}), flush=True)
    dev = torch.device('cuda:0')
    for p in args.parts:
        cp = ColumnShardedPropagator(g, d, 0, p, dev)
        x = cp.local_e0(e0)
        for _ in range(2):
            cp.forward(x, K)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            cp.forward(x, K)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / args.steps
        gather = g.nnz * 4.0 * cp.dl
        print(json.dumps({'ranks': p, 'columns_per_rank': cp.dl, 'row_bytes': 4 * cp.dl, 'ms_per_forward_per_rank': round(ms, 3),
                          'ms_per_layer': round(ms / K, 3), 'gather_GB_per_layer': round(gather / 1e9, 2),
                          'gather_TBs': round(gather / (ms / K * 1e-3) / 1e12, 2),
                          'final_all_gather_bytes_received_per_rank': int((p - 1) * g.n * cp.dl * 4)}), flush=True)
        del cp, x
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
