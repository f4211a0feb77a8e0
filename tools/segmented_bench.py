#!/usr/bin/env python3
"""Timing of the XCD-affine segmented SpMM (tgcn_spmm_segmented_f32) against the plain kernel on a BASELINE config.

    python tools/segmented_bench.py --workload c2 --configs 0,8 8,8 0,16 --tile 256
Each --configs entry is users_blocks,items_blocks (column blocks for user rows / item rows; 0 = rows stay direct); a block
count may carry its class count as blocks:classes (4:4 = four blocks, two XCDs per block)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2')
    ap.add_argument('--configs', nargs='+', default=['0,8', '8,8', '0,16'])
    ap.add_argument('--tile', type=int, nargs='+', default=[256])
    ap.add_argument('--unroll', type=int, nargs='+', default=[0])
    ap.add_argument('--min-row-len', type=int, nargs='+', default=[0])
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--two-phase', action='store_true')
    args = ap.parse_args()
    from textgcn_amd import propagate, synth
    propagate.SEGMENT_TWO_PHASE = args.two_phase
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, K = synth.CONFIGS[args.workload]
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    graph = NormGraph.from_pairs(u, i, n_u, n_i)
    dev = torch.device('cuda:0')
    e0 = synth.embeddings(graph.n, d, seed=0).to(dev)
    prop = propagate.Propagator(graph, dev)
    out = torch.empty_like(e0)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / args.steps

    base = timed(lambda: prop.forward(e0, K, out=out, segmented=False))
    ref = out.clone()
    print(json.dumps({'variant': 'plain', 'ms_per_forward': round(base, 4)}), flush=True)
    for cfg in args.configs:
        bu, bi = (tuple(int(x) for x in t.split(':')) if ':' in t else int(t) for t in cfg.split(','))
        for ml, mr in [(x, y) for x in args.tile for y in args.min_row_len]:
            prop.csr.configure_segments([bu, bi], tile_entries=ml, min_row_len=mr)
            for un in args.unroll:
                ms = timed(lambda: prop.forward(e0, K, out=out, segmented=True, unroll=un))
                err = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
                h = prop.csr._segment_plans[d][0][3]
                print(json.dumps({'variant': f'segmented users={bu} items={bi}', 'tile': ml, 'min_row_len': mr, 'unroll': un,
                                  'ms_per_forward': round(ms, 4), 'vs_plain': round(base / ms, 3), 'normwise_vs_plain': err,
                                  'tiles': len(h['tile_meta']), 'slots': h['n_slots'],
                                  'direct_rows': len(h['direct_rows'])}), flush=True)


if __name__ == '__main__':
    main()
