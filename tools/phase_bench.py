#!/usr/bin/env python3
"""Per-phase timing of one SpMM layer on a BASELINE config: user rows only (gather from the item table), item rows
only (gather from the user table), both -- to see which half of the launch is bound by what.

    python tools/phase_bench.py --workload c2 [--unroll 16]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2')
    ap.add_argument('--unroll', type=int, nargs='+', default=[0])
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--segment-items', type=int, default=0, help='also time the item rows alone through the segmented kernel with this many column blocks')
    ap.add_argument('--tile', type=int, default=1024)
    args = ap.parse_args()
    from textgcn_amd import propagate, synth
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, K = synth.CONFIGS[args.workload]
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    dev = torch.device('cuda:0')
    x = synth.embeddings(g.n, d, seed=0).to(dev)
    parts = {'users': (0, n_u), 'items': (n_u, g.n), 'all': (0, g.n)}
    if args.segment_items:
        parts = {'items': (n_u, g.n), 'items-segmented': (n_u, g.n)}
    for name, (r0, r1) in parts.items():
        rp, ci, va = g.row_block(r0, r1)
        if name == 'items-segmented':
            csr = propagate.DeviceCSR(rp, ci, va, g.n, dev, split_threshold=propagate.DEFAULT_SPLIT_THRESHOLD,
                                      block_specs=[(0, r1 - r0, 0, n_u)], segment=[args.segment_items])
            csr.segment_tile = args.tile
        else:
            csr = propagate.DeviceCSR(rp, ci, va, g.n, dev, split_threshold=propagate.DEFAULT_SPLIT_THRESHOLD)
        y = torch.empty((r1 - r0, d), device=dev)
        acc = torch.empty((r1 - r0, d), device=dev)
        e0 = x[r0:r1].contiguous()
        for un in args.unroll:
            for mode in ('y', 'y+acc'):
                def fn():
                    if mode == 'y':
                        propagate.spmm(csr, x, y=y, unroll=un)
                    else:
                        propagate.spmm(csr, x, y=y, acc_in=e0, acc_out=acc, unroll=un)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.steps):
                    fn()
                b.record()
                torch.cuda.synchronize()
                us = a.elapsed_time(b) / args.steps * 1e3
                print(json.dumps({'rows': name, 'n_rows': r1 - r0, 'entries': csr.nnz, 'unroll': un, 'epilogue': mode,
                                  'us': round(us, 1), 'gather_TBs': round(csr.nnz * d * 4 / us / 1e6, 2)}), flush=True)


if __name__ == '__main__':
    main()
