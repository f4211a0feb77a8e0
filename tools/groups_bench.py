#!/usr/bin/env python3
"""Dev tool (round 4): row groups (tgcn_spmm_groups_f32) against one wave per row on the BASELINE configs, interleaved rounds in
one process; the two forms are compared bit for bit on the way.

    python tools/groups_bench.py --configs c3 c2 c4 [--rounds 8] [--unrolls 0 32]
"""
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
from textgcn_amd import propagate  # noqa: E402
from textgcn_amd.propagate import Propagator  # noqa: E402


def run_cases(args, cfg, gr, prop, e0, K, target, single):
    cases = [('groups', True, un, seg) for un in args.unrolls for seg in (None, False)] + \
            [('wave_per_row', False, 0, seg) for seg in (None, False)]
    outs = {}
    times = {c: [] for c in cases}
    for c in cases:
        prop.csr.use_groups = c[1]
        outs[c] = prop.forward(e0, K, unroll=c[2], segmented=c[3]).clone()
    torch.cuda.synchronize()
    ref = outs[cases[-1]]
    for r in range(args.rounds):
        for c in cases:
            prop.csr.use_groups = c[1]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            prop.forward(e0, K, unroll=c[2], segmented=c[3])
            b.record()
            b.synchronize()
            times[c].append(a.elapsed_time(b))
    with open(args.out, 'a') as f:
        for c in cases:
            ms = float(np.median(times[c]))
            same_as = ('wave_per_row', False, 0, c[3])
            rec = {'lib': os.path.basename(args.lib) if args.lib else 'libtgcn.so', 'config': cfg, 'target_entries': target, 'single_len': single, 'form': c[0], 'unroll': c[2], 'segmented': 'auto' if c[3] is None else 'off',
                   'seg_blocks': prop.csr.segment_blocks, 'ms_fwd': round(ms, 4), 'us_layer': round(ms / K * 1e3, 1),
                   'ms_min': round(float(np.min(times[c])), 4), 'gedges_per_s': round(K * gr.nnz / ms / 1e6, 2),
                   'bits_equal_to_wave_per_row': bool(torch.equal(outs[c].view(torch.int32), outs[same_as].view(torch.int32)))}
            print(json.dumps(rec), flush=True)
            f.write(json.dumps(rec) + '\n')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', nargs='*', default=['c3', 'c2', 'c4'])
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--unrolls', type=int, nargs='*', default=[0])
    ap.add_argument('--out', default='gpurun_out/groups_bench.jsonl')
    ap.add_argument('--targets', type=int, nargs='*', default=[64], help='entries per group (sweep)')
    ap.add_argument('--singles', type=int, nargs='*', default=[32], help='rows of this length or more stay alone (sweep)')
    ap.add_argument('--lib', default=None, help='another build of libtgcn.so (A/B on one box: run the tool once per library)')
    args = ap.parse_args()
    if args.lib:
        from textgcn_amd import _capi
        _capi.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device('cuda:0')
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for cfg in args.configs:
        n_u, n_i, nnz, d, K = synth.CONFIGS[cfg]
        t0 = time.time()
        u, i = synth.interactions(n_u, n_i, nnz, seed=0)
        gr = NormGraph.from_pairs(u, i, n_u, n_i)
        del u, i
        print(f'{cfg}: N={gr.n} nnz(A)={gr.nnz} build {time.time() - t0:.1f}s', flush=True)
        e0 = synth.embeddings(gr.n, d).to(dev)
        for target in args.targets:
            for single in args.singles:
                propagate.GROUP_TARGET_ENTRIES, propagate.GROUP_SINGLE_LEN = target, single
                prop = Propagator(gr, dev)
                run_cases(args, cfg, gr, prop, e0, K, target, single)
                del prop
                torch.cuda.empty_cache()
        del e0, gr
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
