"""Scored pairs/s of the two exact top-k entry points on the same inputs: tgcn_score_topk_f32 (fp32 MFMA filter) and
tgcn_score_topk_prefilter_f32 (bf16 candidates, fp32 rescoring), one stream and three streams, results compared bit for bit.

    python tools/score_mode_bench.py [--shapes c2,c2big,c3,c4] [--reps 20]
One JSON line per (shape, mode, streams)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import scoring  # noqa: E402

SHAPES = {   # users per call, calls, items, d
    'c2': (2048, 12, 50_000, 64),
    'c2x20': (2048, 20, 50_000, 64),
    'c2big': (16384, 6, 50_000, 64),
    'c3': (16384, 6, 60_000, 128),
    'c3small': (2048, 12, 60_000, 128),
    'c4': (2048, 6, 2_000_000, 64),
    'c5': (8192, 4, 60_000, 960),          # the folded ltr_linear operands (K = 128 + 2 x 384 + bias column)
    'c5small': (2048, 8, 60_000, 960),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='c2,c2big,c3,c4')
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--k', type=int, default=40)
    ap.add_argument('--streams', default='1,3')
    ap.add_argument('--users', type=int, default=0, help='users per call instead of the shape\'s (same number of calls)')
    ap.add_argument('--calls', type=int, default=0)
    ap.add_argument('--modes', default='fp32,prefilter,prefilter+pack')
    ap.add_argument('--lib', default=None, help='another build of libtgcn.so to time (A/B on one box: run the tool once per library)')
    args = ap.parse_args()
    if args.lib:
        from textgcn_amd import _capi
        _capi.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device('cuda:0')
    for name in args.shapes.split(','):
        b, calls, n_items, d = SHAPES[name]
        b = args.users or b
        calls = args.calls or calls
        g = torch.Generator().manual_seed(0)
        n_users = b * calls
        ue = (torch.randn(n_users, d, generator=g) * 0.1).to(dev)
        ie = (torch.randn(n_items, d, generator=g) * 0.1).to(dev)
        rng = np.random.default_rng(0)
        per = 50
        mi = np.sort(rng.integers(0, n_items, size=(n_users, per)), axis=1).astype(np.int32)
        batches = []
        for c in range(calls):
            ids = torch.arange(c * b, (c + 1) * b, dtype=torch.int64, device=dev)
            rows = [np.unique(r) for r in mi[c * b:(c + 1) * b]]
            rp = np.zeros(b + 1, dtype=np.int32)
            np.cumsum([len(r) for r in rows], out=rp[1:])
            batches.append((ids, torch.from_numpy(rp).to(dev), torch.from_numpy(np.concatenate(rows)).to(dev)))
        pack = scoring.item_pack(ie)
        modes = {'fp32': dict(prefilter=False), 'prefilter': dict(prefilter=True, item_pack=None),
                 'prefilter+pack': dict(prefilter=True, item_pack=pack)}
        ref = None
        for mode, kw in modes.items():
            if mode not in args.modes.split(',') and mode != 'fp32':
                continue
            for n_streams in [int(x) for x in args.streams.split(',')]:
                main_s = torch.cuda.current_stream(dev)
                side = [torch.cuda.Stream(dev) for _ in range(n_streams)]

                def run():
                    for st in side:
                        st.wait_stream(main_s)
                    keep = []
                    for j, (ids, rp, it) in enumerate(batches):
                        with torch.cuda.stream(side[j % n_streams]):
                            keep.append(scoring.score_topk(ue, ie, args.k, user_ids=ids, mask_rowptr=rp, mask_items=it, round4=True,
                                                           slot=j % n_streams, **kw))
                    for st in side:
                        main_s.wait_stream(st)
                    return keep
                keep = run()
                torch.cuda.synchronize()
                if ref is None:
                    ref = [(v.clone(), i.clone()) for v, i in keep]
                fb = scoring.fallback_count(dev, b, n_items, d, args.k, slot=0)
                same = all(torch.equal(v, rv) and torch.equal(i, ri) for (v, i), (rv, ri) in zip(keep, ref))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 1e3 / args.reps
                print(json.dumps({'lib': os.path.basename(args.lib) if args.lib else 'libtgcn.so', 'shape': name, 'users_per_call': b, 'calls': calls, 'items': n_items, 'd': d, 'mode': mode,
                                  'streams': n_streams, 'us_per_call': round(t / calls * 1e6, 1),
                                  'T_pairs_per_s': round(n_users * n_items / t / 1e12, 4), 'identical_to_fp32': bool(same), 'fallback_users_last_call': fb}), flush=True)


if __name__ == '__main__':
    main()
