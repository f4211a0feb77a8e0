#!/usr/bin/env python3
"""Randomised parity sweep of the fused scoring entry points (development tool; the fixed-seed subset lives in
tests/test_fuzz_gpu.py).  Every case draws a shape around the kernels' tile boundaries, a data style (gaussian, a few distinct
values = many tied scores, wide dynamic range, rows of zeros, a non-finite row), an optional train mask and user-id indirection,
and requires tgcn_score_topk_f32 and tgcn_score_topk_prefilter_f32 (pack built inside the call and handed in) to return the
bits of tgcn_score_dense_f32 -> tgcn_mask_f32 -> tgcn_topk_f32.

    python tools/fuzz_parity.py --seconds 120 [--seed0 0] [--wide] [--huge]      (one JSON line per failure, a summary line at the end)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EDGES = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4096, 4097,
         8191, 8192, 8193, 16383, 16384, 16385]
WIDTHS = [1, 3, 4, 6, 8, 15, 16, 17, 31, 32, 33, 48, 50, 63, 64, 65, 96, 100, 101, 127, 128, 129, 136, 192, 200, 256, 264, 384, 512, 520, 896,
          960, 1000, 1024]


def near(rng, hi, lo=1):
    """a size in [lo, hi]: half the time on or next to a tile boundary"""
    if rng.random() < 0.5:
        c = [e for e in EDGES if lo <= e <= hi]
        if c:
            return int(rng.choice(c))
    return int(rng.integers(lo, hi + 1))


def draw_case(seed, wide=False, huge=False):
    rng = np.random.default_rng(seed)
    d = int(rng.choice(WIDTHS if not wide else [w for w in WIDTHS if w > 128]))
    big = rng.random() < 0.25
    b = near(rng, 2600 if not big else 600)
    i = near(rng, 20000 if not big else 70000, lo=1)
    if d > 256:      # keep the dense reference small
        b, i = min(b, 700), min(i, 30000)
    elif huge and rng.random() < 0.15:      # --huge: catalogues past 131 072 items (the filter's stage summary, the sparser bar forms)
        b, i = near(rng, 300 if rng.random() < 0.8 else 4300), int(rng.integers(131_000, 760_000))
    k = int(min(i, rng.choice([1, 2, 5, 10, 20, 40, 64, 100, int(rng.integers(1, 129))])))
    style = rng.choice(['gauss', 'ties', 'range', 'zeros', 'nonfinite', 'tiny'])
    if style == 'gauss':
        u = rng.standard_normal((b, d)) * 0.1
        it = rng.standard_normal((i, d)) * 0.1
    elif style == 'ties':      # a few distinct values: thousands of exactly equal scores, the order among them is by item id
        u = rng.integers(-2, 3, size=(b, d)).astype(np.float64)
        it = rng.integers(-1, 2, size=(i, d)).astype(np.float64)
    elif style == 'range':
        u = rng.standard_normal((b, d)) * np.exp2(rng.integers(-20, 12, size=(b, 1)))
        it = rng.standard_normal((i, d)) * np.exp2(rng.integers(-20, 12, size=(i, 1)))
    elif style == 'zeros':
        u = rng.standard_normal((b, d)) * (rng.random((b, 1)) < 0.7)
        it = rng.standard_normal((i, d)) * (rng.random((i, 1)) < 0.5)
    elif style == 'tiny':
        u = rng.standard_normal((b, d)) * 1e-20
        it = rng.standard_normal((i, d)) * 1e-20
    else:
        u = rng.standard_normal((b, d))
        it = rng.standard_normal((i, d))
        it[rng.integers(0, i), rng.integers(0, d)] = np.inf
        if i > 3:
            it[rng.integers(0, i), rng.integers(0, d)] = np.nan
        if rng.random() < 0.3:
            u[rng.integers(0, b), rng.integers(0, d)] = -np.inf
    u, it = u.astype(np.float32), it.astype(np.float32)
    mask = None
    if rng.random() < 0.6:
        hi = int(min(i - k, rng.choice([0, 3, 30, 200])))
        cnt = rng.integers(0, hi + 1, size=b) if hi > 0 else np.zeros(b, dtype=np.int64)
        rp = np.zeros(b + 1, dtype=np.int64)
        np.cumsum(cnt, out=rp[1:])
        items = np.concatenate([np.sort(rng.choice(i, size=c, replace=False)) for c in cnt]) if rp[-1] else np.zeros(0, np.int64)
        mask = (rp, items)
    ids = None
    if rng.random() < 0.4:
        n_tab = b + int(rng.integers(0, 50))
        table = np.zeros((n_tab, d), dtype=np.float32)
        ids = rng.permutation(n_tab)[:b].astype(np.int64)
        table[ids] = u
        u = table
    return dict(seed=seed, b=b, i=i, d=d, k=k, style=str(style), masked=mask is not None, ids=ids is not None,
                round4=bool(rng.random() < 0.5)), u, it, mask, ids


def bits(t):
    return t.contiguous().view(torch.int32)


def run_case(dev, desc, u, it, mask, ids):
    """returns None, or a string describing the first mismatch"""
    from textgcn_amd import scoring
    ud, itd = torch.from_numpy(u).to(dev), torch.from_numpy(it).to(dev)
    idd = None if ids is None else torch.from_numpy(ids).to(dev)
    rp = it_ = None
    if mask is not None:
        rp = torch.from_numpy(mask[0].astype(np.int32)).to(dev)
        it_ = torch.from_numpy(mask[1].astype(np.int32)).to(dev)
        if it_.numel() == 0:
            it_ = torch.zeros(1, dtype=torch.int32, device=dev)
    k, r4 = desc['k'], desc['round4']
    s = scoring.score_dense(ud, itd, user_ids=idd)
    if mask is not None:
        scoring.mask_train(s, rp, it_)
    rv, ri = scoring.topk(s, k, round4=r4)
    runs = [('fp32', dict()), ('prefilter', dict(prefilter=True)), ('prefilter+pack', dict(prefilter=True, item_pack=scoring.item_pack(itd)))]
    for name, kw in runs:
        v, idx = scoring.score_topk(ud, itd, k, user_ids=idd, mask_rowptr=rp, mask_items=it_, round4=r4, **kw)
        torch.cuda.synchronize()
        if not torch.equal(idx, ri):
            bad = (idx != ri).any(dim=1).nonzero().flatten()
            r = int(bad[0])
            return f'{name}: item lists differ for {bad.numel()} users, first user {r}: got {idx[r].tolist()[:8]} want {ri[r].tolist()[:8]}'
        if not torch.equal(bits(v), bits(rv)):
            return f'{name}: scores differ in {(bits(v) != bits(rv)).sum().item()} places'
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=120)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--max-cases', type=int, default=100000)
    ap.add_argument('--wide', action='store_true', help='only widths above 128 (the wide bf16 filter)')
    ap.add_argument('--huge', action='store_true', help='one case in seven (of the narrow widths) on a catalogue of 131 000 - 760 000 items')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    t0 = time.time()
    n = fails = 0
    by_style = {}
    seed = args.seed0
    while time.time() - t0 < args.seconds and n < args.max_cases:
        desc, u, it, mask, ids = draw_case(seed, args.wide, args.huge)
        try:
            err = run_case(dev, desc, u, it, mask, ids)
        except Exception as e:       # an exception is a finding too
            err = f'{type(e).__name__}: {e}'
        if err:
            fails += 1
            print(json.dumps({'fail': err, **desc}), flush=True)
        by_style[desc['style']] = by_style.get(desc['style'], 0) + 1
        n += 1
        seed += 1
        if n % 25 == 0:
            print(json.dumps({'progress': n, 'fails': fails, 's': round(time.time() - t0, 1)}), flush=True)
    print(json.dumps({'cases': n, 'fails': fails, 'seed0': args.seed0, 'next_seed': seed, 'styles': by_style}), flush=True)
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
