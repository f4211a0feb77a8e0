#!/usr/bin/env python3
"""A/B timing of the fused scoring call's filter-kernel variants inside one process (development switch
tgcn_dev_set_filter_variant; not part of the ABI).  Single stream and three streams, B users per call."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd import _capi, scoring  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    b, n_i, d, k = int(os.environ.get('B', 2048)), 50000, 64, 40
    g = torch.Generator().manual_seed(0)
    u = (torch.randn((b, d), generator=g) * 0.1).to(dev)
    it = (torch.randn((n_i, d), generator=g) * 0.1).to(dev)
    rng = np.random.default_rng(0)
    cnt = rng.integers(20, 80, size=b)
    rp = np.zeros(b + 1, dtype=np.int32)
    np.cumsum(cnt, out=rp[1:])
    items = np.concatenate([np.sort(rng.choice(n_i, size=c, replace=False)) for c in cnt]).astype(np.int32)
    rp_d, it_d = torch.from_numpy(rp).to(dev), torch.from_numpy(items).to(dev)
    lib = _capi.lib()
    ref = None
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    main_s = torch.cuda.current_stream(dev)
    for variant in [int(x) for x in (sys.argv[1:] or ['0', '1'])]:
        lib.tgcn_dev_set_filter_variant(variant)
        out = scoring.score_topk(u, it, k, mask_rowptr=rp_d, mask_items=it_d, round4=True)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
        same = bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]))
        res = {'variant': variant, 'identical_to_first': same}
        for n_streams in (1, 3):
            def run(n):
                for s in streams[:n_streams]:
                    s.wait_stream(main_s)
                for j in range(n):
                    with torch.cuda.stream(streams[j % n_streams]):
                        scoring.score_topk(u, it, k, mask_rowptr=rp_d, mask_items=it_d, round4=True, slot=j % n_streams)
                for s in streams[:n_streams]:
                    main_s.wait_stream(s)
            run(6)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            run(30)
            e.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(e) / 30 * 1e3
            res[f'us_per_call_{n_streams}_streams'] = round(us, 1)
            res[f'mfma_frac_{n_streams}_streams'] = round(2.0 * d * b * n_i / (us * 1e-6) / 157.3e12, 4)
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
