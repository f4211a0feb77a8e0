#!/usr/bin/env python3
"""How fast can ONE host thread issue fused scoring calls?  40 calls of tgcn_score_topk_prefilter_f32 (2048 users x 50 000 items x 64,
pack handed in: seven launches each) on 1 and 4 streams: host time to enqueue a call against the time the GPU takes per call.
MI355X box: 46 us of host time per call, 81 us per call on one stream, 62-65 on four -- at the reference's batch size the
four-stream pipeline sits within 15-40 % of what one issuing thread can feed; the model classes score 16 384 users per call
(340 us of GPU work per 46 us of host work).

    python tools/issue_rate.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from textgcn_amd import scoring
dev = torch.device('cuda:0')
b, calls, n_items, d = 2048, 40, 50000, 64
g = torch.Generator().manual_seed(0)
ue = (torch.randn(b * calls, d, generator=g) * 0.1).to(dev)
ie = (torch.randn(n_items, d, generator=g) * 0.1).to(dev)
rng = np.random.default_rng(0)
batches = []
for c in range(calls):
    ids = torch.arange(c * b, (c + 1) * b, dtype=torch.int64, device=dev)
    mi = np.sort(rng.integers(0, n_items, size=(b, 50)), axis=1)
    rows = [np.unique(r) for r in mi]
    rp = np.zeros(b + 1, dtype=np.int32); np.cumsum([len(r) for r in rows], out=rp[1:])
    batches.append((ids, torch.from_numpy(rp).to(dev), torch.from_numpy(np.concatenate(rows).astype(np.int32)).to(dev)))
pack = scoring.item_pack(ie)
for ns in (1, 4):
    side = [torch.cuda.Stream(dev) for _ in range(ns)]
    def run():
        keep = []
        for j, (ids, rp, it) in enumerate(batches):
            with torch.cuda.stream(side[j % ns]):
                keep.append(scoring.score_topk(ue, ie, 40, user_ids=ids, mask_rowptr=rp, mask_items=it, round4=True, slot=j % ns, prefilter=True, item_pack=pack))
        return keep
    run(); torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print({'streams': ns, 'cpu_issue_us_per_call': round((t1 - t0) / calls * 1e6, 1), 'total_us_per_call': round((t2 - t0) / calls * 1e6, 1)}, flush=True)
# the C entry point alone (arguments prebuilt, outputs preallocated): what of the 46 us is the library's seven launches
from textgcn_amd import _capi
lib = _capi.lib()
val = torch.empty((b, 40), dtype=torch.float32, device=dev)
idx = torch.empty((b, 40), dtype=torch.int64, device=dev)
ws = scoring._workspace(dev, max(lib.tgcn_score_topk_workspace_bytes(b, n_items, d, 40), 256), 0)
ids, rp, it = batches[0]
args = (_capi.ptr(ue), _capi.ptr(ids), b, _capi.ptr(ie), n_items, d, _capi.ptr(rp), _capi.ptr(it), 40, 1, _capi.ptr(pack), _capi.ptr(val),
        _capi.ptr(idx), _capi.ptr(ws), ws.numel(), _capi.current_stream(dev))
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        lib.tgcn_score_topk_prefilter_f32(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print({'c_entry_point_only_us_per_call': round((t1 - t0) / calls * 1e6, 1)}, flush=True)
