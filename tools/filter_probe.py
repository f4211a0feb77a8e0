#!/usr/bin/env python3
"""Diagnostic: where does a workgroup of the filter kernel spend its cycles?  Builds a PRIVATE copy of the library with
-DTGCN_FILTER_PROBE (shader-clock stamps per workgroup: start, after the prologue + first item stage, after the loop) under
gpurun_out/probe/, runs one fused scoring call of B users x 50 000 items x d = 64 and prints the per-phase cycle statistics.
The shipped libtgcn.so carries no stamp."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from textgcn_amd import _capi, build  # noqa: E402


def main():
    out_dir = os.path.join(ROOT, 'gpurun_out', 'probe')
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, 'libtgcn_probe.so')
    srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
    subprocess.check_call([build.hipcc()] + build.compile_flags() + ['-DTGCN_FILTER_PROBE=' + os.environ.get('PROBE', '1')] + srcs + build.link_flags() + ['-o', so])
    _capi.LIB_PATH = so
    from textgcn_amd import scoring
    lib = _capi.lib()
    raw = ctypes.CDLL(so)
    dev = torch.device('cuda:0')
    b, n_i, d, k = int(os.environ.get('B', 2048)), 50000, 64, 40
    g = torch.Generator().manual_seed(0)
    u = (torch.randn((b, d), generator=g) * 0.1).to(dev)
    it = (torch.randn((n_i, d), generator=g) * 0.1).to(dev)
    for _ in range(3):
        scoring.score_topk(u, it, k, round4=True)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        scoring.score_topk(u, it, k, round4=True)
    e.record()
    torch.cuda.synchronize()
    n_wg = ((b + 127) // 128) * 32
    buf = np.zeros(4 * 16384, dtype=np.uint64)
    assert raw.tgcn_probe_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    t = buf.reshape(-1, 4)[:min(n_wg, 16384)].astype(np.int64)
    pro, loop = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    start = t[:, 0] - t[:, 0].min()
    end = t[:, 2] - t[:, 0].min()
    q = lambda x: [int(v) for v in np.percentile(x, [0, 50, 90, 100])]  # noqa: E731
    hw = t[:, 3]
    xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
    cu, sh, se = (hwid >> 8) & 0xf, (hwid >> 12) & 0x1, (hwid >> 13) & 0x7
    where = xcc * 4096 + se * 64 + sh * 16 + cu
    ids, per_cu = np.unique(where, return_counts=True)
    by_count = {int(c): [int(np.median(loop[np.isin(where, ids[per_cu == c])])), int((per_cu == c).sum())] for c in np.unique(per_cu)}
    wg = np.arange(len(loop))
    bx, by = wg % ((b + 127) // 128), wg // ((b + 127) // 128)
    full = by < by.max()      # the last split is shorter
    print(json.dumps({'median_loop_by_user_tile': [int(np.median(loop[full & (bx == x)])) for x in range(int(bx.max()) + 1)],
                      'median_loop_by_split_first8': [int(np.median(loop[by == y])) for y in range(8)],
                      'cu_pair_loops_sample': [sorted(int(v) for v in loop[where == c]) for c in ids[:12]],
                      'blocks_on_sample_cus': [[(int(bx[j]), int(by[j])) for j in np.nonzero(where == c)[0]] for c in ids[:12]]}))
    print(json.dumps({'distinct_cus_used': int(len(ids)), 'workgroups_per_cu_histogram': {int(c): int((per_cu == c).sum()) for c in np.unique(per_cu)},
                      'median_loop_cycles_and_cus_by_workgroups_on_the_cu': by_count,
                      'per_xcc_workgroups': {int(x): int((xcc == x).sum()) for x in np.unique(xcc)},
                      'per_xcc_median_loop': {int(x): int(np.median(loop[xcc == x])) for x in np.unique(xcc)}}))
    print(json.dumps({'B': b, 'workgroups': int(n_wg), 'us_per_call': a.elapsed_time(e) / 10 * 1e3,
                      'cycles_prologue_min_med_p90_max': q(pro), 'cycles_loop_min_med_p90_max': q(loop),
                      'start_skew_cycles_min_med_p90_max': q(start), 'end_cycles_min_med_p90_max': q(end),
                      'ideal_loop_cycles_alone': 25 * 64 * 64, 'note': 's_memtime ticks; 100 MHz constant clock on gfx9 -> x (shader clock / 100 MHz)'}))


if __name__ == '__main__':
    main()
