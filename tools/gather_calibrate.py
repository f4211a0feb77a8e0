#!/usr/bin/env python3
"""Known-traffic GATHER launches of the production SpMM kernel, for calibrating rocprofv3's memory-side counters on the access
pattern `roofline.traffic` is quoted for.

`tools/pmc_calibrate.py` calibrates on the identity matrix: row r reads row r, a streaming pattern.  The layer launches
gather random 4d-byte rows, so this tool issues, per table size T in a sweep from far below the 256 MiB Infinity Cache to far
above it:

  once : rows of 64 stored entries whose columns are a random PERMUTATION of the table's rows -- every row of the table is
         fetched exactly once per launch, so the bytes that must cross the L2's memory side are known exactly
         (T + 8 B per entry of (col, val) + the row pointers), whatever the caches do inside a launch;
  unif : the `bench.random_row_rate` probe itself: 2^23 entries with uniformly random columns (rows repeat).

Launch order per (T, pattern): 1 warm-up + LAUNCHES timed launches; the profiler's dispatch order is the key the summariser
joins on (`Grid_Size` is printed too).  Under `rocprofv3 --pmc <counter> --kernel-trace` the printed byte counts are compared
with the counters by `tools/summarize_profiles.py --gather-cal`; without a profiler the timings give the delivered row rate per
table size (Infinity-Cache resident vs HBM).

    python tools/gather_calibrate.py [--sizes-mb 16,64,...] [--d 64]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd.propagate import DeviceCSR, spmm  # noqa: E402

LAUNCHES = 3
PER_ROW = 64


def run(csr, x, y):
    spmm(csr, x, y=y, exact=True, variant=1)      # (SPMM_WAVE_PER_ROW: the kernel the counter factors were measured on)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(LAUNCHES):
        spmm(csr, x, y=y, exact=True, variant=1)      # (SPMM_WAVE_PER_ROW: the kernel the counter factors were measured on)
    ev1.record()
    ev1.synchronize()
    return ev0.elapsed_time(ev1) / LAUNCHES * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes-mb', default='16,64,128,192,256,384,512,1024,2048,4096')
    ap.add_argument('--d', type=int, default=64)
    ap.add_argument('--patterns', default='once,unif')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    d = args.d
    rng = np.random.default_rng(7)
    seq = 0
    for mb in [int(s) for s in args.sizes_mb.split(',')]:
        n_rows = (mb << 20) // (4 * d)
        n_rows -= n_rows % PER_ROW
        x = torch.randn((n_rows, d), device=dev)
        for pattern in args.patterns.split(','):
            if pattern == 'once':
                cols = rng.permutation(n_rows).astype(np.int32)
            else:
                cols = rng.integers(0, n_rows, size=1 << 23, dtype=np.int64).astype(np.int32)
            n_out = len(cols) // PER_ROW
            csr = DeviceCSR(np.arange(n_out + 1, dtype=np.int64) * PER_ROW, cols, np.ones(len(cols), dtype=np.float32), n_rows, dev,
                            order_rows=False)
            y = torch.empty((n_out, d), device=dev)
            t = run(csr, x, y)
            distinct = n_rows if pattern == 'once' else int(len(np.unique(cols)))
            rec = {'table_MB': mb, 'pattern': pattern, 'd': d, 'entries': int(len(cols)), 'rows_out': n_out,
                   'grid_size': ((n_out + 3) // 4) * 256, 'first_dispatch': seq, 'launches': LAUNCHES + 1,
                   # bytes that must pass the L2's memory side per launch if nothing is found in L2 across launches:
                   # every DISTINCT table row once (once: all of them) + the (col, val) stream + row pointers
                   'min_read_bytes': distinct * 4 * d + len(cols) * 8 + (n_out + 1) * 4,
                   'gathered_bytes': int(len(cols)) * 4 * d, 'write_bytes': n_out * 4 * d,
                   'us_per_launch': round(t * 1e6, 2), 'gathered_GBs': round(len(cols) * 4 * d / t / 1e9, 1)}
            seq += LAUNCHES + 1
            print(json.dumps(rec), flush=True)
            del csr, y
        del x
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
