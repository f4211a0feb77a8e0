#!/usr/bin/env python3
"""Known-traffic launch of the SpMM kernel for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this access
pattern (MI355X_MICROARCH.md §HBM: other access widths than 16 B/lane are uncalibrated).

A = identity over N rows, d = 64: one launch reads every row of X exactly once (256-byte wave loads, the same
instruction the real kernel gathers with), plus rowptr/colidx/vals, and writes every row of Y once.  With
N = 8M the tables (2 GiB each) are far beyond the 256 MiB Infinity Cache, so the bytes must come from HBM.
Prints the exact byte counts to compare with the counters.
"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textgcn_amd.propagate import DeviceCSR, spmm  # noqa: E402


def main():
    n, d = 8_000_000, 64
    dev = torch.device('cuda:0')
    rowptr = np.arange(n + 1, dtype=np.int64)
    colidx = np.arange(n, dtype=np.int32)
    vals = np.ones(n, dtype=np.float32)
    csr = DeviceCSR(rowptr, colidx, vals, n, dev)
    x = torch.randn((n, d), device=dev)
    y = torch.empty_like(x)
    for _ in range(3):
        spmm(csr, x, y=y, exact=True)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    print(json.dumps({'kernel': 'k_spmm_wave<1, 16>', 'launches': 3,
                      'read_bytes_per_launch': n * d * 4 + (n + 1) * 4 + n * 8, 'write_bytes_per_launch': n * d * 4}))


if __name__ == '__main__':
    main()
