#!/usr/bin/env python3
"""Ablation of the bf16 filters (k_score_prefilter_wide; the `pre_*` variants: the narrow k_score_prefilter): private copies of the
library with ONE ingredient of the loop compiled out (results of those copies are meaningless; only the kernel's duration is read).

    python tools/wide_ablate.py build [variant ...]      # build container: tools/probes/bin/libtgcn_<variant>.so (travels to the box)
    rocprofv3 --kernel-trace --stats ... -- python tools/wide_ablate.py run <variant> [d] [users] [items]     # GPU box, one variant per process
                                                         # (wide: d = 960; narrow: run pre_notests 64 16384 50000)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {'full': [], 'nodma': ['-DTGCN_WIDE_DMA=0'], 'notests': ['-DTGCN_WIDE_TESTS=0'], 'nomfma': ['-DTGCN_WIDE_MFMA=0'],
            'noldsread': ['-DTGCN_WIDE_LDSREAD=0'], 'nodma_notests': ['-DTGCN_WIDE_DMA=0', '-DTGCN_WIDE_TESTS=0'],
            # the narrow filter (run with d = 64 or 128)
            'pre_notests': ['-DTGCN_PRE_TESTS=0'], 'pre_nostage': ['-DTGCN_PRE_STAGE=0'], 'pre_noldsread': ['-DTGCN_PRE_LDSREAD=0'],
            'pre_nomfma': ['-DTGCN_PRE_MFMA=0'], 'pre_nostage_notests': ['-DTGCN_PRE_STAGE=0', '-DTGCN_PRE_TESTS=0'],
            'pre_onlymfma': ['-DTGCN_PRE_STAGE=0', '-DTGCN_PRE_TESTS=0', '-DTGCN_PRE_LDSREAD=0'],
            'pf1': ['-DTGCN_WIDE_PF=1']}
BIN = os.path.join(ROOT, 'tools', 'probes', 'bin')


def main():
    from textgcn_amd import build
    if sys.argv[1] == 'build':
        os.makedirs(BIN, exist_ok=True)
        srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
        only = sys.argv[2:]
        for name, flags in VARIANTS.items():
            if only and name not in only:
                continue
            so = os.path.join(BIN, f'libtgcn_{name}.so')
            subprocess.check_call([build.hipcc()] + build.compile_flags() + flags + srcs + build.link_flags() + ['-o', so])
            print(so)
        return
    name = sys.argv[2]
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 960
    b = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
    n_items = int(sys.argv[5]) if len(sys.argv) > 5 else 60000
    import torch
    from textgcn_amd import _capi
    _capi.LIB_PATH = os.path.join(BIN, f'libtgcn_{name}.so')
    from textgcn_amd import scoring
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    ue = (torch.randn(b, d, generator=g) * 0.1).to(dev)
    ie = (torch.randn(n_items, d, generator=g) * 0.1).to(dev)
    pack = scoring.item_pack(ie)
    for _ in range(3):
        scoring.score_topk(ue, ie, 40, prefilter=True, item_pack=pack)
    torch.cuda.synchronize()
    if name.startswith('stamp'):
        import ctypes
        import numpy as np
        raw = ctypes.CDLL(_capi.LIB_PATH)
        n_waves = ((b + 127) // 128) * 16 * 8
        buf = np.zeros((1 << 16, 8), dtype=np.uint64)
        assert raw.tgcn_wide_stamp_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(buf.nbytes)) == 0
        v = buf[:min(n_waves, 1 << 16)].astype(np.float64)
        v = v[v[:, 5] > 0]
        units = v[:, 5]
        med = lambda x: float(np.median(x))     # noqa: E731
        print({'waves': len(v), 'units_per_wave': med(units), 'loop_cycles_per_unit': med(v[:, 0] / units),
               'clock_GHz': med(v[:, 0] / v[:, 1] * 0.1), 'wait_vmcnt_cycles_per_unit': med(v[:, 2] / units),
               'barrier_cycles_per_unit': med(v[:, 3] / units), 'prologue_cycles': med(v[:, 4]),
               'p90_loop_per_unit': float(np.percentile(v[:, 0] / units, 90)), 'p90_barrier': float(np.percentile(v[:, 3] / units, 90)),
               'p90_wait': float(np.percentile(v[:, 2] / units, 90))})


if __name__ == '__main__':
    main()
