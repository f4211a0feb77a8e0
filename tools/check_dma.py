#!/usr/bin/env python3
"""Checked run of the hand-ordered LDS-DMA waits of the two bf16 filters.

    python tools/check_dma.py build      # build container: tools/probes/bin/libtgcn_checkdma.so (-DTGCN_CHECK_DMA; travels to the box)
    python tools/check_dma.py run [pytest args]      # GPU box: the prefilter tests + the fuzz block on THAT library, then the counters

In the checked build every ring buffer is poisoned before a stage is requested into it, and after the counted wait
(DmaRingWait, csrc/tgcn_score_prefilter.hip) -- before the stage barrier -- every lane compares its pieces of the awaited stage
with their source bytes.  A piece that had not landed shows up as `stale_pieces` (and, the tests comparing every list bit for
bit, usually as a failed test too).  The product library carries none of this; the run's JSON goes to
gpurun_out/check_dma.json (copied to profiles/ when committed)."""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, 'tools', 'probes', 'bin')
SO = os.path.join(BIN, 'libtgcn_checkdma.so')
DEFAULT_TESTS = ['tests/test_score_gpu.py', 'tests/test_fuzz_gpu.py', 'tests/test_ltr.py', '-k',
                 'prefilter or fused or topk or wide or ltr or scoring or predict', '-q', '-m', 'gpu', '-x']


def main():
    from textgcn_amd import build
    if sys.argv[1] == 'build':
        os.makedirs(BIN, exist_ok=True)
        srcs = [os.path.join(build.CSRC, s) for s in build.SOURCES]
        subprocess.check_call([build.hipcc()] + build.compile_flags() + ['-DTGCN_CHECK_DMA'] + srcs + build.link_flags() + ['-o', SO])
        print(SO)
        return
    import pytest
    from textgcn_amd import _capi
    _capi.LIB_PATH = SO            # before anything loads the library
    args = sys.argv[2:] or DEFAULT_TESTS
    os.chdir(ROOT)
    rc = pytest.main(args)
    raw = ctypes.CDLL(SO)
    stale, checked = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    assert raw.tgcn_debug_dma_counts(ctypes.byref(stale), ctypes.byref(checked)) == 0
    rec = {'library': os.path.relpath(SO, ROOT), 'build_flag': '-DTGCN_CHECK_DMA', 'pytest_args': args, 'pytest_rc': int(rc),
           'checked_pieces': int(checked.value), 'stale_pieces': int(stale.value),
           'loaded_library': _capi.lib()._name}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'check_dma.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))
    sys.exit(0 if (rc == 0 and stale.value == 0 and checked.value > 0) else 1)


if __name__ == '__main__':
    main()
