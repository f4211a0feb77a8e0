#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from rocprofv3 PMC passes of tools/prefilter_pmc.py (tools/profile_round.sh, section mfma):
    python tools/mfma_table.py gpurun_out/prof/mfma/pre_64_16384 [more pass directories ...]
One markdown row per scoring kernel of each pass: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8; waits as shares of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import os
import re
import sys


def short(n):
    m = re.search(r'(k_\w+(<[^>]*>)?)', n)
    return m.group(1) if m else n[:40]


def main():
    print('| pass | kernel | launches | us | **matrix pipe busy** | clock GHz | `SQ_WAIT_ANY` | `SQ_WAIT_INST_ANY` |')
    print('|---|---|---|---|---|---|---|---|')
    for d in sys.argv[1:]:
        f = glob.glob(os.path.join(d, '*', '*counter_collection.csv'))[0]
        tr = glob.glob(os.path.join(d, '*', '*kernel_trace.csv'))[0]
        dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(tr))}
        disp = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            disp[(r['Dispatch_Id'], short(r['Kernel_Name']))][r['Counter_Name']] = float(r['Counter_Value'])
        agg = collections.defaultdict(list)
        for (did, kn), c in disp.items():
            if kn.startswith(('k_score_', 'k_rescore', 'k_sample', 'k_refine')):
                c['_ns'] = dur.get(did, 0)
                agg[kn].append(c)
        for kn, cs in sorted(agg.items()):
            m = lambda k: sum(c.get(k, 0.0) for c in cs) / len(cs)      # noqa: E731
            cyc = m('GRBM_GUI_ACTIVE') / 8
            wc = max(m('SQ_WAVE_CYCLES'), 1.0)
            us = m('_ns') / 1e3
            print(f"| {os.path.basename(d.rstrip('/'))} | `{kn}` | {len(cs)} | {us:.1f} | **{m('SQ_VALU_MFMA_BUSY_CYCLES') / max(cyc * 1024, 1):.2f}** | "
                  f"{cyc / max(us, 1e-9) / 1e3:.2f} | {m('SQ_WAIT_ANY') / wc:.2f} | {m('SQ_WAIT_INST_ANY') / wc:.2f} |")


if __name__ == '__main__':
    main()
