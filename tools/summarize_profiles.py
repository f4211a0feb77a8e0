#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the small, tracked summaries under profiles/.

    python tools/summarize_profiles.py --round r01 --kt gpurun_out/prof_kt --pmc gpurun_out/pmc

--kt : a `rocprofv3 --kernel-trace --stats --output-format csv` directory of `python bench.py ...`
--pmc: directory holding bench_<COUNTERS>/ and cal_<COUNTERS>/ passes (`rocprofv3 --pmc ... --kernel-trace`),
       each collected in its own run (MI355X_MICROARCH.md §rocprofv3 PMC slots)
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.json and refreshes profiles/hbm_traffic.json,
which bench.py reads for `roofline.traffic`.
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SPMM_SOURCES = ('textgcn_amd/csrc/tgcn_spmm.hip', 'textgcn_amd/propagate.py', 'textgcn_amd/graph.py')   # as bench.py


def spmm_sources_sha16():
    h = hashlib.sha256()
    for f in SPMM_SOURCES:
        h.update(open(os.path.join(ROOT, f), 'rb').read())
    return h.hexdigest()[:16]


def short(name):
    name = name.replace('tgcn::(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0]


def pmc_table(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(os.path.join(d, '*', '*counter_collection.csv')) + glob.glob(os.path.join(d, '*counter_collection.csv'))
    files = [max(files, key=os.path.getmtime)] if files else []   # gpurun merges successive runs into one directory: newest only
    for f in files:
        rows = [r for r in csv.DictReader(open(f)) if 'tgcn' in r['Kernel_Name']]
        # bench.py also launches the one-wave-per-row kernel for its live random-row-rate probe (a small grid): the layer
        # launches of a workload are the dispatches with the largest grid of that kernel
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[short(r['Kernel_Name'])] = max(biggest[short(r['Kernel_Name'])], int(r['Grid_Size']))
        for r in rows:
            k = short(r['Kernel_Name'])
            if k.startswith('k_spmm_wave') and int(r['Grid_Size']) != biggest[k]:
                continue
            out[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: {'n': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v)} for c, v in cs.items()}
            for k, cs in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--round', default='r01')
    ap.add_argument('--kt')
    ap.add_argument('--pmc')
    ap.add_argument('--workload', default='c2')
    args = ap.parse_args()
    prof = os.path.join(ROOT, 'profiles')
    os.makedirs(prof, exist_ok=True)
    if args.kt:
        f = max(glob.glob(os.path.join(args.kt, '*', '*kernel_stats.csv')), key=os.path.getmtime)   # newest run
        shutil.copyfile(f, os.path.join(prof, f'{args.round}_kernel_stats.csv'))
        print('kernel stats ->', f'profiles/{args.round}_kernel_stats.csv')
    if not args.pmc:
        return
    summary = {}
    for d in sorted(glob.glob(os.path.join(args.pmc, '*_*'))):
        if os.path.isdir(d):
            summary[os.path.basename(d)] = pmc_table(d)
    known = None
    cal_log = os.path.join(args.pmc, 'cal_FETCH_SIZE.log')
    if os.path.exists(cal_log):
        for line in open(cal_log):
            if line.startswith('{'):
                known = json.loads(line)
    summary['calibration_known_bytes'] = known
    # ---- HBM-side traffic of one SpMM layer launch (bench pass), corrected per MI355X_MICROARCH.md §HBM:
    # FETCH_SIZE (KiB) tallies 128-B requests at 64 B for wide coalesced reads -> x2; WRITE_SIZE is exact.
    fs, wsz = summary.get('bench_FETCH_SIZE', {}), summary.get('bench_WRITE_SIZE', {})
    main = [k for k in fs if k.startswith(('k_spmm_seg<', 'k_spmm_wave', 'k_spmm_group'))]
    main = sorted(main, key=lambda k: -fs[k]['FETCH_SIZE']['mean'] * fs[k]['FETCH_SIZE']['n'])[:1]
    spmm = main + [k for k in fs if k.startswith(('k_spmm_seg_reduce', 'k_spmm_long_reduce')) and main
                   and (k.startswith('k_spmm_seg_reduce') == main[0].startswith('k_spmm_seg<'))]
    if spmm:
        k = spmm[0]
        n_main = fs[k]['FETCH_SIZE']['n']
        # one layer = the main launch + its reduce launch: per-layer bytes = sum over the kernels of (total / layers)
        fetch_kib = sum(fs[x]['FETCH_SIZE']['mean'] * fs[x]['FETCH_SIZE']['n'] for x in spmm) / n_main
        write_kib = sum(wsz[x]['WRITE_SIZE']['mean'] * wsz[x]['WRITE_SIZE']['n'] for x in spmm) / wsz[k]['WRITE_SIZE']['n']
        cal = {}
        if known and 'cal_FETCH_SIZE' in summary:
            ck = [x for x in summary['cal_FETCH_SIZE'] if x.startswith('k_spmm')][0]
            cal = {'fetch_counter_bytes': summary['cal_FETCH_SIZE'][ck]['FETCH_SIZE']['mean'] * 1024,
                   'known_read_bytes': known['read_bytes_per_launch'],
                   'write_counter_bytes': summary['cal_WRITE_SIZE'][ck]['WRITE_SIZE']['mean'] * 1024,
                   'known_write_bytes': known['write_bytes_per_launch']}
        traffic = {'hbm_bytes_per_layer': int(2 * fetch_kib * 1024 + write_kib * 1024),
                   'read_bytes_corrected': int(2 * fetch_kib * 1024), 'write_bytes': int(write_kib * 1024),
                   'fetch_size_kib_raw': fetch_kib, 'write_size_kib_raw': write_kib, 'kernels': spmm,
                   'correction': "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x1; the counters sit on "
                                 "the L2's memory side, so Infinity-Cache hits are included (this is L2-miss traffic)",
                   'calibration': cal, 'round': args.round,
                   # bench.py withholds the figure when the SpMM sources no longer hash to this
                   'sources_sha16': spmm_sources_sha16()}
        hits = summary.get('bench_TCC_HIT_sum_TCC_MISS_sum', {}).get(k)
        if hits:
            h, m = hits['TCC_HIT_sum']['mean'], hits['TCC_MISS_sum']['mean']
            traffic['l2_hit_rate'] = h / (h + m)
        tf = os.path.join(prof, 'hbm_traffic.json')
        cur = json.load(open(tf)) if os.path.exists(tf) else {}
        cur[args.workload] = traffic
        json.dump(cur, open(tf, 'w'), indent=1)
        print(json.dumps(traffic, indent=1))
    json.dump(summary, open(os.path.join(prof, f'{args.round}_pmc_{args.workload}.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
