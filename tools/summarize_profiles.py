#!/usr/bin/env python3
"""Turn rocprofv3 output under gpurun_out/ into the small, tracked summaries under profiles/.

    python tools/summarize_profiles.py --round r01 --kt gpurun_out/prof_kt --pmc gpurun_out/pmc

--kt : a `rocprofv3 --kernel-trace --stats --output-format csv` directory of `python bench.py ...`
--pmc: directory holding bench_<COUNTERS>/ and cal_<COUNTERS>/ passes (`rocprofv3 --pmc ... --kernel-trace`),
       each collected in its own run (MI355X_MICROARCH.md §rocprofv3 PMC slots)
--gather-cal: directory written by tools/gather_cal_round.sh (known-traffic gathers over a table-size sweep): writes
       profiles/<round>_gather_calibration.json -- the MEASURED correction factors of the memory-side counters on the gather
       pattern and the delivered row rates per table size.  hbm_traffic.json is built with those factors, never with a literal.
Writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.json and refreshes profiles/hbm_traffic.json,
which bench.py reads for `roofline.traffic`.

What the counters are (measured, <round>_gather_calibration.json): FETCH_SIZE, TCC_EA0_RDREQ_* and TCC_MISS sit on the L2's
fabric side -- a 16 MB table that never leaves the Infinity Cache still reads its full size per launch on all of them -- so
`traffic` is L2-MISS (fabric-side) traffic, of which Infinity-Cache hits are a part; rocprofv3 on gfx950 lists no counter
behind the Infinity Cache (no MALL / UMC block), so HBM bytes can only be bounded or modelled (bench.py: `hbm_bytes_model`).
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


def _by_dispatch(d):
    """[{counter: value}] of the k_spmm_wave dispatches of one pass, in dispatch order"""
    f = max(glob.glob(os.path.join(d, '*', '*counter_collection.csv')), key=os.path.getmtime)
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'k_spmm_wave' in r['Kernel_Name']:
            by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    return [by[i] for i in sorted(by)]


def gather_calibration(d, rnd):
    """tools/gather_calibrate.py under rocprofv3: counters vs the known bytes of each launch (the launch's first dispatch is a
    warm-up and is skipped), per table size and pattern; the correction factors are the means over the each-row-once launches
    of tables beyond the Infinity Cache, where the known bytes cannot be served from any cache across launches."""
    plain = [json.loads(l) for l in open(os.path.join(d, 'plain.log')) if l.startswith('{')]
    passes = {os.path.basename(p)[4:]: _by_dispatch(p) for p in glob.glob(os.path.join(d, 'pmc_*')) if os.path.isdir(p)}
    rows = []
    for p in plain:
        sl = slice(p['first_dispatch'] + 1, p['first_dispatch'] + p['launches'])
        rec = {k: p[k] for k in ('table_MB', 'pattern', 'entries', 'min_read_bytes', 'gathered_bytes', 'write_bytes', 'us_per_launch', 'gathered_GBs')}
        for name, disp in passes.items():
            for c in name.split('+'):
                vals = [x[c] for x in disp[sl] if c in x]
                if vals:
                    rec[c] = sum(vals) / len(vals)
        if 'FETCH_SIZE' in rec:
            rec['FETCH_SIZE_bytes'] = rec['FETCH_SIZE'] * 1024
        if 'WRITE_SIZE' in rec:
            rec['WRITE_SIZE_bytes'] = rec['WRITE_SIZE'] * 1024
        if 'TCC_EA0_RDREQ_DRAM_32B_sum' in rec:
            rec['RDREQ_DRAM_32B_bytes'] = rec['TCC_EA0_RDREQ_DRAM_32B_sum'] * 32
        if 'TCC_MISS_sum' in rec:
            rec['TCC_MISS_x128_bytes'] = rec['TCC_MISS_sum'] * 128
            rec['l2_hit_rate'] = rec['TCC_HIT_sum'] / (rec['TCC_HIT_sum'] + rec['TCC_MISS_sum'])
        if 'TCC_EA0_RDREQ_128B_sum' in rec:
            rec['RDREQ_by_size_bytes'] = 32 * rec['TCC_EA0_RDREQ_32B_sum'] + 64 * rec['TCC_EA0_RDREQ_64B_sum'] + 128 * rec['TCC_EA0_RDREQ_128B_sum']
        rows.append(rec)
    big = [r for r in rows if r['pattern'] == 'once' and r['table_MB'] >= 512]
    mean = lambda xs: sum(xs) / len(xs)   # noqa: E731
    factors = {'what': 'known bytes / counter bytes, mean over the each-row-once launches of tables >= 512 MB (beyond the Infinity Cache)',
               'FETCH_SIZE': mean([r['min_read_bytes'] / r['FETCH_SIZE_bytes'] for r in big]),
               'RDREQ_DRAM_32B_x32': mean([r['min_read_bytes'] / r['RDREQ_DRAM_32B_bytes'] for r in big]),
               'TCC_MISS_x128': mean([r['min_read_bytes'] / r['TCC_MISS_x128_bytes'] for r in big]),
               'WRITE_SIZE': mean([r['write_bytes'] / r['WRITE_SIZE_bytes'] for r in big])}
    small = [r for r in rows if r['pattern'] == 'once' and r['table_MB'] <= 128]
    resident = mean([r['RDREQ_DRAM_32B_bytes'] / r['min_read_bytes'] for r in small])
    out = {'round': rnd, 'tool': 'tools/gather_cal_round.sh -> tools/gather_calibrate.py (production kernel k_spmm_wave<1,16>, d = 64)',
           'factors': factors,
           'infinity_cache_hits_are_counted': {
               'evidence': 'each-row-once launches of 16 / 64 / 128 MB tables (resident in the 256 MB Infinity Cache across the repeated '
                           'launches) still read counter bytes / known bytes = %.3f' % resident,
               'conclusion': 'FETCH_SIZE / TCC_EA0_RDREQ_* / TCC_MISS count L2-miss (fabric-side) requests; Infinity-Cache hits are inside them'},
           'rates_GBs': {
               'unif_table_le_256MB (Infinity Cache + L2)': mean([r['gathered_GBs'] for r in rows if r['pattern'] == 'unif' and 64 <= r['table_MB'] <= 256]),
               'unif_table_4GB (HBM, 1/16 of the table cacheable)': mean([r['gathered_GBs'] for r in rows if r['pattern'] == 'unif' and r['table_MB'] >= 4096]),
               'once_table_ge_1GB (HBM, every row once)': mean([r['gathered_GBs'] for r in rows if r['pattern'] == 'once' and r['table_MB'] >= 1024])},
           'rows': rows}
    json.dump(out, open(os.path.join(ROOT, 'profiles', f'{rnd}_gather_calibration.json'), 'w'), indent=1)
    print(json.dumps({k: out[k] for k in ('factors', 'infinity_cache_hits_are_counted', 'rates_GBs')}, indent=1))
    return out


def load_factors(rnd):
    """the measured gather-pattern factors of this (or the newest earlier) round"""
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_gather_calibration.json')))
    if not files:
        raise SystemExit('no profiles/r*_gather_calibration.json: run tools/gather_cal_round.sh and --gather-cal first '
                         '(the traffic figure is built from MEASURED counter factors only)')
    f = files[-1]
    return json.load(open(f))['factors'], os.path.relpath(f, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--round', default='r01')
    ap.add_argument('--kt')
    ap.add_argument('--pmc')
    ap.add_argument('--gather-cal')
    ap.add_argument('--workload', default='c2')
    args = ap.parse_args()
    prof = os.path.join(ROOT, 'profiles')
    os.makedirs(prof, exist_ok=True)
    if args.gather_cal:
        gather_calibration(args.gather_cal, args.round)
    if args.kt:
        f = max(glob.glob(os.path.join(args.kt, '*', '*kernel_stats.csv')), key=os.path.getmtime)   # newest run
        shutil.copyfile(f, os.path.join(prof, f'{args.round}_kernel_stats.csv'))
        print('kernel stats ->', f'profiles/{args.round}_kernel_stats.csv')
        # the stats file averages a kernel over every configuration of the bench run (c2's and c4's layers are the same two kernels):
        # the trace split by grid size gives the per-configuration layer launches bench.py's roofline.launch_us has to agree with
        tr = f.replace('kernel_stats.csv', 'kernel_trace.csv')
        if os.path.exists(tr):
            import collections
            import csv
            groups = collections.defaultdict(list)
            for r in csv.DictReader(open(tr)):
                n = short(r['Kernel_Name'])
                if n.startswith('k_spmm'):
                    groups[(n, int(r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            table = [{'kernel': k, 'grid_threads': g, 'calls': len(v), 'avg_us': round(sum(v) / len(v) / 1e3, 1),
                      'min_us': round(min(v) / 1e3, 1), 'max_us': round(max(v) / 1e3, 1)} for (k, g), v in sorted(groups.items())]
            with open(os.path.join(prof, f'{args.round}_spmm_launches_by_grid.json'), 'w') as fh:
                json.dump({'what': 'SpMM launches of the profiled bench.py run, split by grid size (= by configuration): a layer of config 4 is '
                                   'the largest k_spmm_seg + the largest k_spmm_reduce_groups', 'launches': table}, fh, indent=1)
            print('per-grid SpMM launches ->', f'profiles/{args.round}_spmm_launches_by_grid.json')
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
    # ---- fabric-side (L2-miss) traffic of one SpMM layer launch (bench pass).  Counter factors are the MEASURED ones of the
    # gather calibration (profiles/<round>_gather_calibration.json): FETCH_SIZE tallies the gathers' 128-B requests at 64 B
    # (x1.996 measured), TCC_EA0_RDREQ_DRAM_32B_sum x 32 B is exact (x0.999) and is preferred when its pass exists.
    factors, factors_src = load_factors(args.round)
    fs, wsz = summary.get('bench_FETCH_SIZE', {}), summary.get('bench_WRITE_SIZE', {})
    dr_name = 'bench_TCC_EA0_RDREQ_DRAM_32B_sum_TCC_HIT_sum_TCC_MISS_sum'
    dr = summary.get(dr_name, {})
    # the layer launches of the workload: the segmented pair when the workload runs it (bench.py's live random-row probe also
    # launches k_spmm_wave, with more bytes per launch than a layer: it must not be mistaken for the layer), else the
    # one-wave-per-row kernel and its long-row reduce
    seg = [k for k in fs if k.startswith('k_spmm_seg<')]
    if seg:
        spmm = seg[:1] + [k for k in fs if k.startswith(('k_spmm_reduce_groups', 'k_spmm_reduce_direct', 'k_spmm_seg_reduce'))]
    else:
        # round 4: a layer is k_spmm_groups (+ k_spmm_long_reduce); the random-row probe runs the one-wave-per-row kernel
        main = [k for k in fs if k.startswith('k_spmm_groups')] or [k for k in fs if k.startswith('k_spmm_wave')]
        main = sorted(main, key=lambda k: -fs[k]['FETCH_SIZE']['mean'] * fs[k]['FETCH_SIZE']['n'])[:1]
        spmm = main + [k for k in fs if k.startswith('k_spmm_long_reduce') and main]
    if spmm:
        k = spmm[0]
        n_main = fs[k]['FETCH_SIZE']['n']
        # one layer = the main launch + its reduce launch: per-layer bytes = sum over the kernels of (total / layers)
        fetch_kib = sum(fs[x]['FETCH_SIZE']['mean'] * fs[x]['FETCH_SIZE']['n'] for x in spmm) / n_main
        write_kib = sum(wsz[x]['WRITE_SIZE']['mean'] * wsz[x]['WRITE_SIZE']['n'] for x in spmm) / wsz[k]['WRITE_SIZE']['n']
        read_fetch = fetch_kib * 1024 * factors['FETCH_SIZE']
        read_dram = None
        if dr and k in dr and 'TCC_EA0_RDREQ_DRAM_32B_sum' in dr[k]:
            c = 'TCC_EA0_RDREQ_DRAM_32B_sum'
            read_dram = sum(dr[x][c]['mean'] * dr[x][c]['n'] for x in spmm if x in dr) / dr[k][c]['n'] * 32 * factors['RDREQ_DRAM_32B_x32']
        read_bytes = read_dram if read_dram is not None else read_fetch
        write_bytes = write_kib * 1024 * factors['WRITE_SIZE']
        traffic = {'fabric_bytes_per_layer': int(read_bytes + write_bytes),
                   'read_bytes': int(read_bytes), 'write_bytes': int(write_bytes),
                   'read_bytes_from_FETCH_SIZE': int(read_fetch), 'read_bytes_from_RDREQ_DRAM_32B': None if read_dram is None else int(read_dram),
                   'fetch_size_kib_raw': fetch_kib, 'write_size_kib_raw': write_kib, 'kernels': spmm,
                   'what': "bytes that crossed the L2's memory side (L2 misses + write-backs) per layer launch; Infinity-Cache hits are "
                           "INSIDE this figure (measured: " + factors_src + "), so it is an upper bound of the HBM bytes, not the HBM bytes",
                   'factors': {'FETCH_SIZE': factors['FETCH_SIZE'], 'RDREQ_DRAM_32B_x32': factors['RDREQ_DRAM_32B_x32'],
                               'WRITE_SIZE': factors['WRITE_SIZE'], 'source': factors_src}, 'round': args.round,
                   # bench.py withholds the figure when the SpMM sources no longer hash to this
                   'sources_sha16': spmm_sources_sha16()}
        if dr.get(k) and 'TCC_HIT_sum' in dr[k]:
            h, m = dr[k]['TCC_HIT_sum']['mean'], dr[k]['TCC_MISS_sum']['mean']
            traffic['l2_hit_rate'] = h / (h + m)
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
