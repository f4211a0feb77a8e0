#!/usr/bin/env python3
"""Benchmark of the MI355X LightGCN propagation + scoring path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|small] ...

`python bench.py --gpus N` may be started plainly for ANY N: with N > 1 and no torch.distributed environment the process
starts the N ranks itself as child processes (`python -m torch.distributed.run --nproc-per-node N ... bench.py <same
arguments>`, before anything touches the GPU) and returns their exit code; started under torch.distributed.run
(RANK / WORLD_SIZE set, as the driver does for N > 1) it is one of the ranks.

One "step" = one K-layer propagation (the `representation` forward, TextGCN/base_model.py:93-106) over the
synthetic graph, inputs resident in HBM.  Headline value = propagated directed edges per second
= steps * K * nnz(A) / t  (SURVEY.md §8d metric 1), whole job.  The second metric of BASELINE.json (scored
user-item pairs/s: dense scores + train mask + top-40 per batch of 2048 users) is timed in a separate region
and reported under "scoring" in the same JSON line.

EVERY N times the same workload -- BASELINE config 4 (U=5M, I=2M, nnz=100M, d=64, K=3: the size the north-star target is
stated on; it fits one GPU) -- so the N = 1, 2, 4, 8 values are one curve on one graph:
N = 1: the whole graph on one GPU (this is the headline; `roofline`, `cpu_baseline` -- a 10 M-entry slice, labelled -- and
`verify` belong to it); the same line carries sub-records for the other single-GPU configurations: "c2" (config 2 with
scoring, training step, full CPU baseline), "c3" (d=128, K=4 + full-catalogue scoring) and "c5" (ltr_linear head), each with
its own roofline, cpu_baseline and verify entries (--sub to choose).
N > 1: config 4 row-sharded over the N ranks (fixed total work -> "strong"), one RCCL all-gather of the propagated user block
and one of the item block per layer, chunked so they run under the SpMM launches.  The graph is built once per node (local
rank 0) and memory-mapped by the other ranks; the line carries the ranks' device identities, a per-layer split of SpMM time /
time waiting for a gathered block, and the two halves ALONE on the same ranks and buffers (`layers.spmm_alone_ms`,
`layers.allgather_alone_ms`) with the overlap efficiency they imply.

roofline: HBM-bound SpMM.  `achieved` = ALGORITHMIC (compulsory) bytes of one layer launch / its mean duration (HIP events
on the launch stream around the timed region), bytes = nnz*8 + (rows+1)*4 + n_src*d*4 + rows*d*4 + fused layer-sum
traffic; `traffic` = bytes that crossed the L2's fabric side per layer launch (L2 misses + write-backs;
Infinity-Cache hits are INSIDE this figure) from the rocprofv3 PMC pass committed under profiles/, built with counter factors
measured on known-traffic gathers (`traffic_source` names both; null when the SpMM sources changed since that pass);
`hbm_bytes_model` = what an ideal 256 MB Infinity Cache would leave for HBM (gfx950 exposes no counter behind that cache);
`gather_bound` = what the same launch would take if every stored entry's 4d-byte row gather ran at the random-row rate
measured live for a table of this size.  The default scoring path (bf16 candidates + fp32 chains) carries its own roofline:
the largest of its three floors (bf16 MFMA time, candidates' row gathers at the live random-row rate, pack + pass-bit streams)
over the measured time of a call (`scoring.bf16_candidates.roofline`).
cpu_baseline: the torch CPU calls the reference makes (torch.sparse.mm on the coalesced COO x K, stack+mean), timed on
this host with all threads and with one (kind "port": oracle/torch_port.py); its output doubles as the `verify` check of
the timed GPU output (outside the timed region).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32 matrix peak (v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TF = 2500.0 # dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16; MI355X_MICROARCH.md: ~2.5 PF dense)
SPMM_SOURCES = ('textgcn_amd/csrc/tgcn_spmm.hip', 'textgcn_amd/propagate.py', 'textgcn_amd/graph.py')


def algorithmic_bytes_per_layer(nnz, n_rows, n_src, d, layer, n_layers, single):
    """Compulsory bytes of one SpMM layer launch: every CSR entry, every source row and every
    output row touched once; fused layer-sum reads acc_in and writes acc_out; the last layer stores no Y."""
    b = nnz * 8 + (n_rows + 1) * 4 + n_src * d * 4
    last = layer == n_layers
    if single:
        return b + n_rows * d * 4
    if not last:
        b += n_rows * d * 4          # Y
    b += 2 * n_rows * d * 4          # acc_in read + acc_out write
    return b


def spmm_sources_sha16():
    h = hashlib.sha256()
    for f in SPMM_SOURCES:
        h.update(open(os.path.join(ROOT, f), 'rb').read())
    return h.hexdigest()[:16]


def load_traffic(workload):
    """(fabric-side bytes per layer launch, source note) from the rocprofv3 PMC pass under profiles/ -- a recorded figure, not
    measured in this run; dropped (None) when the SpMM sources differ from the ones the pass was taken on.  The counters sit on
    the L2's memory side: Infinity-Cache hits are INSIDE the figure (profiles/r03_gather_calibration.json), so it is L2-miss
    traffic, an upper bound of the HBM bytes."""
    p = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        ent = json.load(open(p)).get(workload)
    except Exception:
        ent = None
    if not ent:
        return None, 'no PMC pass recorded for this workload (profiles/hbm_traffic.json)'
    f = ent.get('factors') or {}
    how = (f"TCC_EA0_RDREQ_DRAM_32B_sum x 32 B x {f.get('RDREQ_DRAM_32B_x32', 1):.3f}" if ent.get('read_bytes_from_RDREQ_DRAM_32B')
           else f"FETCH_SIZE x {f.get('FETCH_SIZE', 2):.3f}") + f" + WRITE_SIZE x {f.get('WRITE_SIZE', 1):.3f}"
    src = (f"profiles/hbm_traffic.json[{workload}]: rocprofv3 --pmc passes of round {ent.get('round')}, kernels "
           f"{ent.get('kernels') or ent.get('kernel')}, {how} (factors measured on known-traffic gathers: {f.get('source', 'n/a')}); "
           f"L2-miss bytes incl. Infinity-Cache hits; recorded, not measured in this run")
    if ent.get('sources_sha16') != spmm_sources_sha16():
        return None, src + ' -- STALE: the SpMM sources changed since that pass, figure withheld'
    return ent.get('fabric_bytes_per_layer', ent.get('hbm_bytes_per_layer')), src


INFINITY_CACHE_BYTES = 256 << 20      # MI355X_MICROARCH.md: 256 MiB die-level L3


def hbm_bytes_model(deg, nnz, n_rows, d, fabric_bytes):
    """HBM bytes of one layer launch under an IDEAL Infinity Cache -- a model, because rocprofv3 on gfx950 exposes no counter
    behind that cache (no MALL / UMC block; every TCC_EA counter includes its hits: profiles/r03_gather_calibration.json).
    Streams touched once per layer (CSR entries, row pointers, the layer-sum read, every store) go to HBM whenever the layer's
    footprint exceeds the cache; of the gather misses the cache can at best keep the most-referenced rows that fit its 256 MiB
    (row c is gathered deg(c) times per layer) -- share h of the gathers.  Returns a dict; `bytes` is None when the whole
    per-layer footprint fits the cache (then HBM sees next to nothing after the first layer, and the launch is bound by the
    caches, not by HBM)."""
    row = 4 * d
    stream_read = nnz * 8 + (n_rows + 1) * 4 + n_rows * row          # CSR + rowptr + acc_in
    stores = 2 * n_rows * row                                          # Y + acc_out
    footprint = stream_read + stores + n_rows * row                   # + the gathered table
    out = {'what': 'ideal-LRU Infinity Cache model (no counter exists behind that cache on gfx950): streamed-once bytes + the gather '
                   'misses the 256 MiB cache cannot hold', 'layer_footprint_bytes': int(footprint)}
    if fabric_bytes is None:
        return out
    if footprint <= INFINITY_CACHE_BYTES:
        out.update(bytes=None, note='the layer footprint fits the 256 MiB Infinity Cache: the fabric-side traffic can be served on-die; '
                                    'HBM is not the binding level for this workload')
        return out
    top = np.sort(np.asarray(deg))[::-1][:INFINITY_CACHE_BYTES // row]
    h = float(top.sum()) / float(nnz)
    gather_fabric = max(fabric_bytes - stream_read - stores, 0.0)
    out.update(bytes=int(stream_read + stores + gather_fabric * (1.0 - h)), cacheable_gather_share=round(h, 4),
               gather_fabric_bytes=int(gather_fabric), streamed_once_bytes=int(stream_read + stores))
    return out


def cpu_threads_note():
    return f'{torch.get_num_threads()} torch threads on {os.cpu_count()} host cpus'


def cpu_baseline_propagation(graph, e0, n_layers, gpu_out=None, budget_s=6.0):
    """The reference's own torch calls on this host's CPU (oracle/torch_port.py), bounded sample, all threads then one
    thread (torch's COO SpMM kernel is single-threaded).  The CPU output is also the checker
    of the timed GPU output (`verify`)."""
    from oracle import torch_port
    idx, val = graph.to_coo()
    a = torch_port.norm_matrix(idx, val, graph.n)
    all_threads = torch.get_num_threads()
    recs, ref = [], None
    for threads in (all_threads, 1):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        n_fwd = 0
        while True:
            ref = torch_port.representation(a, e0, n_layers)
            n_fwd += 1
            el = time.perf_counter() - t0
            if el > budget_s or n_fwd >= 3:
                break
        recs.append({'value': n_fwd * n_layers * graph.nnz / el, 'unit': 'edges/s', 'cores': threads, 'kind': 'port',
                     'sample': f'{n_fwd} full {n_layers}-layer CPU forward(s) of the same graph in {el:.1f} s; torch '
                               f'{torch.__version__} sparse.mm on coalesced COO (single-threaded kernel) + stack/mean, '
                               f'{threads} torch thread(s) on {os.cpu_count()} host cpus'})
    torch.set_num_threads(all_threads)
    rec = recs[0]
    rec['one_thread'] = recs[1]
    verify = None
    if gpu_out is not None:
        r = ref.numpy()
        g = gpu_out.cpu().numpy()
        verify = {'what': 'whole timed GPU output vs the CPU port of the reference forward (all rows)',
                  'normwise_max_err': float(np.abs(g.astype(np.float64) - r).max() / np.abs(r).max()), 'bar': 1e-4}
        verify['ok'] = bool(verify['normwise_max_err'] <= verify['bar'])
    return rec, verify


def cpu_baseline_scoring(users_emb, items_emb, mask_rowptr, mask_items, k, gpu_topk=None, budget_s=6.0):
    from oracle import torch_port
    b = users_emb.shape[0]
    t0 = time.perf_counter()
    n = 0
    while True:
        pv, pi = torch_port.score_mask_topk(users_emb, items_emb, mask_rowptr, mask_items, k)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 5:
            break
    rec = {'value': n * b * items_emb.shape[0] / el, 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
           'sample': f'{n} batch(es) of {b} users x {items_emb.shape[0]} items: torch.matmul + -inf mask + topk({k}) in {el:.1f} s'}
    verify = None
    if gpu_topk is not None:
        gv, gi = gpu_topk[0].cpu().numpy(), gpu_topk[1].cpu().numpy()
        same = (gi == pi.numpy()).all(axis=1)
        verify = {'what': f'top-{k} of the first timed batch vs the CPU port (matmul + mask + topk + round)',
                  'rows_identical': int(same.sum()), 'rows': int(len(same)),
                  'max_score_diff': float(np.abs(gv - pv.numpy())[np.isfinite(gv)].max())}
        # BLAS summation order differs from the k-ordered chain in the last bits: a few near-tied neighbours may swap
        verify['ok'] = bool(same.mean() >= 0.98 and verify['max_score_diff'] <= 1.01e-4)
    return rec, verify


def time_steps(step, steps, warmup, barrier):
    """W untimed warm-up steps, then exactly `steps` steps between barriers; (device seconds by HIP events on the launch
    stream, wall seconds)."""
    for _ in range(warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    barrier()
    return ev0.elapsed_time(ev1) / 1e3, time.perf_counter() - t0


def random_row_rate(n_rows, d, dev, entries=1 << 23):
    """Random-row gather rate of THIS chip for a [n_rows, d] fp32 table, measured live with the production kernel:
    rows of 64 stored entries with uniformly random columns (bytes = entries * 4d / launch time)."""
    from textgcn_amd.propagate import DeviceCSR, spmm
    rng = np.random.default_rng(1)
    per_row = 64
    n_out = entries // per_row
    cols = rng.integers(0, n_rows, size=n_out * per_row, dtype=np.int64)
    csr = DeviceCSR(np.arange(n_out + 1, dtype=np.int64) * per_row, cols, np.ones(len(cols), dtype=np.float32), n_rows, dev,
                    order_rows=False)
    x = torch.randn((n_rows, d), device=dev)
    y = torch.empty((n_out, d), device=dev)
    from textgcn_amd._capi import SPMM_WAVE_PER_ROW      # (its own kernel name: a profile never mistakes the probe for a layer)
    for _ in range(2):
        spmm(csr, x, y=y, exact=True, variant=SPMM_WAVE_PER_ROW)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    reps = 5
    for _ in range(reps):
        spmm(csr, x, y=y, exact=True, variant=SPMM_WAVE_PER_ROW)
    ev1.record()
    ev1.synchronize()
    return len(cols) * 4.0 * d / (ev0.elapsed_time(ev1) / 1e3 / reps)


def spmm_roofline(nnz_local, n_rows_local, n_src, d, K, t_dev, steps, traffic, traffic_src, dev, gather=True, deg=None,
                  kernel='one SpMM layer'):
    layer_bytes = [algorithmic_bytes_per_layer(nnz_local, n_rows_local, n_src, d, k, K, False) for k in range(1, K + 1)]
    mean_layer_s = t_dev / (steps * K)
    achieved = float(np.mean(layer_bytes)) / mean_layer_s / 1e9
    r = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
         'traffic_kind': "L2-miss bytes on the L2's fabric side; Infinity-Cache hits included, so NOT an HBM byte count",
         # recorded fabric-side bytes (PMC pass, profiles/) over the live launch time: what the fabric moves
         'traffic_GBs': round(traffic / mean_layer_s / 1e9, 1) if traffic else None,
         'kernel': kernel,
         'algorithmic_bytes_per_launch': int(np.mean(layer_bytes)), 'launch_us': round(mean_layer_s * 1e6, 2),
         'gather_model_GBs': round((nnz_local * (8 + 4 * d) + n_rows_local * d * 4) / mean_layer_s / 1e9, 1)}
    if deg is not None:
        m = hbm_bytes_model(deg, nnz_local, n_rows_local, d, traffic)
        if m.get('bytes'):
            m['GBs'] = round(m['bytes'] / mean_layer_s / 1e9, 1)
            m['frac_of_hbm_peak'] = round(m['bytes'] / mean_layer_s / 1e9 / HBM_PEAK_GBS, 4)
        r['hbm_bytes_model'] = m
    if gather:
        rate = random_row_rate(n_src, d, dev)
        t_gather = nnz_local * 4.0 * d / rate
        r['gather_bound'] = {'what': 'time of one layer if each stored entry cost one 4d-byte row gather at the random-row rate '
                                     'measured live (production kernel, uniform random rows of a table this size)',
                             'table_MB': round(n_src * d * 4 / 1e6, 1), 'random_row_rate_GBs': round(rate / 1e9, 1),
                             'layer_us_at_that_rate': round(t_gather * 1e6, 2),
                             'frac_of_gather_bound': round(t_gather / mean_layer_s, 4)}
    return r


def batch_masks(users, mrp, mit, dev, ids_origin=0):
    """[(user ids, mask rowptr, mask items)] device triple for one scoring call"""
    rowptr = np.zeros(len(users) + 1, dtype=np.int32)
    np.cumsum(mrp[users + 1] - mrp[users], out=rowptr[1:])
    if len(users) and np.all(np.diff(users) == 1):
        items = mit[mrp[users[0]]:mrp[users[-1] + 1]]
    else:
        items = np.concatenate([mit[mrp[x]:mrp[x + 1]] for x in users])
    return (torch.from_numpy(users - ids_origin).to(dev), torch.from_numpy(rowptr).to(dev),
            torch.from_numpy(np.ascontiguousarray(items)).to(dev))


SCORE_STREAMS = {False: 3, True: 4}     # set by --streams-fp32 / --streams-prefilter (sweeps); no environment variable is read


def n_score_streams(prefilter):
    """calls in flight: 3 for the fp32 filter, 4 (LightGCN.predict_streams) for the bf16-candidate path -- sweeps of 2 / 3 / 4 / 6 on
    the final kernels, profiles/r02_experiments.md"""
    return SCORE_STREAMS[bool(prefilter)]


def pack_ksteps(d):
    """16-wide k-steps of a packed item row (tgcn_score_prefilter.hip pack_ksteps)"""
    return 4 if d <= 64 else 8 if d <= 128 else 16 if d <= 256 else 32 if d <= 512 else 64 if d <= 832 else 56 if d <= 896 else 60 if d <= 960 else 64


def candidate_path_roofline(b, n_items, d, k_top, t_call, dev, slot=0):
    """Roofline of ONE call of the default scoring path (tgcn_score_topk_prefilter_f32: bf16 candidate pass + fp32 chains for the
    candidates).  The call has three floors, none of which the others can hide below: the bf16 matrix pipe (2 K' B I flop, K' =
    the packed row width + the bound's k-step), the candidates' fp32 row gathers at the chip's random-row rate (pairs counted by
    tgcn_score_topk_stats on the last call of `slot`, rate measured live for a table of the item table's size), and the streams
    every call must move once (the item pack; the narrow path's pass-bit words, written and -- below 131 072 items -- read).  bound = the largest of the
    three; frac = bound time / measured time of the call."""
    from textgcn_amd import scoring
    # a slot's workspace holds the plan of ITS last call: ask a slot whose last call had this very shape (the tail chunk of a user
    # range is smaller and may have ended on any slot)
    want = (int(b), int(n_items), int(d), int(min(k_top, scoring.MAX_K_PER_PASS)), True)
    slots = [sl for sl in range(16) if scoring.last_call(dev, sl) == want]
    if not slots:
        return {'note': f'no stream slot ended on a call of shape {want}: per-call statistics unavailable'}
    st = scoring.call_stats(dev, b, n_items, d, k_top, True, slot=slots[0])
    ks = pack_ksteps(d)
    flop = 2.0 * 16 * (ks + 1) * b * n_items
    t_mfma = flop / (MFMA_BF16_PEAK_TF * 1e12)
    # random-row rate for rows of 4 d bytes from a table of this many bytes (wide rows: the probe kernel's widest row, same bytes)
    probe_d = d if d in (64, 128, 256) else 256
    rate = random_row_rate(max(64, n_items * d // probe_d), probe_d, dev, entries=1 << 21)
    gather_bytes = st['rescored_pairs'] * 4.0 * d
    t_gather = gather_bytes / rate
    wh = ((n_items + 63) // 64 + 3) & ~3
    # pass-bit words: written once by the filter; read back by the chains' kernel too, unless the call keeps the stage summary
    # (the one-store-per-stage form on 131 072 items and more: tgcn_score_fused.hip, kSummaryMinWords)
    tiles = (b + 255) // 256
    summarised = wh >= 2048 and (b > 4096 or n_items // max(1, min(32, 256 // tiles)) >= 16 * 256)
    stream_bytes = n_items * (32 * ks + 16) + (0 if d > 128 else (1 if summarised else 2) * (tiles * 256) * 2 * wh * 4)
    t_stream = stream_bytes / (HBM_PEAK_GBS * 1e9)
    floors = {'mfma_bf16': t_mfma, 'candidate_row_gathers': t_gather, 'pack_and_mask_streams': t_stream}
    which = max(floors, key=floors.get)
    return {'bound': which, 'what': 'max of three floors of one call: bf16 MFMA time of 2 K\' B I, candidates\' fp32 row gathers / live '
                                    'random-row rate, pack + pass-bit streams / HBM peak', 'frac': round(floors[which] / t_call, 4),
            'floors_us': {k: round(v * 1e6, 2) for k, v in floors.items()}, 'call_us': round(t_call * 1e6, 2),
            'mfma_frac_of_bf16_peak': round(t_mfma / t_call, 4), 'bf16_flop': flop, 'peak_TF': MFMA_BF16_PEAK_TF,
            'users': b, 'rescored_pairs_per_user': round(st['rescored_pairs'] / b, 1), 'kept_pairs_per_user': round(st['kept_pairs'] / b, 1),
            'logged_pairs_per_user': round(st['logged_pairs'] / b, 1), 'fallback_users': st['fallback_users'],
            'random_row_rate_GBs': round(rate / 1e9, 1), 'gather_bytes': int(gather_bytes), 'stream_bytes': int(stream_bytes),
            'note': 'the gather floor is what THIS call rescored, not a constant of the shape: a tighter bar (fewer rescored pairs per user) '
                    'lowers the floor and frac together with call_us -- compare call_us across builds, frac within one'}


def scoring_region(ue, ie, batches, k_top, dev, barrier, prefilter=False, streams=None):
    """Consecutive calls are independent: issued round-robin on a few HIP streams with their own scratch buffers, as
    LightGCN.predict does, so one call's small selection kernels run under the next call's GEMM.  Returns seconds.
    prefilter: tgcn_score_topk_prefilter_f32 (candidates from a bf16 pass, fp32 chains for every score: the same lists); the
    item-norm factor of its bound is computed inside the timed region, once per region as predict does per call."""
    from textgcn_amd import scoring
    main = torch.cuda.current_stream(dev)
    N_SCORE_STREAMS = streams or n_score_streams(prefilter)
    side = [torch.cuda.Stream(dev) for _ in range(N_SCORE_STREAMS)]

    def score_all(bts):   # the predict step of base_model.py:254-263, fused (tgcn_score_topk_f32)
        pack = scoring.item_pack(ie) if prefilter else None
        for st in side:
            st.wait_stream(main)
        keep = []
        for j, (ids, rp, it) in enumerate(bts):
            with torch.cuda.stream(side[j % N_SCORE_STREAMS]):
                keep.append(scoring.score_topk(ue, ie, k_top, user_ids=ids, mask_rowptr=rp, mask_items=it, round4=True,
                                               slot=j % N_SCORE_STREAMS, prefilter=prefilter, item_pack=pack))
        for st in side:
            main.wait_stream(st)
        return keep
    score_all(batches[:N_SCORE_STREAMS])
    barrier()
    # the region is a millisecond or two (20 calls): timed three times, the median reported (a single pass moved by +-15 % from run
    # to run of the whole benchmark)
    times = []
    for _ in range(3):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        keep = score_all(batches)
        ev1.record()
        barrier()
        times.append(ev0.elapsed_time(ev1) / 1e3)
    return sorted(times)[1], keep


# ---------------------------------------------------------------------------------------------------- single GPU (N = 1)
def spmm_kernel_names(prop, exact, segmented):
    if segmented:
        return 'one SpMM layer = k_spmm_seg (column-block tiles) + k_spmm_reduce_groups (piece sums beside the direct rows, in row groups)'
    names = 'k_spmm_groups (row groups; chunk waves of the rows the split plan cuts)'
    if not exact and prop.csr.n_chunks:
        names += ' + k_spmm_long_reduce'
    return 'one SpMM layer = ' + names


def cpu_slice_baseline_and_verify(graph, e0, prop, e0d, entries=10_000_000):
    """CPU sample for a graph whose full CPU forward is minutes (config 4: ~25 s per layer on torch's single-threaded COO kernel):
    the layer-1 product of the first user rows holding ~`entries` stored entries, all threads then one; the same
    rows of the GPU's layer 1 are checked against it, and the hottest row (a ~1 M-entry item row cut into ~1000 chunks) against a
    float64 dot over its entries."""
    from textgcn_amd import propagate
    r = int(np.searchsorted(graph.rowptr, entries))
    e = int(graph.rowptr[r])
    rows = np.repeat(np.arange(r, dtype=np.int64), np.diff(graph.rowptr[:r + 1]))
    a = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, np.asarray(graph.colidx[:e]).astype(np.int64)])),
                                torch.from_numpy(np.array(graph.vals[:e])), (r, graph.n)).coalesce()
    del rows
    all_threads = torch.get_num_threads()
    recs = []
    for threads in (all_threads, 1):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        ref = torch.sparse.mm(a, e0)
        el = time.perf_counter() - t0
        recs.append({'value': e / el, 'unit': 'edges/s', 'cores': threads, 'kind': 'port',
                     'sample': f'ONE layer of the first {r} user rows ({e} stored entries, 1/{graph.nnz // e} of a layer) in '
                               f'{el:.2f} s: torch.sparse.mm on the coalesced COO slice, {threads} thread(s) on {os.cpu_count()} host '
                               f'cpus; the whole-forward rate is this rate (extrapolated, BASELINE.md §3)'})
    torch.set_num_threads(all_threads)
    y1 = torch.empty_like(e0d)
    propagate.spmm(prop.csr, e0d, y=y1)
    got = y1[:r].cpu().numpy()
    hot = int(np.argmax(graph.degrees()))
    a0, a1 = int(graph.rowptr[hot]), int(graph.rowptr[hot + 1])
    ref_hot = (np.asarray(graph.vals[a0:a1]).astype(np.float64)[:, None] * e0.numpy()[np.asarray(graph.colidx[a0:a1])].astype(np.float64)).sum(axis=0)
    got_hot = y1[hot].cpu().numpy().astype(np.float64)
    v = {'what': f'layer 1 of the timed path: rows 0..{r} vs torch.sparse.mm on the CPU; hottest row {hot} ({a1 - a0} entries) '
                 f'vs a float64 dot',
         'normwise_max_err': float(np.abs(got.astype(np.float64) - ref.numpy()).max() / np.abs(ref.numpy()).max()),
         'hottest_row_normwise_err': float(np.abs(got_hot - ref_hot).max() / np.abs(ref_hot).max()), 'bar': 1e-4}
    v['ok'] = bool(v['normwise_max_err'] <= 1e-4 and v['hottest_row_normwise_err'] <= 1e-4)
    return dict(recs[0], one_thread=recs[1]), v


def scoring_record(ue, ie, users_all, mrp, mit, n_i, d, dev, args, barrier, world=1, reduce_max_sum=None, cpu=True, large=True):
    """BASELINE metric 2 on the propagated tables: `args.score_batches` calls of `args.score_batch_size` users (fp32 MFMA filter =
    the record's value; the bf16-candidate path -- the model classes' default -- beside it, outputs compared), the model
    classes' 16384-user calls, and the CPU port on a bounded sample."""
    from textgcn_amd import scoring as _sc
    k_top = 40
    bsz = args.score_batch_size
    n_batches = min(args.score_batches, max(1, len(users_all) // bsz))
    batches = [batch_masks(users_all[b * bsz:(b + 1) * bsz], mrp, mit, dev, ids_origin=users_all[0]) for b in range(n_batches)]
    ts, keep = scoring_region(ue, ie, batches, k_top, dev, barrier)
    first_topk = keep[0]
    lc = [sl for sl in range(n_score_streams(False)) if _sc.last_call(dev, sl) == (int(batches[0][0].numel()), n_i, d, k_top, False)]
    fb_fp32 = _sc.fallback_count(dev, int(batches[0][0].numel()), n_i, d, k_top, slot=lc[0]) if lc else None
    pairs = sum(int(bt[0].numel()) for bt in batches) * n_i
    if reduce_max_sum is not None:
        mx, sm = reduce_max_sum([ts, float(pairs)])
        ts, pairs = mx[0], sm[1]
    flops = 2.0 * d * pairs
    rec = {
        'metric': f'scored user-item pairs/sec (scores + train mask + top-40 fused, B={bsz} per call, {n_score_streams(False)} streams)',
        'value': pairs / ts, 'unit': 'pairs/s', 'batches': n_batches, 'ms_per_batch': ts / n_batches * 1e3, 'n_items': n_i,
        'users_to_exact_fallback_last_call': fb_fp32,
        'roofline': {'bound': 'mfma', 'achieved': round(flops / ts / 1e12 / max(world, 1), 2), 'peak': MFMA_F32_PEAK_TF,
                     'unit': 'TFLOP/s', 'frac': round(flops / ts / 1e12 / max(world, 1) / MFMA_F32_PEAK_TF, 4), 'traffic': None},
    }
    # the same calls with the candidates found by the bf16 matrix pass (every score and the order still come from the
    # fp32 chains): checked bit for bit against the fp32-filter outputs above, reported beside them -- `value` of this
    # record stays the fp32 path
    if d <= 128:
        tp, keep_p = scoring_region(ue, ie, batches, k_top, dev, barrier, prefilter=True)
        same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(keep, keep_p))
        if reduce_max_sum is not None:
            tp = reduce_max_sum([tp, 0.0])[0][0]
        rec['bf16_candidates'] = {
            'what': 'tgcn_score_topk_prefilter_f32: bf16 MFMA pass with a proven error bound keeps a superset of the candidates, '
                    'k-ordered fp32 chains rescore them; top-k lists and scores identical to the fp32 path',
            'value': pairs / tp, 'unit': 'pairs/s', 'ms_per_batch': tp / n_batches * 1e3, 'streams': n_score_streams(True),
            'identical_to_fp32_path': bool(same), 'speedup': round(ts / tp, 3),
            'roofline': candidate_path_roofline(int(batches[0][0].numel()), n_i, d, k_top, tp / n_batches, dev)}
        del keep_p
    # the model class scores 16384 users per call (LightGCN.predict_chunk); same kernels, fewer launches
    big = min(16384, len(users_all))
    if large and reduce_max_sum is None and big > bsz:
        from textgcn_amd import scoring
        n_big = max(1, min(4, len(users_all) // big))
        bb = [batch_masks(users_all[b * big:(b + 1) * big], mrp, mit, dev) for b in range(n_big)]
        # calls in flight as LightGCN.predict_tensors keeps them (two for calls that run for milliseconds: config 4's catalogue)
        inflight = (scoring.calls_in_flight(big, n_i, n_score_streams(False)), scoring.calls_in_flight(big, n_i, n_score_streams(True)))
        tb, _ = scoring_region(ue, ie, bb, k_top, dev, barrier, streams=inflight[0])
        pb = sum(int(bt[0].numel()) for bt in bb) * n_i
        rec['large_batch'] = {'users_per_call': big, 'value': pb / tb, 'unit': 'pairs/s', 'ms_per_call': tb / n_big * 1e3, 'streams': inflight[0],
                              'mfma_frac': round(2.0 * d * pb / tb / 1e12 / MFMA_F32_PEAK_TF, 4)}
        if d <= 128:
            tbp, _ = scoring_region(ue, ie, bb, k_top, dev, barrier, prefilter=True, streams=inflight[1])
            rec['large_batch']['bf16_candidates'] = {
                'value': pb / tbp, 'unit': 'pairs/s', 'ms_per_call': tbp / n_big * 1e3, 'streams': inflight[1],
                'roofline': candidate_path_roofline(big, n_i, d, k_top, tbp / n_big, dev)}
    if cpu:
        # bounded CPU sample: the [users, I] matrix of the reference's matmul is 4 I bytes per user (8 MB at config 4)
        nb = int(min(batches[0][0].numel(), max(64, (1 << 30) // (4 * n_i))))
        bt = batches[0]
        rp = bt[1][:nb + 1].cpu().numpy()
        rec['cpu_baseline'], rec['verify'] = cpu_baseline_scoring(
            ue[bt[0][:nb]].cpu(), ie.cpu(), rp, bt[2][:int(rp[-1])].cpu().numpy() if int(rp[-1]) else np.zeros(1, dtype=np.int32), 40,
            gpu_topk=(first_topk[0][:nb], first_topk[1][:nb]))
    return rec


def record_single_gpu(wl, dev, args, steps, warmup, cpu=True, scoring=True, train=False):
    """One BASELINE configuration on ONE GPU, whole graph: timed K-layer forward, roofline, CPU baseline + verify, scoring."""
    from textgcn_amd import propagate, synth
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, K = synth.CONFIGS[wl]
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    graph = NormGraph.from_pairs(u, i, n_u, n_i)
    if not train:
        del u, i
    e0 = synth.embeddings(graph.n, d, seed=0)
    build_s = time.time() - t0
    thr = args.split_threshold or propagate.DEFAULT_SPLIT_THRESHOLD
    prop = propagate.Propagator(graph, dev, split_threshold=thr, segment=None if args.no_segment else 'auto')
    e0d = e0.to(dev)
    out = torch.empty_like(e0d)

    def barrier():
        torch.cuda.synchronize()

    def step():
        prop.forward(e0d, K, exact=args.exact, out=out)
    t_dev, t_wall = time_steps(step, steps, warmup, barrier)
    t = max(t_wall, t_dev)
    seg = bool(not args.exact and prop.csr.segment_blocks and any(prop.csr.segment_blocks))
    seg_note = 'none'
    if seg:
        seg_note = (f'user rows x{prop.csr.segment_blocks[0]}, item rows x{prop.csr.segment_blocks[1]} column blocks, '
                    f'{prop.csr.segment_tile}-entry tiles (tgcn_spmm_segmented_f32)')
    if not args.exact and not args.no_segment:
        traffic, tsrc = load_traffic(wl)    # the PMC pass was taken on the default path of the workload
    else:
        traffic, tsrc = None, 'no PMC pass for this mode'
    rec = {
        'metric': f'propagated edges/sec ({K}-layer SpMM, d={d})', 'value': steps * K * graph.nnz / t, 'unit': 'edges/s', 'n_gpus': 1,
        'steps': steps, 'warmup': warmup, 'ms_per_step': t / steps * 1e3, 'higher_is_better': True, 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{wl}: U={n_u} I={n_i} nnz={nnz} d={d} K={K}', 'nnz_A': graph.nnz, 'n_nodes': graph.n,
                   'max_degree': int(graph.degrees().max()),
                   'mode': 'exact (one fmaf chain per row)' if args.exact else f'rows in groups of consecutive rows, one wave each; rows > {thr} entries split in chunks',
                   'parity_of_this_mode': 'bit-identical to the reference CPU forward' if args.exact else
                   'rows cut by the long-row split / XCD segments are summed piecewise: normwise <= 1e-5 vs the exact chain '
                   '(tests), bar 1e-4; all other rows bit-identical',
                   'xcd_segments': seg_note, 'sharding': 'none (whole graph on one GPU)', 'graph_build_s': round(build_s, 1)},
        'roofline': spmm_roofline(graph.nnz, graph.n, graph.n, d, K, t_dev, steps, traffic, tsrc, dev, deg=graph.degrees(),
                                  kernel=spmm_kernel_names(prop, args.exact, seg)),
    }
    if cpu:
        if graph.nnz <= 30_000_000:
            rec['cpu_baseline'], rec['verify'] = cpu_baseline_propagation(graph, e0, K, gpu_out=out)
        else:
            rec['cpu_baseline'], rec['verify'] = cpu_slice_baseline_and_verify(graph, e0, prop, e0d)
    if scoring and not args.no_scoring:
        step()
        ue, ie = out[:n_u].contiguous(), out[n_u:].contiguous()
        mrp, mit = graph.train_mask()      # = train_mask_csr(u, i, n_u): the generator's pairs are distinct
        rec['scoring'] = scoring_record(ue, ie, np.arange(n_u), mrp, mit, n_i, d, dev, args, barrier, cpu=cpu)
        del ue, ie
    if train:
        rec['training'] = record_train_step(dev, u, i, graph, n_u, n_i, d, K)
    del prop, e0d, out
    torch.cuda.empty_cache()
    return rec


def _model_dataset(u, i, n_u, n_i, graph, text=None):
    import pandas as pd
    from textgcn_amd.graph import train_mask_csr
    rp, items = train_mask_csr(u, i, n_u)
    ds = types.SimpleNamespace(
        n_users=n_u, n_items=n_i, graph=graph, norm_matrix=None, mask_rowptr=rp, mask_items=items,
        true_test_lil=[[0]], train_user_dict=None, test_df=pd.DataFrame({'user_id': [0], 'asin': [0]}),
        user_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['u0']}), item_mapping=pd.DataFrame({'remap_id': [0], 'org_id': ['i0']}),
        all_items=range(n_i))
    if text:
        ds.__dict__.update(text)
    return ds


def _timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps / 1e3


def records_c3_c5(dev, want_c5=True, cpu=True):
    """BASELINE configs 3 and 5 through the MODEL CLASSES (LightGCN / LTRLinear), as a user of main.py would run them:
    c3 = U=180k I=60k nnz=1.6M d=128 K=4 + full-catalogue scoring of every user; c5 = ltr_linear on the frozen c3 tables
    + four text tables of width 384 (folded K = 960 GEMM)."""
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    from textgcn_amd.ltr import LTRLinear
    from textgcn_amd.model import LightGCN
    n_u, n_i, nnz, d, K = synth.CONFIGS['c3']
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    g = NormGraph.from_pairs(u, i, n_u, n_i)
    build_s = time.time() - t0
    p = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device=dev, load=None, batch_size=2048, quiet=True)
    ds = _model_dataset(u, i, n_u, n_i, g)
    m = LightGCN(p, ds)
    with torch.no_grad():    # weights from the CPU generator so that the CPU port sees the same table
        e0 = synth.embeddings(g.n, d, seed=0)
        m.embedding_user.weight.copy_(e0[:n_u])
        m.embedding_item.weight.copy_(e0[n_u:])
    users = np.arange(n_u)

    def fwd():
        with torch.no_grad():
            return m.representation
    reps = 10
    t_fwd = _timed(fwd, reps)
    m.score_prefilter = False     # the record's own numbers: fp32 MFMA filter; the bf16-candidate path is reported beside them
    t_all = _timed(lambda: m.predict_tensors(users), 2)    # representation + fused scoring of every user
    t_score = t_all - t_fwd
    pairs = n_u * n_i
    traffic, tsrc = load_traffic('c3')
    c3 = {'metric': f'propagated edges/sec ({K}-layer SpMM, d={d})', 'value': K * g.nnz / t_fwd, 'unit': 'edges/s', 'n_gpus': 1,
          'steps': reps, 'ms_per_step': t_fwd * 1e3, 'dtype': 'f32', 'data': 'synthetic',
          'config': {'workload': f'c3: U={n_u} I={n_i} nnz={nnz} d={d} K={K}', 'nnz_A': g.nnz, 'graph_build_s': round(build_s, 1),
                     'through': 'textgcn_amd.LightGCN.representation / predict_tensors'},
          'roofline': spmm_roofline(g.nnz, g.n, g.n, d, K, t_fwd, 1, traffic, tsrc, dev, deg=g.degrees(),
                                    kernel=spmm_kernel_names(m._engine, False, False)),
          'scoring': {'metric': 'scored user-item pairs/sec (full catalogue: every user x every item, mask + top-40 fused)',
                      'value': pairs / t_score, 'unit': 'pairs/s', 'ms_total': t_score * 1e3,
                      'roofline': {'bound': 'mfma', 'achieved': round(2.0 * d * pairs / t_score / 1e12, 2), 'peak': MFMA_F32_PEAK_TF,
                                   'unit': 'TFLOP/s', 'frac': round(2.0 * d * pairs / t_score / 1e12 / MFMA_F32_PEAK_TF, 4),
                                   'traffic': None}}}
    t1 = time.time()
    m.predict(users[:16384], with_scores=True)
    c3['scoring']['predict_wall_s_16384_users_incl_tolist'] = round(time.time() - t1, 3)
    # evaluate() for EVERY user (one random relevant item each): representation + full-catalogue top-k + metrics on the device
    from textgcn_amd.metrics import true_lists_csr
    m.test_users = users
    m._true_csr_host = true_lists_csr([[int(x)] for x in np.random.default_rng(3).integers(0, n_i, n_u)])
    m._true_dev = None
    m.evaluate()
    torch.cuda.synchronize()
    t1 = time.time()
    m.evaluate()
    torch.cuda.synchronize()
    c3['scoring']['evaluate_wall_s_all_users'] = round(time.time() - t1, 4)
    ref_v, ref_i = m.predict_tensors(users)
    m.score_prefilter = True      # the model class's default
    pv, pi = m.predict_tensors(users)
    t_all_p = _timed(lambda: m.predict_tensors(users), 2)
    m.evaluate()
    torch.cuda.synchronize()
    t1 = time.time()
    m.evaluate()
    torch.cuda.synchronize()
    c3['scoring']['bf16_candidates'] = {
        'what': 'LightGCN.score_prefilter = True (the class default): tgcn_score_topk_prefilter_f32',
        'value': pairs / (t_all_p - t_fwd), 'unit': 'pairs/s', 'ms_total': (t_all_p - t_fwd) * 1e3,
        'identical_to_fp32_path': bool(torch.equal(pv, ref_v) and torch.equal(pi, ref_i)),
        'evaluate_wall_s_all_users': round(time.time() - t1, 4),
        # per call of predict_chunk users (slot 0's last call is a full chunk); the calls overlap on the model's streams, so
        # the time of one call is the region's time / number of calls
        'roofline': candidate_path_roofline(min(m.predict_chunk, n_u), n_i, d, max(m.k), (t_all_p - t_fwd) / -(-n_u // m.predict_chunk), dev)}
    del ref_v, ref_i, pv, pi
    ue, ie = fwd()
    if cpu:
        c3['cpu_baseline'], c3['verify'] = cpu_baseline_propagation(g, e0, K, gpu_out=torch.cat([ue, ie]), budget_s=4.0)
        bt = batch_masks(users[:2048], ds.mask_rowptr, ds.mask_items, dev)
        from textgcn_amd import scoring
        gpu_topk = scoring.score_topk(ue.contiguous(), ie.contiguous(), 40, user_ids=bt[0], mask_rowptr=bt[1], mask_items=bt[2],
                                      round4=True)
        c3['scoring']['cpu_baseline'], c3['scoring']['verify'] = cpu_baseline_scoring(
            ue[:2048].cpu(), ie.cpu(), bt[1].cpu().numpy(), bt[2].cpu().numpy(), 40, gpu_topk=gpu_topk, budget_s=4.0)
    out = {'c3': c3}
    if not want_c5:
        return out
    gen = torch.Generator().manual_seed(5)
    t = 384
    text = {'items_as_desc': torch.randn((n_i, t), generator=gen), 'items_as_avg_reviews': torch.randn((n_i, t), generator=gen),
            'users_as_avg_reviews': torch.randn((n_u, t), generator=gen), 'users_as_avg_desc': torch.randn((n_u, t), generator=gen)}
    p5 = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device=dev, load=None, load_base=None, freeze=True,
                               batch_size=2048, quiet=True, ltr_layers=[])
    ltr = LTRLinear(p5, _model_dataset(u, i, n_u, n_i, g, text))
    with torch.no_grad():
        ltr.embedding_user.weight.copy_(e0[:n_u])
        ltr.embedding_item.weight.copy_(e0[n_u:])
    ltr.score_prefilter = False   # the record's own numbers: fp32 MFMA filter (as for c3); the bf16-candidate path beside them
    t_ltr = _timed(lambda: ltr.predict_tensors(users), 1) - t_fwd
    v_f, i_f = ltr.predict_tensors(users)
    ltr.score_prefilter = True    # the class default
    v_p, i_p = ltr.predict_tensors(users)
    same5 = bool(torch.equal(i_f, i_p) and torch.equal(v_f, v_p))
    del v_f, i_f, v_p, i_p
    t_ltr_pre = _timed(lambda: ltr.predict_tensors(users), 1) - t_fwd
    kf = d + 2 * t
    c5 = {'metric': 'scored user-item pairs/sec (ltr_linear: 5 text/embedding features + Linear(5,1) folded into one K=960 GEMM, '
                    'mask + top-40, full catalogue)', 'value': pairs / t_ltr, 'unit': 'pairs/s', 'n_gpus': 1, 'ms_total': t_ltr * 1e3,
          'dtype': 'f32', 'data': 'synthetic',
          'config': {'workload': f'c5: c3 graph + 4 text tables x {t}, ltr_linear', 'folded_K': int(ltr._k()),
                     'through': 'textgcn_amd.LTRLinear.predict_tensors'},
          'roofline': {'bound': 'mfma', 'achieved': round(2.0 * kf * pairs / t_ltr / 1e12, 2), 'peak': MFMA_F32_PEAK_TF,
                       'unit': 'TFLOP/s', 'frac': round(2.0 * kf * pairs / t_ltr / 1e12 / MFMA_F32_PEAK_TF, 4), 'traffic': None,
                       'algorithmic_flops': 2.0 * kf * pairs},
          'bf16_candidates': {'what': 'LTRLinear.score_prefilter = True (the class default): the folded K = 960 operands through '
                                      'tgcn_score_topk_prefilter_f32 (k_score_prefilter_wide + fp32 chains)',
                              'value': pairs / t_ltr_pre, 'unit': 'pairs/s', 'ms_total': t_ltr_pre * 1e3,
                              'identical_to_fp32_path': same5,
                              'roofline': candidate_path_roofline(min(ltr.ltr_predict_chunk, n_u), n_i, int(ltr._k()), max(ltr.k),
                                                                  t_ltr_pre / -(-n_u // ltr.ltr_predict_chunk), dev)}}
    if cpu:
        from oracle import torch_port
        nb = 256
        w = ltr.layers[0].weight.detach().cpu()
        bias = ltr.layers[0].bias.detach().cpu()
        uc, ic = ue[:nb].cpu(), ie.cpu()
        args = (uc, text['users_as_avg_reviews'][:nb], text['users_as_avg_desc'][:nb], ic, text['items_as_avg_reviews'],
                text['items_as_desc'], w, bias)
        t0 = time.perf_counter()
        n = 0
        while True:
            ref = torch_port.ltr_score_batchwise(*args)
            n += 1
            el = time.perf_counter() - t0
            if el > 6.0 or n >= 3:
                break
        c5['cpu_baseline'] = {'value': n * nb * n_i / el, 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                              'sample': f'{n} batch(es) of {nb} users x {n_i} items: the reference\'s 5 matmuls + cat + nn.Linear(5,1) '
                                        f'(ltr_models.py:131-146,200-204) in {el:.1f} s'}
        ids = torch.arange(nb, device=dev)
        with torch.no_grad():
            got = ltr.score_batchwise(ue[ids], ie, ids).cpu().numpy()
        err = float(np.abs(got.astype(np.float64) - ref.numpy()).max() / np.abs(ref.numpy()).max())
        c5['verify'] = {'what': f'[{nb}, {n_i}] ltr scores of the HIP path vs the CPU port', 'normwise_max_err': err, 'bar': 1e-4,
                        'ok': bool(err <= 1e-4)}
    out['c5'] = c5
    return out


def record_train_step(dev, u, i, graph, n_u, n_i, d, K, steps=10):
    """One BPR training step through the MODEL CLASS as `fit` issues it (get_loss -> backward -> fused Adam), dropout 0.4 drawn
    on the device: SURVEY.md §8(f) N1.  Wall time per step (the step has one host sync, the NaN check)."""
    from textgcn_amd.model import LightGCN
    ds = _model_dataset(u, i, n_u, n_i, graph)
    p = types.SimpleNamespace(k=[20, 40], emb_size=d, n_layers=K, device=dev, load=None, batch_size=2048, quiet=True, dropout=0.4,
                              lr=1e-3)
    m = LightGCN(p, ds)
    m.optimizer = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    rng = np.random.default_rng(0)
    batch = torch.from_numpy(np.stack([rng.integers(0, n_u, 2048), rng.integers(0, n_i, 2048), rng.integers(0, n_i, 2048)], axis=1))
    m._train_epoch([batch] * 3, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m._train_epoch([batch] * steps, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {'metric': 'BPR training step: dropout values + K-layer forward + pair loss + transposed backward + Adam, batch 2048',
            'ms_per_step': ms, 'steps': steps, 'native_loss_node': bool(m._native_loss()),
            'through': 'textgcn_amd.LightGCN._train_epoch (get_loss -> backward -> optimizer.step, as fit() does)'}


def shared_workload(wl, rank, local_rank, barrier):
    """N > 1: the synthetic graph and E0 are generated ONCE per node -- local rank 0 builds them (30 s and ~12 GB of host memory
    for config 4; eight concurrent builds would be eight times both) and publishes the arrays as .npy files in a node-local
    directory (memory-backed /dev/shm when present), the other ranks wait at a barrier and memory-map them: a rank then only
    touches the pages of its own row blocks.  Returns (NormGraph over the mapped arrays, E0 as a torch tensor over the map,
    directory).  The building rank removes the directory at exit, whatever way the run ends (atexit + the finally of main)."""
    import atexit
    import shutil
    import tempfile
    from textgcn_amd import synth
    from textgcn_amd.graph import NormGraph
    n_u, n_i, nnz, d, _ = synth.CONFIGS[wl]
    base = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
    path = os.path.join(base, f"tgcn_bench_{os.environ.get('MASTER_PORT', '0')}_{wl}")
    if local_rank == 0:
        shutil.rmtree(path, ignore_errors=True)      # a crashed earlier run with the same port
        os.makedirs(path, exist_ok=True)
        atexit.register(shutil.rmtree, path, ignore_errors=True)
        u, i = synth.interactions(n_u, n_i, nnz, seed=0)
        g = NormGraph.from_pairs(u, i, n_u, n_i)
        del u, i
        for name, arr in (('rowptr', g.rowptr), ('colidx', g.colidx), ('vals', g.vals),
                          ('e0', synth.embeddings(g.n, d, seed=0).numpy())):
            np.save(os.path.join(path, name + '.npy'), arr)
        del g
    barrier()
    m = {name: np.load(os.path.join(path, name + '.npy'), mmap_mode='c') for name in ('rowptr', 'colidx', 'vals', 'e0')}
    return NormGraph(n_u, n_i, m['rowptr'], m['colidx'], m['vals']), torch.from_numpy(m['e0']), path


def device_identity(dev):
    """what tells two ranks' GPUs apart in the JSON line: PCI address (domain:bus:device) where torch exposes it, + name"""
    p = torch.cuda.get_device_properties(dev)
    pci = ':'.join(f'{getattr(p, a):0{w}x}' for a, w in (('pci_domain_id', 4), ('pci_bus_id', 2), ('pci_device_id', 2)) if hasattr(p, a))
    return f"{pci or 'pci?'} {getattr(p, 'uuid', '')} {p.name}".strip()


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a torch.distributed environment: start the N ranks as CHILD processes (never an exec;
    this process has not touched the GPU) and hand back their exit code.  Rank 0's JSON line goes to this process's stdout."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    print(f'bench.py: starting {n} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def run_sharded(args, world, rank, local_rank, dev, dev_index, rehearsal):
    """N > 1 (or --force-sharded): BASELINE config 4 (default) row-sharded over the ranks."""
    import torch.distributed as dist
    from textgcn_amd import propagate, synth
    from textgcn_amd.dist import ColumnShardedPropagator, ShardedPropagator
    from textgcn_amd.graph import NormGraph
    wl = args.workload or 'c4'
    n_u, n_i, nnz, d, K = synth.CONFIGS[wl]
    wl_name = f'{wl}: U={n_u} I={n_i} nnz={nnz} d={d} K={K}'
    t0 = time.time()
    share_dir = None
    if world > 1:
        graph, e0, share_dir = shared_workload(wl, rank, local_rank, dist.barrier)
    else:
        u, i = synth.interactions(n_u, n_i, nnz, seed=0)
        graph = NormGraph.from_pairs(u, i, n_u, n_i)
        del u, i
        e0 = synth.embeddings(graph.n, d, seed=0)
    build_s = time.time() - t0
    thr = args.split_threshold or propagate.DEFAULT_SPLIT_THRESHOLD
    chunks = args.chunks or (4 if graph.nnz >= 50_000_000 else 1)
    if args.shard == 'features':
        sp = ColumnShardedPropagator(graph, d, rank, world, dev, split_threshold=thr, force_collective=args.force_sharded)
        e_cols = sp.local_e0(e0)

        def step():      # the K layers on this rank's columns + the one all-gather that assembles the d columns everywhere
            return sp.assemble(sp.forward(e_cols, K, exact=args.exact))
        n_rows_local, n_src, nnz_local = graph.n, graph.n, graph.nnz
    else:
        sp = ShardedPropagator(graph, rank, world, dev, split_threshold=thr, balance=args.balance, chunks=chunks,
                               force_collective=args.force_sharded)
        eu, ei = sp.local_e0(e0)

        def step():
            sp.forward(eu, ei, K, exact=args.exact)
        n_rows_local, n_src, nnz_local = sp.bu + sp.bi, sp.n_pad, sp.nnz_local

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def reduce_max_sum(vals):
        """(max over ranks, sum over ranks) of a small list of floats"""
        t = torch.tensor(vals, dtype=torch.float64, device='cpu' if rehearsal else dev)
        mx, sm = t.clone(), t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        return mx.tolist(), sm.tolist()

    try:
        t_dev, t_wall = time_steps(step, args.steps, args.warmup, barrier)
        t = reduce_max_sum([max(t_wall, t_dev)])[0][0]
        d_local = sp.dl if args.shard == 'features' else d
        roofline = spmm_roofline(nnz_local, n_rows_local, n_src, d_local, K, t_dev, args.steps, None, 'no PMC pass for this mode', dev,
                                 gather=False, kernel='one SpMM layer on this rank = one launch pair per row chunk (2 x chunks): k_spmm_groups + k_spmm_long_reduce, or '
                                        'k_spmm_seg + k_spmm_reduce_groups for the chunks config.xcd_segments counts')
        roofline['note'] = 'per rank: this rank\'s row blocks, launch time includes waiting for the all-gathered tables'
        result = {
            'metric': f'propagated edges/sec ({K}-layer SpMM, d={d})', 'value': args.steps * K * graph.nnz / t, 'unit': 'edges/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t / args.steps * 1e3, 'higher_is_better': True,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': wl_name, 'nnz_A': graph.nnz, 'n_nodes': graph.n, 'max_degree': int(graph.degrees().max()),
                       'mode': 'exact (one fmaf chain per row)' if args.exact else f'rows in groups of consecutive rows, one wave each; rows > {thr} entries split in chunks',
                       'parity_of_this_mode': 'bit-identical to the reference CPU forward' if args.exact else
                       'rows cut by the long-row split are summed piecewise: normwise <= 1e-5 vs the exact chain (tests), bar 1e-4; '
                       'all other rows bit-identical; the result does not depend on the number of ranks',
                       'xcd_segments': sp.segment_note() if hasattr(sp, 'segment_note') else 'none',
                       'sharding': (
                           f'feature-sharded x{world}: all rows, {sp.dl} of {d} columns per rank, no per-layer exchange, one RCCL all-gather of '
                           f'the combined table per forward' if args.shard == 'features' else
                           f'row-sharded x{world} ({args.balance}-balanced blocks padded to the largest, {sp.lay_u.chunks} row chunk(s) per '
                           f'block), RCCL all-gather per chunk and layer (users || item half-step)'),
                       'graph_build_s': round(build_s, 1)},
            'roofline': roofline,
        }
        if args.shard == 'rows' and d % world == 0 and d // world in (8, 16, 32, 64) and args.feature_partition:
            # the same forward under the feature partition (every rank: all rows, d / world columns, no per-layer exchange, one
            # all-gather at the end), timed the same way in the same run: the record's `value` stays the row partition's
            cp = ColumnShardedPropagator(graph, d, rank, world, dev, split_threshold=thr, force_collective=args.force_sharded)
            e_cols = cp.local_e0(e0)

            def step_cols():
                return cp.assemble(cp.forward(e_cols, K, exact=args.exact))
            tc_dev, tc_wall = time_steps(step_cols, args.steps, args.warmup, barrier)
            tc = reduce_max_sum([max(tc_wall, tc_dev)])[0][0]
            result['feature_partition'] = {
                'what': f'all rows, {cp.dl} of {d} columns per rank, no per-layer exchange, one RCCL all-gather of the combined table per '
                        f'forward (ColumnShardedPropagator); bit-identical to the row partition',
                'value': args.steps * K * graph.nnz / tc, 'unit': 'edges/s', 'ms_per_step': tc / args.steps * 1e3}
            cp.close()
            del cp, e_cols
            torch.cuda.empty_cache()
        # evidence that N ranks ran on N distinct devices, and where a layer's time went on the slowest rank
        ident = [None] * world
        dist.all_gather_object(ident, f'rank {rank}: cuda:{dev_index} {device_identity(dev)}')
        result['config']['world_size'] = dist.get_world_size()
        result['config']['backend'] = dist.get_backend()
        result['config']['devices'] = ident
        result['config']['distinct_devices'] = len({x.split(': ', 1)[1] for x in ident})
        if args.shard == 'rows':
            sp.record_events = True
            reps = 3
            tot = None
            for _ in range(reps):
                step()
                lt = sp.layer_times()
                tot = lt if tot is None else [{'layer': a['layer'], 'compute_ms': a['compute_ms'] + b['compute_ms'],
                                               'wait_on_gather_ms': a['wait_on_gather_ms'] + b['wait_on_gather_ms']}
                                              for a, b in zip(tot, lt)]
            sp.record_events = False
            mine = [{'layer': a['layer'] if a['layer'] <= K else 'final gather', 'compute_ms': round(a['compute_ms'] / reps, 3),
                     'wait_on_gather_ms': round(a['wait_on_gather_ms'] / reps, 3)} for a in tot]
            # the two halves alone, same ranks, same buffers: all SpMM launches without a gather, all gathers without an SpMM
            barrier()
            alone = sp.phase_times(eu, ei, K, reps=3, exact=args.exact)
            barrier()
            every = [None] * world
            dist.all_gather_object(every, (mine, alone))
            spmm_alone = [max(r[1]['spmm_alone_ms'][j] for r in every) for j in range(K)]
            gather_alone = [max(r[1]['allgather_alone_ms'][j] for r in every) for j in range(K)]
            longer = max(sum(spmm_alone), sum(gather_alone))
            result['layers'] = {
                'what': 'per layer, HIP events on the launch stream (3 extra forwards after the timed region): ms the stream spent in '
                        'its SpMM launches (both half-steps) and ms it sat waiting for an all-gathered block; then the two halves ALONE '
                        '(3 passes each): every SpMM launch of the layer with no gather issued, every all-gather the layer issues with '
                        'no SpMM launched (max over ranks)',
                'rank0': mine,
                'max_over_ranks': [{'layer': m['layer'], 'compute_ms': max(r[0][j]['compute_ms'] for r in every),
                                    'wait_on_gather_ms': max(r[0][j]['wait_on_gather_ms'] for r in every)} for j, m in enumerate(mine)],
                'spmm_alone_ms': spmm_alone, 'allgather_alone_ms': gather_alone,
                'overlap_efficiency': round(longer / (t / args.steps * 1e3), 4) if t > 0 else None,
                'overlap_efficiency_is': 'max(sum spmm_alone, sum allgather_alone) / measured ms_per_step: 1.0 = the shorter half is '
                                         'completely hidden under the longer one'}
        if world > 1:
            result['scaling'] = 'strong'      # fixed total work (config 4) split over the ranks
        result['config']['scaling_note'] = ('fixed total work (the same workload for every N, default BASELINE config 4) split over the '
                                            'ranks; N = 1 = `bench.py --gpus 1`, the whole graph on one GPU')

        # ---------------- second metric: scored pairs/s (every rank scores its own users)
        if not args.no_scoring:
            if args.shard == 'features':    # every rank holds the whole combined table; users are split evenly for scoring
                full = step()
                per = -(-n_u // world)
                users_all = np.arange(rank * per, min((rank + 1) * per, n_u))
                ue, ie = full[users_all[0]:users_all[-1] + 1], full[n_u:]
            else:
                ue, itab = sp.forward(eu, ei, K, exact=args.exact)
                ie = sp.items_in_order(itab)
                users_all = np.arange(*sp.user_range())
            mrp, mit = graph.train_mask()
            result['scoring'] = scoring_record(ue.contiguous(), ie.contiguous(), users_all, mrp, mit, n_i, d, dev, args, barrier, world=world,
                                               reduce_max_sum=reduce_max_sum, cpu=False, large=False)
        if args.shard == 'rows' and not args.no_user_partition:
            # the same forward under the USER partition (users never leave their rank; item rows are per-rank partial sums joined by
            # one all-reduce of the item table per layer): timed the same way, checked against the row partition's output (equal to
            # rounding: an item row is a sum of per-rank chains); the record's `value` stays the prescribed row partition's
            from textgcn_amd.dist import UserShardedPropagator
            ref_u, ref_tab = sp.forward(eu, ei, K, exact=args.exact, copy=True)
            ref_i = sp.items_in_order(ref_tab)
            del ref_tab
            up = UserShardedPropagator(graph, rank, world, dev, split_threshold=thr, chunks=chunks, force_collective=args.force_sharded)
            ueu, uei = up.local_e0(e0)

            def step_users():
                up.forward(ueu, uei, K, exact=args.exact)
            tu_dev, tu_wall = time_steps(step_users, args.steps, args.warmup, barrier)
            tu = reduce_max_sum([max(tu_wall, tu_dev)])[0][0]
            got_u, got_i = up.forward(ueu, uei, K, exact=args.exact)
            # the two partitions cut the users at the same nnz-balanced bounds: this rank's users are the same rows
            n_mine = up.u1 - up.u0
            err_i = float((got_i - ref_i).abs().max() / ref_i.abs().max())
            err_u = float((got_u[:n_mine] - ref_u[:n_mine]).abs().max() / ref_u[:n_mine].abs().max()) if n_mine else 0.0
            errs = reduce_max_sum([err_u, err_i])[0]
            up.record_events = True
            step_users()
            lt = up.layer_times()
            up.record_events = False
            every = [None] * world
            dist.all_gather_object(every, lt)
            result['user_partition'] = {
                'what': 'users partitioned (nnz-balanced), item table replicated: user rows local, item rows = per-rank partial sums joined '
                        'by ONE all-reduce of [I, d] per layer (dist.UserShardedPropagator); equal to the row partition to rounding',
                'value': args.steps * K * graph.nnz / tu, 'unit': 'edges/s', 'ms_per_step': tu / args.steps * 1e3,
                'bytes_exchanged_per_layer_per_rank': int(2 * (world - 1) / max(world, 1) * n_i * d * 4),
                'row_partition_bytes_received_per_layer_per_rank': int((world - 1) / max(world, 1) * graph.n * d * 4),
                'vs_row_partition': {'normwise_max_err_users': errs[0], 'normwise_max_err_items': errs[1], 'bar': 1e-4,
                                     'ok': bool(max(errs) <= 1e-4)},
                'layers_max_over_ranks': [{'layer': m['layer'], 'compute_ms': round(max(r[j]['compute_ms'] for r in every), 3),
                                           'wait_on_reduce_ms': round(max(r[j]['wait_on_reduce_ms'] for r in every), 3)}
                                          for j, m in enumerate(lt)]}
            del up, ueu, uei, ref_u, ref_i, got_u, got_i
            torch.cuda.empty_cache()
        if rank == 0:
            print(json.dumps(result), flush=True)
    finally:
        sp.close()
        try:
            dist.barrier()
        except Exception:
            pass
        if share_dir is not None and local_rank == 0:
            import shutil
            shutil.rmtree(share_dir, ignore_errors=True)
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default=None, help='c4 (default for every N), c2, c3, small, tiny')
    ap.add_argument('--exact', action='store_true', help='no long-row split: bit-identical to the CPU reference')
    ap.add_argument('--split-threshold', type=int, default=None)
    ap.add_argument('--no-segment', action='store_true', help='keep every row on the row-group kernel (no XCD-affine segments)')
    ap.add_argument('--score-batches', type=int, default=20)
    ap.add_argument('--score-batch-size', type=int, default=2048, help='users per scoring call (reference batch_size = 2048)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-scoring', action='store_true')
    ap.add_argument('--sub', default=None, help="comma list of sub-records at N = 1: c2,c3,c5,train (default: all for the default "
                                                "workload, none otherwise); 'none' to skip")
    ap.add_argument('--chunks', type=int, default=None, help='row chunks per block for the pipelined all-gather (N > 1)')
    ap.add_argument('--balance', default='nnz', choices=['nnz', 'rows'])
    ap.add_argument('--shard', default='rows', choices=['rows', 'features'],
                    help="N > 1: 'rows' = 1-D row partition with per-layer RCCL all-gathers (the north-star design, default); "
                         "'features' = every rank holds all rows and d/N columns: no per-layer exchange, one all-gather at the end")
    ap.add_argument('--no-user-partition', action='store_true',
                    help='N > 1, --shard rows: do not ALSO time the user partition (users local, one all-reduce of the item table per '
                         'layer: dist.UserShardedPropagator) in the same run')
    ap.add_argument('--feature-partition', action='store_true',
                    help='N > 1, --shard rows: ALSO time the forward under the feature (column) partition in the same run (a second '
                         'full-graph propagator per rank; off by default)')
    ap.add_argument('--streams-fp32', type=int, default=3, help='scoring calls in flight, fp32 filter path (sweeps)')
    ap.add_argument('--streams-prefilter', type=int, default=4, help='scoring calls in flight, bf16-candidate path (sweeps)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='N = 1 only: run the N > 1 code path (row blocks, chunked RCCL all-gathers on a 1-rank communicator, '
                         'per-rank scoring, all-reduced timing) on the one GPU -- a rehearsal of the multi-GPU run, not a headline')
    args = ap.parse_args()
    SCORE_STREAMS[False], SCORE_STREAMS[True] = args.streams_fp32, args.streams_prefilter

    # rehearsal switch for the one-GPU box: every rank on cuda:0, gloo staged through the host (tests only; the
    # driver's multi-GPU runs use RCCL, one GPU per rank).  Refused where more than one GPU is visible: there it could
    # only turn a real multi-GPU run into a silent one-GPU run.
    rehearsal = os.environ.get('TGCN_BENCH_REHEARSAL') == '1'
    n_visible = torch.cuda.device_count()        # counting does not initialise the GPU (no exec hazard, nothing allocated)
    if rehearsal and n_visible > 1:
        raise SystemExit(f'TGCN_BENCH_REHEARSAL=1 (all ranks on cuda:0 over gloo) is for one-GPU boxes; {n_visible} GPUs are visible here -- unset it')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # started plainly (the form the driver's N = 1 command has): become the launcher
        if not rehearsal and n_visible < args.gpus:
            raise SystemExit(f'--gpus {args.gpus} needs {args.gpus} visible GPUs (one per rank), found {n_visible} '
                             '(one-GPU rehearsal of the N > 1 path: TGCN_BENCH_REHEARSAL=1)')
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if world > 1 and int(os.environ.get('LOCAL_WORLD_SIZE', world)) != world:
        raise SystemExit('bench.py is a single-node benchmark (the graph is shared through node-local memory): LOCAL_WORLD_SIZE != WORLD_SIZE')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP path has no CPU fallback')
    if world > 1 and not rehearsal and n_visible < world:
        raise SystemExit(f'--gpus {world} needs {world} visible GPUs (one per rank), found {n_visible}')
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if world == 1 and 'RANK' not in os.environ:      # --force-sharded from a plain `python bench.py`
            import tempfile
            rdv = tempfile.mkdtemp(prefix='tgcn_bench_rdv_')     # a file store: no TCP port to collide on
            dist.init_process_group(backend='nccl', init_method=f'file://{rdv}/store', rank=0, world_size=1, device_id=dev)
        elif rehearsal:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=dev)
        run_sharded(args, world, rank, local_rank, dev, dev_index, rehearsal)
        return

    # ---------------- N = 1: the headline workload (config 4 unless --workload), then the other single-GPU configurations
    default_wl = args.workload is None
    wl = args.workload or 'c4'
    cpu = not args.no_cpu_baseline
    result = record_single_gpu(wl, dev, args, args.steps, args.warmup, cpu=cpu)
    sub = args.sub if args.sub is not None else ('c2,c3,c5,train' if default_wl else 'none')
    sub = [] if sub == 'none' else [s.strip() for s in sub.split(',') if s.strip()]
    if 'c2' in sub and wl != 'c2':
        result['c2'] = record_single_gpu('c2', dev, args, 50, 5, cpu=cpu, train='train' in sub)
    if 'c3' in sub or 'c5' in sub:
        result.update(records_c3_c5(dev, want_c5='c5' in sub, cpu=cpu))
        torch.cuda.empty_cache()
    print(json.dumps(result), flush=True)


if __name__ == '__main__':
    main()
