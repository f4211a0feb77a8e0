#!/usr/bin/env python3
"""Benchmark of the MI355X LightGCN propagation + scoring path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|small] ...
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...      (N > 1, one rank per GPU)

One "step" = one K-layer propagation (the `representation` forward, TextGCN/base_model.py:93-106) over the
synthetic graph, inputs resident in HBM.  Headline value = propagated directed edges per second
= steps * K * nnz(A) / t  (SURVEY.md §8d metric 1), whole job.  The second metric of BASELINE.json (scored
user-item pairs/s: dense scores + train mask + top-40 per batch of 2048 users) is timed in a separate region
and reported under "scoring" in the same JSON line.

N = 1 runs BASELINE config 2 (U=100k, I=50k, nnz=5M, d=64, K=3).  N > 1 runs the same graph family scaled
with N (U=100k*N, I=50k*N, nnz=5M*N: fixed work per GPU -> "weak"), row-sharded with one RCCL all-gather of
the propagated user block and one of the item block per layer.

roofline: HBM-bound SpMM.  `achieved` = ALGORITHMIC (compulsory) bytes of one layer launch / its mean duration,
bytes = nnz*8 + (rows+1)*4 + n_src*d*4 + rows*d*4 + fused layer-sum traffic; `traffic` = HBM
bytes per layer launch from the rocprofv3 PMC pass committed under profiles/ (null when no profile matches).
cpu_baseline: the torch CPU calls the reference makes (torch.sparse.mm on the coalesced COO x K, stack+mean),
timed on this host on a bounded sample (kind "port": oracle/torch_port.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32 matrix peak (v_mfma_f32_32x32x2_f32)


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


def load_traffic(workload):
    """HBM bytes per layer launch measured by rocprofv3 --pmc (profiles/hbm_traffic.json), or None."""
    p = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload, {}).get('hbm_bytes_per_layer')
        except Exception:
            return None
    return None


def cpu_baseline_propagation(graph, e0, n_layers, budget_s=12.0):
    """The reference's own torch calls on this host's CPU (oracle/torch_port.py), bounded sample."""
    from oracle import torch_port
    idx, val = graph.to_coo()
    a = torch_port.norm_matrix(idx, val, graph.n)
    threads = torch.get_num_threads()
    # sample: whole forwards while the budget lasts (at least one)
    t0 = time.perf_counter()
    n_fwd = 0
    while True:
        torch_port.representation(a, e0, n_layers)
        n_fwd += 1
        el = time.perf_counter() - t0
        if el > budget_s or n_fwd >= 5:
            break
    return {'value': n_fwd * n_layers * graph.nnz / el, 'unit': 'edges/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n_fwd} full {n_layers}-layer CPU forward(s) of the same graph in {el:.1f} s; '
                      f'torch {torch.__version__} sparse.mm on coalesced COO (single-threaded kernel) + stack/mean, '
                      f'{threads} torch threads on {os.cpu_count()} host cpus'}


def cpu_baseline_scoring(users_emb, items_emb, mask_rowptr, mask_items, k, budget_s=8.0):
    from oracle import torch_port
    b = users_emb.shape[0]
    t0 = time.perf_counter()
    n = 0
    while True:
        torch_port.score_mask_topk(users_emb, items_emb, mask_rowptr, mask_items, k)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 5:
            break
    return {'value': n * b * items_emb.shape[0] / el, 'unit': 'pairs/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} batch(es) of {b} users x {items_emb.shape[0]} items: torch.matmul + -inf mask + topk({k}) in {el:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default=None, help='c2 (default), c3, c4, small, tiny')
    ap.add_argument('--exact', action='store_true', help='no long-row split: bit-identical to the CPU reference')
    ap.add_argument('--split-threshold', type=int, default=None)
    ap.add_argument('--no-segment', action='store_true', help='keep every row on the one-wave-per-row kernel (no XCD-affine segments)')
    ap.add_argument('--score-batches', type=int, default=20)
    ap.add_argument('--score-batch-size', type=int, default=2048, help='users per scoring call (reference batch_size = 2048)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-scoring', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N')
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP path has no CPU fallback')
    # rehearsal switch for the one-GPU box: every rank on cuda:0, gloo staged through the host (tests only; the
    # driver's multi-GPU runs use RCCL, one GPU per rank)
    rehearsal = os.environ.get('TGCN_BENCH_REHEARSAL') == '1'
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend='gloo')
        else:
            dist.init_process_group(backend='nccl', device_id=dev)

    from textgcn_amd import propagate, scoring, synth
    from textgcn_amd.graph import NormGraph, train_mask_csr

    wl = args.workload or 'c2'
    n_u, n_i, nnz, d, K = synth.CONFIGS[wl]
    if world > 1 and args.workload is None:
        n_u, n_i, nnz = n_u * world, n_i * world, nnz * world   # weak scaling: config 2 per GPU
        wl_name = f'c2 x {world} (weak): U={n_u} I={n_i} nnz={nnz} d={d} K={K}'
    else:
        wl_name = f'{wl}: U={n_u} I={n_i} nnz={nnz} d={d} K={K}'
    t0 = time.time()
    u, i = synth.interactions(n_u, n_i, nnz, seed=0)
    graph = NormGraph.from_pairs(u, i, n_u, n_i)
    e0 = synth.embeddings(graph.n, d, seed=0)
    build_s = time.time() - t0
    thr = args.split_threshold or propagate.DEFAULT_SPLIT_THRESHOLD

    if world == 1:
        prop = propagate.Propagator(graph, dev, split_threshold=thr, segment=None if args.no_segment else 'auto')
        e0d = e0.to(dev)
        out = torch.empty_like(e0d)

        def step():
            prop.forward(e0d, K, exact=args.exact, out=out)
        n_rows_local, n_src, nnz_local = graph.n, graph.n, graph.nnz
    else:
        from textgcn_amd.dist import ShardedPropagator
        sp = ShardedPropagator(graph, rank, world, dev, split_threshold=thr)
        eu, ei = sp.local_e0(e0)

        def step():
            sp.forward(eu, ei, K, exact=args.exact)
        n_rows_local, n_src, nnz_local = sp.bu + sp.bi, sp.n_pad, sp.nnz_local

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def reduce_max_sum(vals):
        """(max over ranks, sum over ranks) of a small list of floats"""
        t = torch.tensor(vals, dtype=torch.float64, device='cpu' if rehearsal else dev)
        mx, sm = t.clone(), t.clone()
        torch.distributed.all_reduce(mx, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(sm, op=torch.distributed.ReduceOp.SUM)
        return mx.tolist(), sm.tolist()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t_wall0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    t_wall = time.perf_counter() - t_wall0
    t_dev = ev0.elapsed_time(ev1) / 1e3
    t = max(t_wall, t_dev)
    if world > 1:
        t = reduce_max_sum([t])[0][0]
    edges = args.steps * K * graph.nnz
    value = edges / t
    seg_note = 'none'
    if world == 1 and not args.exact and prop.csr.segment_blocks and any(prop.csr.segment_blocks):
        seg_note = (f'user rows x{prop.csr.segment_blocks[0]}, item rows x{prop.csr.segment_blocks[1]} column blocks, '
                    f'{prop.csr.segment_tile}-entry tiles (tgcn_spmm_segmented_f32)')

    # ---------------- roofline of the dominant kernel (one SpMM layer launch on this rank)
    layer_bytes = [algorithmic_bytes_per_layer(nnz_local, n_rows_local, n_src, d, k, K, False) for k in range(1, K + 1)]
    mean_layer_s = t_dev / (args.steps * K)
    achieved = float(np.mean(layer_bytes)) / mean_layer_s / 1e9
    # the PMC pass was taken on the default path of the workload (profiles/hbm_traffic.json)
    traffic = load_traffic(wl if world == 1 and not args.exact and not args.no_segment else None)
    roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                # measured memory-side bytes (PMC pass, profiles/) over the live launch time: what the fabric moves
                'traffic_GBs': round(traffic / mean_layer_s / 1e9, 1) if traffic else None,
                'kernel': 'one SpMM layer (k_spmm_seg + k_spmm_seg_reduce, or k_spmm_wave + k_spmm_long_reduce)',
                'algorithmic_bytes_per_launch': int(np.mean(layer_bytes)), 'launch_us': round(mean_layer_s * 1e6, 2),
                'gather_model_GBs': round((nnz_local * (8 + 4 * d) + n_rows_local * d * 4) / mean_layer_s / 1e9, 1)}

    result = {
        'metric': 'propagated edges/sec (3-layer SpMM, d=64)', 'value': value, 'unit': 'edges/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': wl_name, 'nnz_A': graph.nnz, 'n_nodes': graph.n, 'max_degree': int(graph.degrees().max()),
                   'mode': 'exact (one fmaf chain per row)' if args.exact else f'one-wave-per-row kernel: rows > {thr} entries split in chunks',
                   'xcd_segments': seg_note,
                   'sharding': 'none' if world == 1 else f'row-sharded x{world}, RCCL all-gather per layer (users ∥ item half-step)',
                   'graph_build_s': round(build_s, 1)},
        'roofline': roofline,
    }

    # ---------------- second metric: scored pairs/s (rank 0's users; every rank scores its own users)
    if not args.no_scoring:
        k_top = 40
        bsz = args.score_batch_size
        if world == 1:
            ue, ie = out[:n_u], out[n_u:]
            users_all = np.arange(n_u)
        else:
            ue, ie = sp.forward(eu, ei, K, exact=args.exact)
            ie = ie[:n_i]
            users_all = np.arange(*sp._user_range(rank))
        mrp, mit = train_mask_csr(u, i, n_u)
        n_batches = min(args.score_batches, max(1, len(users_all) // bsz))
        batches = []
        for bidx in range(n_batches):
            bu_ = users_all[bidx * bsz:(bidx + 1) * bsz]
            rp = (mrp[bu_ + 1] - mrp[bu_])
            rowptr = np.zeros(len(bu_) + 1, dtype=np.int32)
            np.cumsum(rp, out=rowptr[1:])
            items = np.concatenate([mit[mrp[x]:mrp[x + 1]] for x in bu_])
            ids = torch.from_numpy(bu_ - users_all[0]).to(dev)
            batches.append((ids, torch.from_numpy(rowptr).to(dev), torch.from_numpy(items).to(dev)))
        ue = ue.contiguous()
        ie = ie.contiguous()

        # consecutive calls are independent: issued round-robin on three HIP streams with their own scratch buffers, as
        # LightGCN.predict does, so one call's small selection kernels run under the next call's GEMM
        main = torch.cuda.current_stream(dev)
        side = [torch.cuda.Stream(dev) for _ in range(3)]

        def score_all(bts):   # the predict step of base_model.py:254-263, fused (tgcn_score_topk_f32)
            for st in side:
                st.wait_stream(main)
            keep = []
            for j, (ids, rp, it) in enumerate(bts):
                with torch.cuda.stream(side[j % 3]):
                    keep.append(scoring.score_topk(ue, ie, k_top, user_ids=ids, mask_rowptr=rp, mask_items=it, round4=True,
                                                   slot=j % 3))
            for st in side:
                main.wait_stream(st)
            return keep
        score_all(batches[:3])
        barrier()
        ev0.record()
        score_all(batches)
        ev1.record()
        barrier()
        ts = ev0.elapsed_time(ev1) / 1e3
        pairs = sum(int(bt[0].numel()) for bt in batches) * n_i
        if world > 1:
            mx, sm = reduce_max_sum([ts, float(pairs)])
            ts, pairs = mx[0], sm[1]
        flops = 2.0 * d * pairs
        result['scoring'] = {
            'metric': f'scored user-item pairs/sec (scores + train mask + top-40 fused, B={bsz} per call, 3 streams)', 'value': pairs / ts,
            'unit': 'pairs/s', 'batches': n_batches, 'ms_per_batch': ts / n_batches * 1e3,
            'roofline': {'bound': 'mfma', 'achieved': round(flops / ts / 1e12 / max(world, 1), 2), 'peak': MFMA_F32_PEAK_TF,
                         'unit': 'TFLOP/s', 'frac': round(flops / ts / 1e12 / max(world, 1) / MFMA_F32_PEAK_TF, 4), 'traffic': None},
        }
        # the model class scores 16384 users per call (LightGCN.predict_chunk); same kernels, fewer launches
        big = min(16384, len(users_all))
        if world == 1 and big > bsz:
            n_big = max(1, min(4, len(users_all) // big))
            bb = []
            for bidx in range(n_big):
                bu_ = users_all[bidx * big:(bidx + 1) * big]
                rowptr = np.zeros(len(bu_) + 1, dtype=np.int32)
                np.cumsum(mrp[bu_ + 1] - mrp[bu_], out=rowptr[1:])
                items = mit[mrp[bu_[0]]:mrp[bu_[-1] + 1]]
                bb.append((torch.from_numpy(bu_).to(dev), torch.from_numpy(rowptr).to(dev), torch.from_numpy(np.ascontiguousarray(items)).to(dev)))
            score_all(bb[:1])
            barrier()
            ev0.record()
            score_all(bb)
            ev1.record()
            barrier()
            tb = ev0.elapsed_time(ev1) / 1e3
            pb = sum(int(bt[0].numel()) for bt in bb) * n_i
            result['scoring']['large_batch'] = {'users_per_call': big, 'value': pb / tb, 'unit': 'pairs/s',
                                                'ms_per_call': tb / n_big * 1e3,
                                                'mfma_frac': round(2.0 * d * pb / tb / 1e12 / MFMA_F32_PEAK_TF, 4)}

    # ---------------- CPU baseline beside it (rank 0, N = 1 only)
    if world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline_propagation(graph, e0, K)
        if not args.no_scoring:
            bt = batches[0]
            result['scoring']['cpu_baseline'] = cpu_baseline_scoring(
                ue[bt[0]].cpu(), ie.cpu(), bt[1].cpu().numpy(), bt[2].cpu().numpy(), 40)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
