#!/usr/bin/env python3
"""One rank of a row-sharded propagation rehearsal (started by tests/test_dist_*.py, never by pytest directly).

  --mode cpu : gloo on CPU tensors; the local SpMM is the CPU oracle (injected stand-in) -- exercises the
               partitioning / padding / all-gather / layer-sum logic of textgcn_amd.dist without a GPU.
  --mode gpu : every rank uses cuda:0 (one-GPU box) with the real HIP kernels; gloo staged through the host.
  --mode nccl: backend nccl (= RCCL).  ONE rank on cuda:0 with the world == 1 short-circuit bypassed, so the production
               collective code runs -- in-place all_gather_into_tensor(async_op=True) + work.wait() on torch's communicator
               (--collective torch) or tgcn_allgather_rows on libtgcn's own communicator and side stream (--collective capi);
               with --device-per-rank, --world N ranks on cuda:0..N-1 of a multi-GPU box (the real thing).
Writes users_full / items_full of rank 0 (after gathering the user blocks) to --out.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_spmm(csr, x, y=None, acc_in=None, acc_out=None, acc_div=1.0, exact=False):
    from oracle import lgcn_oracle as orc
    out = orc.spmm_csr(csr.rowptr.numpy(), csr.colidx.numpy(), csr.vals.numpy(), x.numpy())
    if y is not None:
        y.copy_(torch.from_numpy(out))
    if acc_out is not None:
        t = (acc_in.numpy() + out).astype(np.float32)
        if acc_div != 1.0:
            t = (t / np.float32(acc_div)).astype(np.float32)
        acc_out.copy_(torch.from_numpy(t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rank', type=int, required=True)
    ap.add_argument('--world', type=int, required=True)
    ap.add_argument('--port', type=int, required=True)
    ap.add_argument('--mode', choices=['cpu', 'gpu', 'nccl'], required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--n-users', type=int, default=203)
    ap.add_argument('--n-items', type=int, default=97)
    ap.add_argument('--nnz', type=int, default=2500)
    ap.add_argument('--d', type=int, default=64)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--single', action='store_true')
    ap.add_argument('--balance', default='nnz')
    ap.add_argument('--chunks', type=int, default=1)
    ap.add_argument('--collective', default='torch')
    ap.add_argument('--sample', type=int, default=0, help='save only this many seeded sample rows of each table (large graphs)')
    ap.add_argument('--graph-seed', type=int, default=1)
    ap.add_argument('--shard', choices=['rows', 'features', 'users'], default='rows')
    ap.add_argument('--device-per-rank', action='store_true', help='--mode nccl on a multi-GPU box: rank r uses cuda:r')
    ap.add_argument('--split-threshold', type=int, default=64)
    ap.add_argument('--exact', action='store_true')
    ap.add_argument('--no-segment', action='store_true', help='keep every row chunk on the row-group kernel (bit-identity against segment=None)')
    args = ap.parse_args()

    from textgcn_amd import synth
    from textgcn_amd.dist import ShardedPropagator
    from textgcn_amd.graph import NormGraph

    if args.mode == 'nccl':
        dev_index = args.rank if args.device_per_rank else 0
        torch.cuda.set_device(dev_index)
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{args.port}', rank=args.rank, world_size=args.world,
                                device_id=torch.device('cuda', dev_index))
    else:
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{args.port}', rank=args.rank, world_size=args.world)
    u, i = synth.interactions(args.n_users, args.n_items, args.nnz, seed=args.graph_seed)
    g = NormGraph.from_pairs(u, i, args.n_users, args.n_items)
    e0 = synth.embeddings(g.n, args.d, seed=2)
    if args.shard == 'features':      # column partition: no per-layer exchange, one all-gather at the end
        from textgcn_amd.dist import ColumnShardedPropagator
        dev = 'cuda:0' if args.mode != 'nccl' else 'cuda'
        cp = ColumnShardedPropagator(g, args.d, args.rank, args.world, dev, split_threshold=64, force_collective=(args.mode == 'nccl'))
        full = cp.assemble(cp.forward(cp.local_e0(e0), args.layers, single=args.single))
        if args.rank == 0:
            np.savez(args.out, users=full[:args.n_users].cpu().numpy(), items=full[args.n_users:].cpu().numpy())
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.shard == 'users':         # user partition: local user rows, partial item sums, one all-reduce of the item table per layer
        from textgcn_amd.dist import UserShardedPropagator
        if args.mode == 'cpu':
            up = UserShardedPropagator(g, args.rank, args.world, 'cpu', local_spmm=oracle_spmm, split_threshold=None, chunks=args.chunks)
        else:
            up = UserShardedPropagator(g, args.rank, args.world, 'cuda:0' if args.mode != 'nccl' else 'cuda', split_threshold=args.split_threshold,
                                       chunks=args.chunks, force_collective=(args.mode == 'nccl'))
        eu, ei = up.local_e0(e0)
        users_local, items = up.forward(eu, ei, args.layers, single=args.single, exact=(args.mode == 'cpu' or args.exact))
        users_full = up.gather_users(users_local)
        if args.rank == 0:
            np.savez(args.out, users=users_full.cpu().numpy(), items=items.cpu().numpy(), user_bounds=up.bounds, nnz_local=up.nnz_local)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.mode == 'cpu':
        sp = ShardedPropagator(g, args.rank, args.world, 'cpu', local_spmm=oracle_spmm, split_threshold=None,
                               balance=args.balance, chunks=args.chunks)
    elif args.mode == 'nccl':
        sp = ShardedPropagator(g, args.rank, args.world, 'cuda', split_threshold=64, balance=args.balance, chunks=args.chunks,
                               collective=args.collective, force_collective=True)
        assert sp.backend == 'nccl' and sp.uses_collective and (sp._capi_comm is not None) == (args.collective == 'capi')
    else:
        sp = ShardedPropagator(g, args.rank, args.world, 'cuda:0', split_threshold=args.split_threshold, balance=args.balance,
                               chunks=args.chunks, segment=None if args.no_segment else 'auto')
    eu, ei = sp.local_e0(e0)
    users_local, items_full = sp.forward(eu, ei, args.layers, single=args.single, exact=(args.mode == 'cpu' or args.exact))
    if args.sample:      # large graphs: seeded sample rows of rank 0's own users and of the gathered item table
        rng = np.random.default_rng(123)
        u0, u1 = sp.user_range()
        su = np.sort(rng.choice(u1 - u0, size=min(args.sample, u1 - u0), replace=False))
        si = np.sort(rng.choice(args.n_items, size=min(args.sample, args.n_items), replace=False))
        if args.rank == 0:
            dev = users_local.device
            items_rows = items_full[torch.from_numpy(sp.lay_i.table_rows(si)).to(dev)]
            np.savez(args.out, users=users_local[torch.from_numpy(su).to(dev)].cpu().numpy(), items=items_rows.cpu().numpy(),
                     user_rows=su + u0, item_rows=si, nnz_local=sp.nnz_local, user_bounds=sp.lay_u.bounds, item_bounds=sp.lay_i.bounds)
    else:
        users_full = sp.gather_users(users_local)
        if args.rank == 0:
            np.savez(args.out, users=users_full.cpu().numpy(), items=sp.items_in_order(items_full).cpu().numpy(),
                     nnz_local=sp.nnz_local, user_bounds=sp.lay_u.bounds, item_bounds=sp.lay_i.bounds,
                     segment_note=np.array(sp.segment_note() if hasattr(sp, 'segment_note') else 'none'))
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
