#!/usr/bin/env python3
"""CPU rehearsal of what `bench.py --gpus N` does on the host before any kernel runs: N ranks over gloo,
rank 0 builds the workload once and publishes it (bench.shared_workload), every rank maps it and cuts its own row blocks
(ShardedPropagator's host-side construction, device 'cpu').  Prints wall time and peak RSS per rank -- the numbers that decide
whether 8 ranks fit the driver's time limit and the node's memory.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29577 \
        tools/shared_build_rehearsal.py --workload c4
"""
import argparse
import json
import os
import resource
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c4')
    ap.add_argument('--chunks', type=int, default=4)
    args = ap.parse_args()
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    import bench
    from textgcn_amd.dist import ShardedPropagator
    t0 = time.time()
    graph, e0, path = bench.shared_workload(args.workload, rank, int(os.environ.get('LOCAL_RANK', rank)), dist.barrier)
    t_share = time.time() - t0
    sp = ShardedPropagator(graph, rank, world, 'cpu', local_spmm=lambda *a, **k: None, split_threshold=1024, chunks=args.chunks)
    eu, ei = sp.local_e0(e0)
    t_all = time.time() - t0
    rec = {'rank': rank, 'world': world, 'workload': args.workload, 'shared_build_s': round(t_share, 1), 'with_row_blocks_s': round(t_all, 1),
           'peak_rss_GB': round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2), 'nnz_local': sp.nnz_local,
           'e0_local_rows': int(eu.shape[0] + ei.shape[0])}
    every = [None] * world
    dist.all_gather_object(every, rec)
    dist.barrier()
    if rank == 0:
        import shutil
        shm = sum(os.path.getsize(os.path.join(path, f)) for f in os.listdir(path)) / 1e9
        shutil.rmtree(path, ignore_errors=True)
        print(json.dumps({'shared_files_GB': round(shm, 2), 'ranks': every}))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
