"""CPU tests (gloo, world_size 2) of the data-parallel gradient exchange: the bucketed all-reduce of the flat
gradient buffer (chainer_maskrcnn/optimizers.py, the MI355X-native form of the MultiprocessParallelUpdater
wiring of train.py:117-121).  The HIP kernels are not involved: this pins bucketing, ordering and the SUM
semantics (un-scaled learning rate) that the RCCL path uses unchanged."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chainer_maskrcnn.optimizers import GradientSynchronizer


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, bucket_bytes, average, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        sync = GradientSynchronizer(g, bucket_bytes=bucket_bytes, average=average)
        # buckets tile the buffer exactly, from the end towards the start
        ends = [b[1] for b in sync.buckets]
        starts = [b[0] for b in sync.buckets]
        assert ends[0] == n and starts[-1] == 0 and all(starts[i] == ends[i + 1] for i in range(len(ends) - 1))
        sync.begin()
        launched = []
        for off in (n, (3 * n) // 4, n // 2, n // 3, 0):          # backward progresses from the last layer to the first
            sync.mark_ready(off)
            launched.append(sync.next)
            assert all(b[0] >= off for b in sync.buckets[:sync.next])      # never touches unfinished gradients
        sync.finish()
        assert launched == sorted(launched) and sync.next == len(sync.buckets)
        q.put((rank, g.clone(), len(sync.buckets)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n,bucket_bytes,average', [(100003, 64 << 10, False), (4096, 1 << 20, False), (50000, 32 << 10, True)])
def test_bucketed_allreduce_two_ranks(n, bucket_bytes, average):
    ctx = mp.get_context('spawn')
    outs = None
    for attempt in range(3):       # the rendezvous port is picked, released and re-bound by rank 0: retried if something else took it
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bucket_bytes, average, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            got = [q.get(timeout=120) for _ in procs]
        except Exception:
            got = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if got is not None and all(p.exitcode == 0 for p in procs):
            outs = got
            break
    assert outs is not None, 'two-rank gloo group failed three times'
    want = torch.arange(n, dtype=torch.float32) * 3            # rank0 (x1) + rank1 (x2): SUM, like the reference
    if average:
        want = want / 2
    for rank, g, nb in outs:
        assert torch.equal(g, want), rank
        assert nb == max(1, -(-n * 4 // bucket_bytes))


def test_single_process_is_a_no_op():
    g = torch.ones(1000)
    sync = GradientSynchronizer(g, bucket_bytes=1024)
    sync.begin(); sync.mark_ready(0); sync.finish()
    assert torch.equal(g, torch.ones(1000))


def test_after_bucket_walks_every_bucket_once_without_a_process_group():
    """The optimizer's sectioned update (MomentumSGD.sectioned_update) rides on the bucket walk: with a callback the buckets are
    visited even when there is nothing to all-reduce (one rank), each exactly once, from the end of the buffer, never above the
    offset the backward pass has reported; bucket starts are 64-float aligned (the float4 update kernel runs on a bucket)."""
    n = 100003
    g = torch.arange(n, dtype=torch.float32)
    sync = GradientSynchronizer(g, bucket_bytes=64 << 10)
    seen = []
    sync.after_bucket = lambda s, e: seen.append((s, e))
    sync.begin()
    for off in (n, 70000, 70000, 33333, 1):
        sync.mark_ready(off)
        assert all(s >= off for s, _ in seen)
    assert seen and seen[-1][0] > 0
    sync.finish()
    assert seen == sync.buckets
    assert seen[0][1] == n and seen[-1][0] == 0 and all(seen[i][0] == seen[i + 1][1] for i in range(len(seen) - 1))
    assert all(s % 64 == 0 for s, _ in seen)
    assert torch.equal(g, torch.arange(n, dtype=torch.float32))         # nothing was reduced
    sync.after_bucket = None
    sync.begin(); sync.mark_ready(0); sync.finish()
    assert len(seen) == len(sync.buckets)                                # without a callback a one-rank walk is a no-op again


def _worker_after_bucket(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        p = torch.zeros(n)
        sync = GradientSynchronizer(g, bucket_bytes=32 << 10)

        def update(s, e):               # the update sees the SUMMED gradient of its bucket
            p[s:e] -= 0.5 * g[s:e]
        sync.after_bucket = update
        sync.begin()
        for off in (n // 2, 0):
            sync.mark_ready(off)
        sync.finish()
        q.put((rank, p))
    finally:
        dist.destroy_process_group()


def test_after_bucket_runs_behind_the_all_reduce_two_ranks():
    n = 50000
    ctx = mp.get_context('spawn')
    outs = None
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_after_bucket, args=(r, 2, port, n, q)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            got = [q.get(timeout=120) for _ in procs]
        except Exception:
            got = None
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if got is not None and all(p.exitcode == 0 for p in procs):
            outs = got
            break
    assert outs is not None, 'two-rank gloo group failed three times'
    want = -0.5 * 3 * torch.arange(n, dtype=torch.float32)
    for rank, p in outs:
        assert torch.equal(p, want), rank
