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
