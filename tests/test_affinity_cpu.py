"""Host logic of the one-process-per-GPU layout: utils/affinity.rank_cpus - which cores a rank's host threads
(enqueue loop, loader workers, RCCL proxy) are pinned to.  Pure sysfs / arithmetic: runs without a GPU (the sysfs reads fail here and
the even split is taken; the NUMA branch is driven through a stubbed topology)."""
import os

from chainer_maskrcnn.utils import affinity


def test_even_split_partitions_the_allowed_set():
    allowed = list(range(3, 35))                     # 32 cores, not starting at 0
    shares = [affinity.rank_cpus(r, 8, allowed=allowed) for r in range(8)]
    assert all(shares) and sorted(c for s in shares for c in s) == allowed          # a partition: every core once, no rank empty
    assert all(len(s) == 4 for s in shares)
    # more ranks than cores: nobody gets an empty set (falls back to the whole allowed set)
    assert all(affinity.rank_cpus(r, 8, allowed=[0, 1, 2]) for r in range(8))
    # local_rank beyond n_local wraps instead of indexing out of range
    assert affinity.rank_cpus(9, 8, allowed=allowed) == shares[1]


def test_cpulist_parser():
    assert affinity._cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert affinity._cpulist('') == []


def test_numa_branch_and_uuid_visible_devices(monkeypatch, tmp_path):
    """Two GPUs per NUMA node: ranks on the same node split ITS cores; a visible-device list of UUIDs (index -> node map unknown) or any
    sysfs failure falls back to the even split."""
    allowed = list(range(16))
    monkeypatch.setattr(affinity, '_gpu_numa_nodes', lambda: [0, 0, 1, 1])
    real_open = open

    def fake_open(path, *a, **k):
        if str(path).startswith('/sys/devices/system/node/node'):
            node = int(str(path).split('node')[-1].split('/')[0])
            p = tmp_path / ('n%d' % node)
            p.write_text('0-7\n' if node == 0 else '8-15\n')
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr('builtins.open', fake_open)
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    assert affinity.rank_cpus(0, 4, allowed=allowed) == [0, 1, 2, 3]
    assert affinity.rank_cpus(1, 4, allowed=allowed) == [4, 5, 6, 7]
    assert affinity.rank_cpus(2, 4, allowed=allowed) == [8, 9, 10, 11]
    assert affinity.rank_cpus(3, 4, allowed=allowed) == [12, 13, 14, 15]
    # numeric visible list reorders the devices: rank 0 -> device 2 -> node 1
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '2,0')
    assert affinity.rank_cpus(0, 2, allowed=allowed) == list(range(8, 16))
    assert affinity.rank_cpus(1, 2, allowed=allowed) == list(range(0, 8))
    # UUIDs: even split of the allowed set
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', 'GPU-abc,GPU-def')
    assert affinity.rank_cpus(0, 2, allowed=allowed) == list(range(0, 8))
    assert affinity.rank_cpus(1, 2, allowed=allowed) == list(range(8, 16))


def test_pin_rank_can_be_switched_off(monkeypatch):
    monkeypatch.setenv('MRCNN_NO_AFFINITY', '1')
    before = os.sched_getaffinity(0)
    assert affinity.pin_rank(0, 2) is None
    assert os.sched_getaffinity(0) == before
