"""Per-rank CPU affinity for the one-process-per-GPU layout (the counterpart of the worker placement
``MultiprocessParallelUpdater`` leaves to the OS, train.py:117-121).

Each rank's host threads (the Python enqueue loop, the loader threads, RCCL's proxy thread) are pinned to cores of the NUMA
node its GPU hangs off, and ranks that share a node split its cores - an 8-GPU MI355X host has two sockets with four GPUs
each, and a rank whose enqueue thread migrates to the far socket pays cross-socket latency on every doorbell write.
Everything here reads sysfs only (no GPU runtime call); on any failure the allowed set is split evenly by local rank.
"""
import os


def _cpulist(s):
    out = []
    for part in s.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _gpu_numa_nodes():
    """NUMA node of every KFD GPU node in enumeration order (= HIP device order without HIP_VISIBLE_DEVICES), from
    /sys/class/kfd/kfd/topology/nodes/*/properties (simd_count > 0 marks a GPU) -> its PCI device's numa_node."""
    base = '/sys/class/kfd/kfd/topology/nodes'
    nodes = []
    for n in sorted(os.listdir(base), key=int):
        props = dict(l.split() for l in open(os.path.join(base, n, 'properties')) if len(l.split()) == 2)
        if int(props.get('simd_count', 0)) == 0:
            continue
        dom, loc = int(props.get('domain', 0)), int(props['location_id'])
        bdf = '%04x:%02x:%02x.%d' % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        try:
            nodes.append(int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read()))
        except (OSError, ValueError):
            nodes.append(-1)
    return nodes


def rank_cpus(local_rank, n_local, allowed=None):
    """The cores rank ``local_rank`` of ``n_local`` should run on (a non-empty subset of ``allowed``)."""
    allowed = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    n_local = max(1, n_local)
    local_rank = local_rank % n_local
    try:
        nodes = _gpu_numa_nodes()
        visible = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')
        if visible:
            toks = [v.strip() for v in visible.split(',') if v.strip()]
            if not all(v.isdigit() for v in toks):        # UUIDs ("GPU-..."): the index -> node map is unknown, split evenly
                raise ValueError('non-numeric visible-device list')
            nodes = [nodes[int(v)] for v in toks]
        node = nodes[local_rank]
        if node >= 0:
            cpus = [c for c in _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if c in set(allowed)]
            peers = [r for r in range(min(n_local, len(nodes))) if nodes[r] == node]
            if cpus and local_rank in peers:
                k = peers.index(local_rank)
                share = cpus[k * len(cpus) // len(peers):(k + 1) * len(cpus) // len(peers)]
                if share:
                    return share
    except Exception:
        pass
    share = allowed[local_rank * len(allowed) // n_local:(local_rank + 1) * len(allowed) // n_local]
    return share or allowed


def pin_rank(local_rank, n_local):
    """os.sched_setaffinity for this process; returns the core list (for the log).  MRCNN_NO_AFFINITY=1 leaves the OS alone."""
    if os.environ.get('MRCNN_NO_AFFINITY') == '1' or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = rank_cpus(local_rank, n_local)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return cpus
