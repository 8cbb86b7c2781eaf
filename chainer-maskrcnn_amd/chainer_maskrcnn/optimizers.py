"""MomentumSGD + WeightDecay over the flat parameter buffer, and the data-parallel gradient exchange.

Mirror of the optimizer wiring of train.py:107-109 (``MomentumSGD(lr, 0.9)``, ``setup(model)``,
``add_hook(WeightDecay(0.0005))``) and of the multi-GPU updater of train.py:117-121
(``MultiprocessParallelUpdater``: per-step NCCL reduce of ONE flat gradient buffer + bcast of the
parameters).  MI355X-native form: every rank keeps a replica and all-reduces
(SUM, un-scaled learning rate = the reference's semantics) contiguous buckets of the flat gradient
buffer over RCCL on a side HIP stream while backward is still producing the earlier layers'
gradients; every rank then applies the same fused update (no broadcast).
"""
import torch

from chainer_maskrcnn._hip import ops


def init_process_group(backend='nccl', **kw):
    """torch.distributed.init_process_group for the one-process-per-GPU layout.  With RCCL (backend 'nccl') the collectives'
    internal streams are created at HIGH priority: the training step runs on a high-priority compute stream
    (MomentumSGD.update), and the bucket all-reduces - a few small, latency-critical kernels per step - must not queue
    behind the normal-priority weight-gradient GEMMs.  Falls back to the default options where the installed torch lacks
    the switch."""
    opts = None
    if backend == 'nccl':
        try:
            opts = torch.distributed.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        except Exception:
            opts = None
    if opts is not None:
        # decided from the signature, not by catching TypeError: a TypeError raised AFTER partial initialisation would lead to a
        # second init call ("default process group initialised twice") that masks the real error
        import inspect
        try:
            takes = 'pg_options' in inspect.signature(torch.distributed.init_process_group).parameters
        except (TypeError, ValueError):
            takes = False
        if takes:
            return torch.distributed.init_process_group(backend, pg_options=opts, **kw)
    return torch.distributed.init_process_group(backend, **kw)


def rccl_evidence(dev, world):
    """What the collective library saw, gathered over the ranks (bench.py prints it as config.rccl for N > 1): world size, backend,
    RCCL version, every rank's device index and PCI bus id, and - when NCCL_DEBUG=INFO was set and NCCL_DEBUG_FILE names a file - the
    transport lines of rank 0's log.  Two ranks on one device without MRCNN_BENCH_SINGLE_DEVICE is an error (RuntimeError on every
    rank, raised after the gather so that no rank is left waiting in a collective)."""
    import os
    backend = torch.distributed.get_backend()
    prop = torch.cuda.get_device_properties(dev)
    try:
        bus = '%04x:%02x:%02x' % (getattr(prop, 'pci_domain_id', 0), prop.pci_bus_id, prop.pci_device_id)
    except AttributeError:
        bus = None
    mine = {'rank': torch.distributed.get_rank(), 'local_rank': int(os.environ.get('LOCAL_RANK', 0)), 'device_index': dev.index,
            'device_name': prop.name, 'pci_bus_id': bus, 'pid': os.getpid(),
            'visible_devices': os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')}
    ranks = [None] * world
    torch.distributed.all_gather_object(ranks, mine)
    try:
        ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        ver = None
    ev = {'world_size': world, 'backend': backend, 'rccl_version': ver, 'ranks': ranks,
          'distinct_devices': len({(r['pci_bus_id'], r['device_index'], r['visible_devices']) for r in ranks}),
          'nccl_debug': os.environ.get('NCCL_DEBUG'), 'transport': None}
    log = os.environ.get('NCCL_DEBUG_FILE')
    if ev['nccl_debug'] and log and mine['rank'] == 0:
        try:
            path = log.replace('%h', os.uname().nodename).replace('%p', str(os.getpid()))
            keys = ('via P2P', 'via SHM', 'via NET', 'via direct', 'Connected all', 'comm ', 'nRanks', 'XGMI', 'Ring ', 'Tree ')
            lines = [l.strip() for l in open(path, errors='replace') if any(k in l for k in keys)]
            ev['transport'] = lines[:40]
            if os.environ.get('MRCNN_RCCL_LOG_IS_OURS') == '1':       # bench.py asked for this log itself: parsed, so it goes
                os.remove(path)
        except OSError as e:
            ev['transport'] = ['(could not read %s: %s)' % (log, e)]
    if ev['distinct_devices'] < world and os.environ.get('MRCNN_BENCH_SINGLE_DEVICE') != '1':
        raise RuntimeError('data-parallel launch with %d ranks resolved to %d distinct device(s): %s - one process per GPU is '
                           'required (set MRCNN_BENCH_SINGLE_DEVICE=1 only for the functional check on a one-GPU box)'
                           % (world, ev['distinct_devices'], [(r['rank'], r['device_index'], r['pci_bus_id']) for r in ranks]))
    return ev


class WeightDecay(object):
    def __init__(self, rate):
        self.rate = rate


class GradientSynchronizer(object):
    """Bucketed all-reduce of the flat gradient buffer, overlapped with backward.

    Buckets are contiguous slices cut from the END of the buffer (backward fills it from the end).
    ``mark_ready(offset)`` is called by the train chain with the lowest parameter offset whose
    gradient is final; every bucket lying entirely above it is all-reduced on the side stream.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): ~25 MB buckets keep each ring/direct
    transfer in the bandwidth regime while leaving >= 7 buckets to overlap with backward.
    """

    def __init__(self, grads, bucket_bytes=25 << 20, group=None, average=False, sync_single_rank=False):
        """sync_single_rank: run the collectives even in a one-rank group (a SUM over one rank is the identity) - the
        RCCL code path of an N-GPU job, exercised on a one-GPU box (tests/test_dp_gpu.py)."""
        self.grads = grads
        self.group = group
        self.average = average
        n = grads.numel()
        per = max(1, bucket_bytes // 4)
        self.buckets = []                 # (start, end) from the end of the buffer
        end = n
        while end > 0:
            start = max(0, (end - per) // 64 * 64)       # 256-byte aligned starts: a bucket is also a unit of the float4 update kernel
            self.buckets.append((start, end))
            end = start
        self.next = 0
        self.timing = False               # bench.py: HIP events around every bucket's all-reduce
        self._events, self._bwd_end = [], None
        self.stream = torch.cuda.Stream() if grads.is_cuda else None
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1
        self.active = self.world > 1 or (sync_single_rank and torch.distributed.is_initialized())
        # after_bucket(start, end) (optional): called once per bucket, on the bucket's stream, behind its all-reduce - the optimizer's
        # update of exactly those parameters (MomentumSGD.sectioned_update).  With a callback the buckets are also walked when there is
        # nothing to reduce (one rank): the update of a finished section then runs beside the rest of the backward pass.
        self.after_bucket = None

    def begin(self):
        self.next = 0
        self._events = []

    def _all_reduce(self, sl):
        if sl.is_cuda and torch.distributed.get_backend(self.group) == 'gloo':
            # functional check of the multi-process path on ONE GPU (tests; RCCL refuses two ranks on one device): stage
            # through the host explicitly
            host = sl.cpu()
            torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.SUM, group=self.group)
            sl.copy_(host)
        else:
            torch.distributed.all_reduce(sl, op=torch.distributed.ReduceOp.SUM, group=self.group)

    def _reduce(self, start, end):
        sl = self.grads[start:end]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())      # the bucket's producers have been enqueued ...
            from chainer_maskrcnn._hip import nn as hnn
            self.stream.wait_stream(hnn.side_stream(self.grads.device))   # ... on the main and on the weight-gradient stream
            with torch.cuda.stream(self.stream):
                if self.active:
                    ev = None
                    if self.timing:
                        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                        ev[0].record()
                    self._all_reduce(sl)
                    if ev is not None:
                        ev[1].record()
                        self._events.append((end - start, ev[0], ev[1]))
                if self.after_bucket is not None:
                    self.after_bucket(start, end)
        else:
            if self.active:
                self._all_reduce(sl)
            if self.after_bucket is not None:
                self.after_bucket(start, end)

    def mark_ready(self, offset):
        if not (self.active or self.after_bucket is not None):
            return
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= offset:
            self._reduce(*self.buckets[self.next])
            self.next += 1

    def finish(self):
        """All buckets reduced (and updated, with a callback) and visible to the compute stream."""
        if not (self.active or self.after_bucket is not None):
            return
        self.mark_ready(0)
        if self.stream is not None:
            if self.timing:
                self._bwd_end = torch.cuda.Event(enable_timing=True)
                self._bwd_end.record()               # backward (main stream) is done here; what follows is exposed wait
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.average:
            self.grads.div_(self.world)          # extension (--grad-average); the reference sums

    def timing_report(self):
        """After a synchronised step with ``timing`` on: per-bucket all-reduce ms (in launch order, last layers first) and
        the fraction of the communication time that ran under backward (1 - exposed / total)."""
        if not self._events:
            return None
        ms = [e0.elapsed_time(e1) for _, e0, e1 in self._events]
        total = sum(ms)
        exposed = max(0.0, self._bwd_end.elapsed_time(self._events[-1][2]))
        return {'bucket_mb': [round(n * 4 / 2 ** 20, 1) for n, _, _ in self._events], 'bucket_ms': [round(v, 3) for v in ms],
                'allreduce_ms_total': round(total, 3), 'exposed_ms': round(exposed, 3),
                'overlap_fraction': round(1.0 - min(exposed, total) / total, 4) if total > 0 else None}


class MomentumSGD(object):
    LOCAL_BUCKET_BYTES = 8 << 20        # sections of the one-rank sectioned update (the data-parallel path uses its all-reduce buckets)

    def __init__(self, lr=0.01, momentum=0.9, high_priority_stream=None, sectioned_update=None):
        """high_priority_stream (default: on, MRCNN_STEP_STREAM_PRIORITY=0 turns it off): ``update(lossfun, ...)`` issues the
        step on a HIGH-priority HIP stream.  The step's main stream is the critical path (forward, data gradients); the
        weight-gradient and auxiliary streams only have to be done by the end of the step, and at equal priority their
        workgroups take CU slots from it - same-process A/B on configs[2] (tools/ab_prio.py): 25.55 -> 25.17 ms.

        sectioned_update (OPT-IN, MRCNN_SECTIONED_UPDATE=1 or the argument; round 5): ``update(lossfun, ...)`` with a train chain
        applies the SGD step section by section - a contiguous slice of the flat parameter buffer is updated as soon as the
        backward pass has left it (the chain's grad_ready_hook; one rank) or as soon as its gradient bucket has been all-reduced
        (data-parallel: on the collective stream, behind the all-reduce) - instead of one 176-MB pass (155 us at 5.7 TB/s) behind
        the last gradient.  Element-wise the same arithmetic: parameters and momentum are bit-identical to the single pass
        (tests/test_step_gpu.py).  Measured on configs[2], one process (tools/ab_step.py): 21.47 ms single pass, 21.52 ms with
        8-MB sections, 21.57 ms with 25-MB sections - the step is bound by the chip's throughput, the update's 0.9 GB
        cost the same HBM time beside the backward pass as behind it, so the single pass stays the default.  An exception raised
        inside the backward pass leaves the sections already finished updated."""
        self.lr, self.momentum = lr, momentum
        if sectioned_update is None:
            import os
            sectioned_update = os.environ.get('MRCNN_SECTIONED_UPDATE', '0') == '1'
        self.sectioned_update = sectioned_update
        self._local_sync = None
        self._updated_down_to = None
        self._section_views = {}
        self.weight_decay = 0.0
        self.target = None
        self.sync = None
        self.t = 0
        if high_priority_stream is None:
            import os
            high_priority_stream = os.environ.get('MRCNN_STEP_STREAM_PRIORITY', '1') != '0'
        self.high_priority_stream = high_priority_stream
        self._hi = {}

    def _step_stream(self):
        dev = self.ps.params.device
        if not self.high_priority_stream or dev.type != 'cuda' or torch.cuda.is_current_stream_capturing():
            return None
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in self._hi:
            self._hi[key] = torch.cuda.Stream(device=dev, priority=-1)
        return self._hi[key]

    def setup(self, link):
        """link: the train chain (has .faster_rcnn.ps) or the model (has .ps)."""
        self.target = link
        self.ps = link.faster_rcnn.ps if hasattr(link, 'faster_rcnn') else link.ps
        return self

    def add_hook(self, hook):
        if isinstance(hook, WeightDecay):
            self.weight_decay = hook.rate
        else:
            raise TypeError('only WeightDecay hooks exist on this path (train.py:109)')

    def enable_data_parallel(self, bucket_bytes=25 << 20, average=False, broadcast=True, sync_single_rank=False):
        """One process per GPU; every rank keeps a replica and applies the same update to the summed gradients.  The
        replicas must START equal: rank 0's parameters, momentum and BatchNorm running statistics are broadcast once
        here (the reference's MultiprocessParallelUpdater broadcasts the master's parameters every step,
        SURVEY.md section 3.5; with identical updates once is enough) - so per-rank ``--weight`` files or seeds cannot
        silently diverge."""
        self.sync = GradientSynchronizer(self.ps.grads, bucket_bytes, average=average, sync_single_rank=sync_single_rank)
        self.target.grad_ready_hook = self.sync.mark_ready
        if broadcast and self.sync.active:
            bufs = [self.ps.params, self.ps.momentum] + [self.ps.buffers[k] for k in sorted(self.ps.buffers)]
            gloo = torch.distributed.get_backend() == 'gloo'
            for b in bufs:
                if gloo and b.is_cuda:
                    h = b.cpu()
                    torch.distributed.broadcast(h, src=0)
                    b.copy_(h)
                else:
                    torch.distributed.broadcast(b, src=0)

    # ---- trainer state (checkpoint / resume) -----------------------------------------------------
    def state_dict(self):
        """Everything a bit-identical continuation needs: parameters, momentum, BatchNorm running statistics, the
        update count, the hyper-parameters and the device-resident sampler seeds of the train chain."""
        ps = self.ps
        d = {'params': ps.params.detach().cpu(), 'momentum': ps.momentum.detach().cpu(),
             'buffers': {k: v.detach().cpu() for k, v in ps.buffers.items()},
             't': self.t, 'lr': self.lr, 'sgd_momentum': self.momentum, 'weight_decay': self.weight_decay, 'samplers': {}}
        for name in ('proposal_target_creator', 'anchor_target_creator'):
            c = getattr(self.target, name, None)
            if c is not None:
                d['samplers'][name] = {'seed': c.seed, 'state': None if c._state is None else c._state.detach().cpu()}
        return d

    def load_state_dict(self, d):
        ps = self.ps
        if tuple(d['params'].shape) != tuple(ps.params.shape):
            raise ValueError('trainer state is for a different model: %r parameters, this model has %r'
                             % (tuple(d['params'].shape), tuple(ps.params.shape)))
        ps.params.copy_(d['params'])
        ps.momentum.copy_(d['momentum'])
        for k, v in d['buffers'].items():
            ps.buffers[k].copy_(v)
        self.t, self.lr, self.momentum, self.weight_decay = d['t'], d['lr'], d['sgd_momentum'], d['weight_decay']
        for name, st in d.get('samplers', {}).items():
            c = getattr(self.target, name, None)
            if c is not None:
                c.seed = st['seed']
                c._state = None if st['state'] is None else st['state'].to(ps.params.device)

    def update(self, lossfun=None, *args, **kwds):
        """Chainer semantics: with ``lossfun`` -> loss = lossfun(*args); backward; update.  Without: update only."""
        hi = self._step_stream() if lossfun is not None else None
        if hi is None:
            return self._update(lossfun, *args, **kwds)
        cur = torch.cuda.current_stream(hi.device)
        if cur == hi:
            return self._update(lossfun, *args, **kwds)
        hi.wait_stream(cur)                     # inputs produced on the caller's stream
        with torch.cuda.stream(hi):
            loss = self._update(lossfun, *args, **kwds)
        cur.wait_stream(hi)                     # the caller's stream sees the updated parameters / loss (no host sync)
        return loss

    def _sgd_section(self, start, end):
        ps = self.ps
        key = (start, end, ps.params.data_ptr(), ps.grads.data_ptr(), ps.momentum.data_ptr())
        views = self._section_views.get(key)
        if views is None:
            if len(self._section_views) > 4096:         # (rebound flat buffers: do not keep the old ones alive through stale views)
                self._section_views.clear()
            views = self._section_views[key] = (ps.params[start:end], ps.grads[start:end], ps.momentum[start:end])
        ops.sgd_momentum_wd(views[0], views[1], views[2], self.lr, self.momentum, self.weight_decay)
        self._updated_down_to = start

    def _sectioning(self, lossfun):
        """The synchronizer that walks the sections of this update, or None (plain single pass)."""
        if not self.sectioned_update or lossfun is None or not hasattr(lossfun, 'grad_ready_hook') or not self.ps.params.is_cuda:
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        if self.sync is not None:
            return None if self.sync.average else self.sync          # (--grad-average divides the whole buffer after the last bucket)
        if lossfun.grad_ready_hook is not None:
            return None                                             # someone else listens to the chain
        if self._local_sync is None or self._local_sync.grads is not self.ps.grads:
            self._local_sync = GradientSynchronizer(self.ps.grads, self.LOCAL_BUCKET_BYTES)
        return self._local_sync

    def _update(self, lossfun=None, *args, **kwds):
        loss = None
        sect = self._sectioning(lossfun)
        self._updated_down_to = None
        if sect is not None:
            sect.after_bucket = self._sgd_section
            if sect is not self.sync:
                sect.begin()
                lossfun.grad_ready_hook = sect.mark_ready
        try:
            loss = self._step(lossfun, sect, *args, **kwds)
        finally:
            if sect is not None:
                sect.after_bucket = None
                if sect is not self.sync:
                    lossfun.grad_ready_hook = None
        return loss

    def _step(self, lossfun, sect, *args, **kwds):
        loss = None
        if lossfun is not None:
            if self.sync is not None:
                self.sync.begin()
            if hasattr(lossfun, 'backward_follows'):    # the train chain may start branches of the backward pass inside its forward call
                lossfun.backward_follows = True
            try:
                loss = lossfun(*args, **kwds)
            finally:
                if hasattr(lossfun, 'backward_follows'):
                    lossfun.backward_follows = False
            if hasattr(lossfun, 'unit_upstream'):       # the train chain: d loss = 1 here, no gradient-scaling pass
                lossfun.unit_upstream = True
            try:
                loss.backward()
            finally:
                if hasattr(lossfun, 'unit_upstream'):
                    lossfun.unit_upstream = False
        if self.sync is not None:
            self.sync.finish()
        elif sect is not None:
            sect.finish()
        rest = self.ps.params.numel() if self._updated_down_to is None else self._updated_down_to
        if rest > 0:        # everything (no sections), or what no section covered
            ops.sgd_momentum_wd(self.ps.params[:rest], self.ps.grads[:rest], self.ps.momentum[:rest], self.lr, self.momentum, self.weight_decay)
        self.t += 1
        return loss


class GraphedStep(object):
    """One whole training step (forward, backward, SGD update) captured into a HIP graph and replayed.

    The step has no device->host copy and static shapes (padded RoI / sample rows, device-resident counts and
    sampler seeds), so the ~800 kernel launches of a step can be recorded once; a replay costs one host call and
    removes the launch gaps (MI355X_MICROARCH.md price list, rows 'boundary' / 'graph-replay-floor').
    Inputs are copied into static buffers; ``chain.observation`` tensors are rewritten by every replay.
    Single-GPU only (the data-parallel path issues its RCCL collectives eagerly).
    """

    def __init__(self, optimizer, chain, example_batch, scale=1.0, warmup=3):
        if optimizer.sync is not None:
            raise RuntimeError('GraphedStep: data-parallel steps run eagerly')
        self.optimizer, self.chain, self.scale = optimizer, chain, scale
        # graph replay and the eager high-priority step stream do not mix: with that stream created in the process the replay
        # of the captured step measured 36.4 ms instead of 26.6 ms (ROCm 7.2); a graphed optimizer stays on normal priority
        optimizer.high_priority_stream = False
        self.static = [t.clone() for t in example_batch]
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optimizer.update(chain, *self.static, scale)
        cur.wait_stream(s)
        torch.cuda.synchronize()
        self._capture()

    def _capture(self):
        # the learning rate is a kernel argument baked into the captured launch: re-capture when it changes
        # (ExponentialShift('lr', 0.1), train.py:139-140)
        self._lr = self.optimizer.lr
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.optimizer.update(self.chain, *self.static, self.scale)

    def __call__(self, *batch):
        if self.optimizer.lr != self._lr:
            torch.cuda.synchronize()
            self._capture()
        for dst, src in zip(self.static, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.optimizer.t += 1
        return self.chain.observation['loss']
