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

    def __init__(self, grads, bucket_bytes=25 << 20, group=None, average=False):
        self.grads = grads
        self.group = group
        self.average = average
        n = grads.numel()
        per = max(1, bucket_bytes // 4)
        self.buckets = []                 # (start, end) from the end of the buffer
        end = n
        while end > 0:
            start = max(0, end - per)
            self.buckets.append((start, end))
            end = start
        self.next = 0
        self.stream = torch.cuda.Stream() if grads.is_cuda else None
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_initialized() else 1

    def begin(self):
        self.next = 0

    def _reduce(self, start, end):
        sl = self.grads[start:end]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())      # the bucket's producers have been enqueued ...
            from chainer_maskrcnn._hip import nn as hnn
            self.stream.wait_stream(hnn.side_stream(self.grads.device))   # ... on the main and on the weight-gradient stream
            with torch.cuda.stream(self.stream):
                torch.distributed.all_reduce(sl, op=torch.distributed.ReduceOp.SUM, group=self.group)
        else:
            torch.distributed.all_reduce(sl, op=torch.distributed.ReduceOp.SUM, group=self.group)

    def mark_ready(self, offset):
        if self.world == 1:
            return
        while self.next < len(self.buckets) and self.buckets[self.next][0] >= offset:
            self._reduce(*self.buckets[self.next])
            self.next += 1

    def finish(self):
        """All buckets reduced and visible to the compute stream."""
        if self.world == 1:
            return
        self.mark_ready(0)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.average:
            self.grads.div_(self.world)          # extension (--grad-average); the reference sums


class MomentumSGD(object):
    def __init__(self, lr=0.01, momentum=0.9):
        self.lr, self.momentum = lr, momentum
        self.weight_decay = 0.0
        self.target = None
        self.sync = None
        self.t = 0

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

    def enable_data_parallel(self, bucket_bytes=25 << 20, average=False):
        self.sync = GradientSynchronizer(self.ps.grads, bucket_bytes, average=average)
        self.target.grad_ready_hook = self.sync.mark_ready

    def update(self, lossfun=None, *args, **kwds):
        """Chainer semantics: with ``lossfun`` -> loss = lossfun(*args); backward; update.  Without: update only."""
        loss = None
        if lossfun is not None:
            if self.sync is not None:
                self.sync.begin()
            loss = lossfun(*args, **kwds)
            loss.backward()
        if self.sync is not None:
            self.sync.finish()
        ops.sgd_momentum_wd(self.ps.params, self.ps.grads, self.ps.momentum, self.lr, self.momentum, self.weight_decay)
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
        self.static = [t.clone() for t in example_batch]
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(warmup):
                optimizer.update(chain, *self.static, scale)
        cur.wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            optimizer.update(chain, *self.static, scale)

    def __call__(self, *batch):
        for dst, src in zip(self.static, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.optimizer.t += 1
        return self.chain.observation['loss']
