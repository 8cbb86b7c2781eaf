"""Prefetching batch loader: dataset + transform -> the padded batch tensors the train chain takes.

The reference iterates with ``MultithreadIterator`` / one ``SerialIterator`` per GPU and ``concat_examples``
(train.py:117-126; batch size 1 per GPU).  MI355X-native form: worker threads decode and transform ahead of the
step (PIL decoding and NumPy release the GIL), every rank walks its own shard of a seeded permutation, batches are
assembled in pinned host memory and copied on a dedicated HIP stream so that the transfer (0.7 ms for a bs-2 1024^2
batch at PCIe 5 rates) overlaps the previous step.  Batch layout = ``utils.synthetic.make_batch``:
  imgs (N,3,H,W) f32 (zero-padded bottom/right to the largest image of the batch), bboxes (N,G,4) f32,
  labels (N,G) i32 with -1 on padded rows, masks (N,G,H,W) u8  |  keypoints (N,G,K,3) f32.
"""
import queue
import threading

import numpy as np


def collate(examples, max_gt=None, keypoints=False, size_multiple=64):
    """examples: outputs of Transform / KeypointTransform.  G = max_gt or the largest instance count in the batch
    (extra instances are dropped, missing ones padded with label -1).  H, W are rounded up to ``size_multiple`` (the
    coarsest pyramid level has stride 64)."""
    N = len(examples)
    G = max_gt or max(1, max(e[1].shape[0] for e in examples))
    H = max(e[0].shape[1] for e in examples)
    W = max(e[0].shape[2] for e in examples)
    H, W = -(-H // size_multiple) * size_multiple, -(-W // size_multiple) * size_multiple
    out = {'imgs': np.zeros((N, 3, H, W), np.float32), 'bboxes': np.zeros((N, G, 4), np.float32),
           'labels': np.full((N, G), -1, np.int32), 'scales': np.zeros((N,), np.float32),
           'sizes': np.zeros((N, 2), np.float32)}      # each image's own (h, w) inside the zero-padded batch tensor
    if keypoints:
        K = examples[0][3].shape[1] if examples[0][3].ndim == 3 else 17
        out['keypoints'] = np.zeros((N, G, K, 3), np.float32)
    else:
        out['masks'] = np.zeros((N, G, H, W), np.uint8)
    for i, (img, bbox, label, extra, scale) in enumerate(examples):
        h, w = img.shape[1:]
        out['imgs'][i, :, :h, :w] = img
        g = min(G, bbox.shape[0])
        out['bboxes'][i, :g] = bbox[:g]
        out['labels'][i, :g] = label[:g]
        out['scales'][i] = scale
        out['sizes'][i] = (h, w)
        if keypoints:
            out['keypoints'][i, :g] = extra[:g]
        else:
            out['masks'][i, :g, :h, :w] = extra[:g]
    return out


class _PinnedRing(object):
    """Persistent pinned staging buffers (pinning memory per batch costs milliseconds): K slots, each a grow-only byte
    buffer per tensor name; a slot is reused only after the copies issued from it have completed (event)."""

    def __init__(self, slots=3):
        self.slots = [dict(bufs={}, event=None) for _ in range(slots)]
        self.i = 0

    def next_slot(self):
        slot = self.slots[self.i % len(self.slots)]
        self.i += 1
        if slot['event'] is not None:
            slot['event'].synchronize()
        return slot

    @staticmethod
    def upload(slot, name, arr, dev):
        import torch
        arr = np.ascontiguousarray(arr)
        n = arr.nbytes
        buf = slot['bufs'].get(name)
        if buf is None or buf.numel() < n:
            buf = torch.empty((max(n, 1) * 5 // 4,), dtype=torch.uint8).pin_memory()
            slot['bufs'][name] = buf
        if n:       # NumPy copy: a torch CPU copy_ fans out over every core of the box (OpenMP) and starves the workers
            np.copyto(buf.numpy()[:n], arr.reshape(-1).view(np.uint8))
        out = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0].reshape(-1)).dtype, device=dev)
        if n:
            out.view(-1).view(torch.uint8).copy_(buf[:n], non_blocking=True)
        return out


class BatchLoader(object):
    """Endless iterator of device batches.  ``rank``/``world`` shard the (seeded, per-epoch) permutation."""

    def __init__(self, dataset, transform, batch_size=1, shuffle=True, seed=0, rank=0, world=1, num_workers=4,
                 prefetch=4, max_gt=None, keypoints=False, device=None, skip_empty=True, start_ticket=0):
        self.dataset, self.transform = dataset, transform
        self.bs, self.shuffle, self.seed, self.rank, self.world = batch_size, shuffle, seed, rank, world
        self.max_gt, self.keypoints, self.device, self.skip_empty = max_gt, keypoints, device, skip_empty
        self._idx = queue.Queue(maxsize=prefetch * batch_size * 2)
        self._out = {}
        self._cv = threading.Condition()
        # start_ticket: resume - the first `start_ticket` examples of the (seeded) sequence are skipped without being
        # decoded; `ticket` (examples consumed so far) is what a trainer checkpoint stores
        self._start = int(start_ticket)
        self._next_put, self._next_get = self._start, self._start
        self._stop = False
        self._threads = [threading.Thread(target=self._feed, daemon=True)]
        self._threads += [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, num_workers))]
        self._window = prefetch * batch_size
        self._stream = None
        self._ring = _PinnedRing()
        for t in self._threads:
            t.start()

    # index producer: (ticket, dataset index)
    def _feed(self):
        epoch, ticket = 0, 0
        n = len(self.dataset)
        while not self._stop:
            order = np.random.RandomState(self.seed + epoch).permutation(n) if self.shuffle else np.arange(n)
            for i in order[self.rank::self.world]:
                if ticket < self._start:
                    ticket += 1
                    continue
                while not self._stop:
                    try:
                        self._idx.put((ticket, int(i)), timeout=0.1)
                        break
                    except queue.Full:
                        continue
                ticket += 1
            epoch += 1

    def _work(self):
        while not self._stop:
            try:
                ticket, i = self._idx.get(timeout=0.1)
            except queue.Empty:
                continue
            with self._cv:                                   # bounded run-ahead keeps examples in ticket order
                while ticket >= self._next_get + self._window and not self._stop:
                    self._cv.wait(0.1)
            try:
                ex = self.transform(self.dataset[i])
            except Exception as e:                           # surfaced in the consumer thread
                ex = e
            with self._cv:
                self._out[ticket] = ex
                self._cv.notify_all()

    def _next_example(self):
        while True:
            with self._cv:
                while self._next_get not in self._out:
                    self._cv.wait(0.1)
                ex = self._out.pop(self._next_get)
                self._next_get += 1
                self._cv.notify_all()
            if isinstance(ex, Exception):
                raise ex
            if self.skip_empty and ex[1].shape[0] == 0:      # images whose annotations were all filtered out (:93-96)
                continue
            return ex

    def __iter__(self):
        return self

    @property
    def ticket(self):
        """Examples consumed so far (pass as start_ticket to continue the same sequence)."""
        return self._next_get

    def _next_device_batch(self):
        """transform = RawTransform: raw uint8 images / masks are uploaded at their original size and resized on the GPU
        into the padded batch tensors (10x fewer PCIe bytes, no host resize)."""
        import torch
        from chainer_maskrcnn._hip import lib, check, ptr
        dev = torch.device(self.device)
        exs = [self._next_example() for _ in range(self.bs)]
        N = len(exs)
        G = self.max_gt or max(1, max(e[1].shape[0] for e in exs))
        H = -(-max(e[5][0] for e in exs) // 64) * 64
        W = -(-max(e[5][1] for e in exs) // 64) * 64
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        bboxes = np.zeros((N, G, 4), np.float32)
        labels = np.full((N, G), -1, np.int32)
        scales = np.zeros((N,), np.float32)
        sizes = np.zeros((N, 2), np.float32)
        slot = self._ring.next_slot()
        with torch.cuda.stream(self._stream):
            st = self._stream.cuda_stream
            imgs = torch.zeros((N, 3, H, W), dtype=torch.float32, device=dev)
            extra = (torch.zeros((N, G, exs[0][3].shape[1], 3), dtype=torch.float32, device=dev) if self.keypoints
                     else torch.zeros((N, G, H, W), dtype=torch.uint8, device=dev))
            keep = []
            for i, (img, bbox, label, ext, scale, (oh, ow)) in enumerate(exs):
                g = min(G, bbox.shape[0])
                bboxes[i, :g], labels[i, :g], scales[i], sizes[i] = bbox[:g], label[:g], scale, (oh, ow)
                raw = self._ring.upload(slot, 'img%d' % i, img, dev)
                check(lib().mrcnn_image_resize_u8_f32(ptr(raw), img.shape[0], img.shape[1], ptr(imgs[i]), oh, ow, H, W, 255.0, st))
                keep.append(raw)
                if self.keypoints:
                    extra[i, :g] = self._ring.upload(slot, 'kp%d' % i, ext[:g], dev)
                elif g > 0:
                    m = self._ring.upload(slot, 'mask%d' % i, ext[:g], dev)
                    check(lib().mrcnn_mask_resize_nearest_u8(ptr(m), g, ext.shape[1], ext.shape[2], ptr(extra[i]), oh, ow, H, W, st))
                    keep.append(m)
            out = {'imgs': imgs, 'bboxes': self._ring.upload(slot, 'bboxes', bboxes, dev),
                   'labels': self._ring.upload(slot, 'labels', labels, dev),
                   'keypoints' if self.keypoints else 'masks': extra, 'scales': scales, 'sizes': sizes}
            slot['event'] = torch.cuda.Event()
            slot['event'].record(self._stream)
        cur = torch.cuda.current_stream(dev)
        cur.wait_stream(self._stream)
        for v in list(out.values()) + keep:
            if torch.is_tensor(v):
                v.record_stream(cur)
        return out

    def __next__(self):
        if getattr(self.transform, 'out_size', None) is not None and self.device is not None and str(self.device).startswith('cuda'):
            return self._next_device_batch()
        batch = collate([self._next_example() for _ in range(self.bs)], self.max_gt, self.keypoints)
        if self.device is None:
            return batch
        import torch
        dev = torch.device(self.device)
        if dev.type != 'cuda':
            return {k: torch.from_numpy(v) for k, v in batch.items()}
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=dev)
        out = {}
        slot = self._ring.next_slot()
        with torch.cuda.stream(self._stream):
            for k, v in batch.items():
                if k == 'scales':                      # consumed by the host (a float per step): stays a NumPy array
                    out[k] = v
                    continue
                out[k] = self._ring.upload(slot, k, v, dev)
            slot['event'] = torch.cuda.Event()
            slot['event'].record(self._stream)
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        for v in out.values():
            if torch.is_tensor(v):
                v.record_stream(torch.cuda.current_stream(dev))
        return out

    def close(self):
        self._stop = True
        with self._cv:
            self._cv.notify_all()
