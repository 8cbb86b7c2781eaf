"""Does splitting a stack of Winograd 3x3 layers over the batch axis onto several HIP streams hide the HBM-bound transform
kernels under the MFMA-bound GEMMs of the other chunks?  (mask head: 512 RoIs x 14x14x256, four layers; FPN conv_p2: 2 x 256^2.)
Prints ms for the whole stack with 1, 2, 4 streams, forward and backward."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
hnn.set_winograd_pass_tiles(0, 0, 0)


def run(shape, layers, nsplit, mode, reps=10, sequential=False):
    N, H, W, C = shape
    x = torch.randn(shape, device=dev)
    ws = [torch.randn((C, 3, 3, C), device=dev) * 0.02 for _ in range(layers)]
    b = torch.zeros((C,), device=dev)
    gws = [torch.empty_like(w) for w in ws]
    main = torch.cuda.current_stream(dev)
    # sequential: the chunks run one after the other on ONE stream and workspace (is a chunk's V / M still in the
    # 256 MiB Infinity Cache when the next kernel reads it?)
    streams = [main] * nsplit if sequential else [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
    chunks = list(torch.chunk(x, nsplit, 0))
    gchunks = [torch.randn_like(c) for c in chunks]

    def once():
        if not sequential:
            for s in streams:
                s.wait_stream(main)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                h = chunks[i]
                if mode == 'fwd':
                    for w in ws:
                        h = hnn.conv2d_fwd_raw(h, w, b, 1, 1, True)
                else:
                    g = gchunks[i]
                    for w, gw in zip(ws, gws):
                        hnn.conv2d_bwd_filter_raw(h, g, tuple(w.shape), 1, 1, False, gw=gw, gb=None, accumulate=(i > 0))
                        g = hnn.conv2d_bwd_data_raw(g, w, tuple(h.shape), 1, 1, relu_x=h)
        if not sequential:
            for s in streams:
                main.wait_stream(s)
    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for shape, layers in (((512, 14, 14, 256), 4), ((2, 256, 256, 256), 2), ((8, 128, 128, 256), 2)):
    for mode in ('fwd', 'bwd'):
        print(shape, layers, mode, ' '.join('%d streams %.3f ms' % (n, run(shape, layers, n, mode)) for n in (1, 2, 4) if shape[0] % n == 0), flush=True)
        print(shape, layers, mode, ' '.join('%d sequential chunks %.3f ms' % (n, run(shape, layers, n, mode, sequential=True)) for n in (1, 2, 3, 4, 8) if shape[0] >= n), flush=True)
