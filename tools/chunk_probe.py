"""Does a Winograd convolution run faster in CHUNKS whose intermediates (V, M) stay in the 256-MiB Infinity Cache?  The mask-head layer
(R x 14 x 14 x 256 -> 256, F(4x4), emulated arithmetic) on R RoIs in one call against the same RoIs in calls of R / n; the FPN p2 layer
(2 x 256 x 256 x 256) against its two images one at a time.  Times are whole calls (transforms + GEMM + output transform)."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib, check
dev = torch.device('cuda:0')
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
check(lib().mrcnn_conv2d_set_split_operands(mode, mode, mode))
hnn.set_winograd_pass_tiles(0, 0, 0)


def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def probe(name, shape, parts_list):
    x = torch.randn(shape, device=dev); w = torch.randn((256, 3, 3, shape[3]), device=dev) * 0.02
    gy = torch.randn(shape[:3] + (256,), device=dev)
    for parts in parts_list:
        n = shape[0] // parts
        xs = [x[i * n:(i + 1) * n] for i in range(parts)]; gs = [gy[i * n:(i + 1) * n] for i in range(parts)]
        tf = timeit(lambda: [hnn.conv2d_fwd_raw(a, w, None, 1, 1, False) for a in xs])
        tb = timeit(lambda: [hnn.conv2d_bwd_data_raw(g, w, tuple(a.shape), 1, 1) for g, a in zip(gs, xs)])
        vs = [hnn.conv2d_fwd_raw(a, w, None, 1, 1, False, keep_v=True)[1] for a in xs]
        tw = timeit(lambda: [hnn.conv2d_bwd_filter_raw(a, g, tuple(w.shape), 1, 1, False, wino_v=v, accumulate=(i > 0)) for i, (a, g, v) in enumerate(zip(xs, gs, vs))])
        print('%-28s %d call(s) of %4d: fwd %7.1f us  bwd-data %7.1f us  bwd-filter %7.1f us' % (name, parts, n, tf, tb, tw), flush=True)


probe('mask head 448x14x14x256', (448, 14, 14, 256), (1, 2, 4))
probe('mask head 512x14x14x256', (512, 14, 14, 256), (1, 2, 4))
probe('fpn p2 2x256x256x256', (2, 256, 256, 256), (1, 2))
check(lib().mrcnn_conv2d_set_split_operands(0, 0, 0))
