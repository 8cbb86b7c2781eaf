"""Persistent cross-tile pipelined GEMM (k_conv_pipe) against the per-tile kernel: identical bits, and the timings of the layer
shapes that qualify (plain forward / backward-data launches of several rounds)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib as _lib, check
lib = _lib()
dev = torch.device('cuda:0')
SHAPES = [  # N, H, W, Cin, Cout, k, pad, fwd tile
    (512, 14, 14, 256, 256, 3, 1, 0), (2, 256, 256, 256, 256, 3, 1, 0), (2, 256, 256, 64, 256, 1, 0, 2), (2, 256, 256, 256, 64, 1, 0, 2),
    (2, 128, 128, 128, 512, 1, 0, 2), (2, 128, 128, 512, 128, 1, 0, 2), (2, 64, 64, 256, 1024, 1, 0, 2), (2, 64, 64, 1024, 256, 1, 0, 2),
    (2, 256, 256, 64, 64, 3, 1, 2), (2, 128, 128, 128, 128, 3, 1, 2), (2, 64, 64, 256, 256, 3, 1, 2), (2, 128, 128, 256, 256, 3, 1, 0),
    (512, 14, 14, 256, 384, 1, 0, 0), (2, 256, 256, 256, 256, 1, 0, 2), (2, 37, 53, 64, 96, 1, 0, 2)]


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {0: 0.0, 1: 0.0}
for (N, H, W, Ci, Co, k, p, ft) in SHAPES:
    hnn.set_winograd_pass_tiles(ft, 0, 0)
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, k, k, Ci), device=dev) * 0.05
    gy = torch.randn((N, H, W, Co), device=dev)
    res, tm = {}, {}
    for on in (0, 1):
        check(lib.mrcnn_conv2d_set_persistent(on, 1.5))
        y = hnn.conv2d_fwd_raw(x, w, None, 1, p, True)
        st = hnn.conv2d_fwd_bnstats_raw(x, w, 1, p)
        gx = hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, p)
        res[on] = (y, gx) + ((st[0], st[2]) if st is not None else ())
        tm[on] = (timeit(lambda: hnn.conv2d_fwd_raw(x, w, None, 1, p, True)), timeit(lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, p)))
        tot[on] += sum(tm[on])
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    print('%4dx%3dx%3d %5d->%5d k%d: identical %s   fwd %7.1f -> %7.1f us   bwd_data %7.1f -> %7.1f us' % (N, H, W, Ci, Co, k, same, tm[0][0], tm[1][0], tm[0][1], tm[1][1]), flush=True)
check(lib.mrcnn_conv2d_set_persistent(0, 1.5))
hnn.set_winograd_pass_tiles(2, 0, 0)
print('sum: per-tile %.1f us, persistent %.1f us' % (tot[0], tot[1]))
