"""k_wino_output<4> at 2 / 3 / 4 waves per SIMD: transform-only timing of the forward Winograd call (GEMM skipped) on the step's big layers."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib, check
dev = torch.device('cuda:0')
SHAPES = [(2, 256, 256, 256, 256), (512, 14, 14, 256, 256), (2, 128, 128, 256, 256), (2, 64, 64, 512, 512)]
def timeit(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
hnn.set_winograd_pass_tiles(0, 0, 0)
for (N, H, W, Ci, Co) in SHAPES:
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, 3, 3, Ci), device=dev) * 0.05
    b = torch.zeros((Co,), device=dev)
    check(lib().mrcnn_conv2d_set_debug_skip(1))
    for rep in range(2):
        for occ in (0, 3, 4):
            check(lib().mrcnn_debug_wino_output_occupancy(occ))
            t1 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, b, 1, 1, True))
            print('%4dx%3dx%3d %d->%d: occ %d  input+filter+output transforms %.1f us' % (N, H, W, Ci, Co, occ, t1 * 1e6))
    check(lib().mrcnn_debug_wino_output_occupancy(0))
    check(lib().mrcnn_conv2d_set_debug_skip(0))
