"""Transform-only timing of the Winograd layers (GEMM launches skipped through mrcnn_conv2d_set_debug_skip(1)) + per-kernel
durations from rocprofv3 when run under it."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib, check
dev = torch.device('cuda:0')
SHAPES = [(2, 256, 256, 256, 256), (512, 14, 14, 256, 256), (2, 128, 128, 256, 256), (2, 64, 64, 256, 256), (512, 7, 7, 256, 256)]
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
hnn.set_winograd_pass_tiles(0, 0, 0)
for (N, H, W, Ci, Co) in SHAPES:
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, 3, 3, Ci), device=dev) * 0.05
    b = torch.zeros((Co,), device=dev); gy = torch.randn((N, H, W, Co), device=dev)
    act = 4.0 * N * H * W * Ci
    vb = lib().mrcnn_conv2d_winograd_v_bytes(N, H, W, Ci, Co, 3, 3, 1, 1)
    check(lib().mrcnn_conv2d_set_debug_skip(1))
    t1 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, b, 1, 1, True))
    check(lib().mrcnn_conv2d_set_debug_skip(0))
    t0 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, b, 1, 1, True))
    # forward transforms: input (act -> V), filter, output (M = V-sized -> act)
    byt = act + vb + vb + act
    print('%4dx%3dx%3d %d->%d: fwd total %.3f ms, transforms only %.3f ms = %.0f GB/s over %.0f MB (V %.0f MB)' % (N, H, W, Ci, Co, t0 * 1e3, t1 * 1e3, byt / t1 / 1e9, byt / 1e6, vb / 1e6))
