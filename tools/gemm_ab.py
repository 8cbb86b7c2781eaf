"""GEMM launches of fixed convolution shapes in the emulated arithmetic (split operands 3,3,3), GEMM kernels only
(mrcnn_conv2d_set_debug_skip(2)): microseconds per call and the worst error against the float32-MFMA kernels.  Run once per library
build (MRCNN_HIP_LIB_AB=<other .so>) and compare: compile-time variants of the K loop cannot be switched inside one process.
usage: python tools/gemm_ab.py [tag]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib, check
dev = torch.device('cuda:0')
tag = sys.argv[1] if len(sys.argv) > 1 else 'lib'
SHAPES = [  # N, H, W, Cin, Cout, k, pad   (the direct 1x1 layers of ResNet-50 / FPN at 2 x 1024^2 + two heads layers)
    (2, 256, 256, 64, 256, 1, 0), (2, 256, 256, 256, 64, 1, 0), (2, 128, 128, 128, 512, 1, 0), (2, 128, 128, 512, 128, 1, 0),
    (2, 64, 64, 256, 1024, 1, 0), (2, 64, 64, 1024, 256, 1, 0), (2, 32, 32, 512, 2048, 1, 0), (2, 32, 32, 2048, 512, 1, 0),
    (2, 128, 128, 512, 256, 1, 0), (2, 64, 64, 1024, 256, 1, 0), (2, 32, 32, 2048, 256, 1, 0), (2, 256, 256, 256, 32, 1, 0),
    (512, 1, 1, 12544, 1024, 1, 0), (512, 28, 28, 256, 1024, 1, 0), (2, 32, 32, 1024, 2048, 1, 0), (2, 64, 64, 512, 1024, 1, 0),
]
def timeit(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = [0.0, 0.0, 0.0]
print('%s  (us per call: forward / backward-data / backward-filter, GEMM launches only; max rel error vs the float32 kernels)' % tag)
for (N, H, W, Ci, Co, k, p) in SHAPES:
    g = torch.Generator(device='cpu').manual_seed(Ci * 7 + Co)
    x = torch.randn((N, H, W, Ci), generator=g).to(dev); w = (torch.randn((Co, k, k, Ci), generator=g) * 0.05).to(dev)
    gy = torch.randn((N, H, W, Co), generator=g).to(dev)
    outs = {}
    for mode in ((0, 0, 0), (3, 3, 3)):
        check(lib().mrcnn_conv2d_set_split_operands(*mode))
        outs[mode] = (hnn.conv2d_fwd_raw(x, w, None, 1, p, False).clone(), hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, p).clone(),
                      hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, p, False)[0].clone())
    err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs[(3, 3, 3)], outs[(0, 0, 0)]))
    check(lib().mrcnn_conv2d_set_debug_skip(2))
    t1 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, None, 1, p, False))
    t2 = timeit(lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, p))
    t3 = timeit(lambda: hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, p, False))
    check(lib().mrcnn_conv2d_set_debug_skip(0))
    for i, t in enumerate((t1, t2, t3)): tot[i] += t
    print('%4dx%3dx%3d %5d->%5d  %7.1f %7.1f %7.1f   err %.1e' % (N, H, W, Ci, Co, t1, t2, t3, err), flush=True)
print('sum %8.1f %8.1f %8.1f  total %.1f us' % (tot[0], tot[1], tot[2], sum(tot)))
