"""Winograd threshold sweep on the 3x3 layer shapes of the step (direct vs F(2x2) vs auto tile)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
SHAPES = [(2, 256, 256, 64, 64), (2, 128, 128, 128, 128), (2, 64, 64, 256, 256), (2, 32, 32, 512, 512), (2, 16, 16, 256, 256),
          (2, 8, 8, 256, 256), (512, 7, 7, 256, 256), (512, 14, 14, 256, 256), (2, 128, 128, 256, 256)]
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('%-28s %s' % ('shape', '   '.join('%-26s' % c for c in ('direct', 'F(2x2)', 'F(4x4)'))))
for (N, H, W, Ci, Co) in SHAPES:
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, 3, 3, Ci), device=dev) * 0.05
    b = torch.zeros((Co,), device=dev); gy = torch.randn((N, H, W, Co), device=dev)
    row = []
    for cfg in ((1 << 20, 1 << 30, 0), (32, 1, 2), (32, 1, 4)):
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(*cfg))
        t1 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, b, 1, 1, True))
        t2 = timeit(lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, 1))
        t3 = timeit(lambda: hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, 1, False))
        row.append('%6.0f %6.0f %6.0f us     ' % (t1, t2, t3))
    print('%4dx%3dx%3d %4d->%4d      %s' % (N, H, W, Ci, Co, '   '.join(row)))
_hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, 0))
