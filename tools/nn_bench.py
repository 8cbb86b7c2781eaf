"""BatchNorm forward / backward timing at the ResNet-50 shapes of the 2 x 1024^2 step (HIP events, algorithmic GB/s)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import ops
dev = torch.device('cuda:0')
SHAPES = [(524288, 64, 1), (131072, 64, 6), (131072, 256, 4), (32768, 128, 8), (32768, 512, 5), (8192, 256, 12),
          (8192, 1024, 7), (2048, 512, 6), (2048, 2048, 4)]       # (pixels, channels, count per step)


def timed(f, n=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tf = tb = 0.0
print('%8s %5s %3s %9s %8s %9s %8s' % ('pixels', 'C', 'n', 'fwd us', 'GB/s', 'bwd us', 'GB/s'))
for P, C, cnt in SHAPES:
    x = torch.randn((P, C), device=dev); gy = torch.randn((P, C), device=dev)
    g = torch.ones((C,), device=dev); b = torch.zeros((C,), device=dev)
    y, m, s = ops.bn_train_fwd(x, g, b, relu=True)
    f = timed(lambda: ops.bn_train_fwd(x, g, b, relu=True))
    bw = timed(lambda: ops.bn_train_bwd(gy, x, None, g, m, s, relu=True, beta=b))
    nb = P * C * 4
    print('%8d %5d %3d %9.1f %8.0f %9.1f %8.0f' % (P, C, cnt, f, 3 * nb / f / 1e3, bw, 5 * nb / bw / 1e3))
    tf += f * cnt; tb += bw * cnt
print('per step: fwd %.2f ms  bwd %.2f ms' % (tf / 1e3, tb / 1e3))
