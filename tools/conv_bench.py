"""Fixed-shape convolution microbench (TF/s per kind) for A/B-ing kernel variants in one process."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
SHAPES = [  # N, H, W, Cin, Cout, k, pad
    (2, 256, 256, 256, 256, 3, 1), (512, 14, 14, 256, 256, 3, 1), (2, 64, 64, 256, 256, 3, 1),
    (2, 64, 64, 1024, 256, 1, 0), (2, 64, 64, 256, 1024, 1, 0), (2, 128, 128, 128, 128, 3, 1),
    (2, 256, 256, 64, 256, 1, 0), (2, 256, 256, 64, 64, 3, 1), (2, 32, 32, 512, 512, 3, 1), (512, 1, 1, 12544, 1024, 1, 0),
]
if os.environ.get('CONV_BENCH_SET') == 'medium':
    SHAPES = [(2, 128, 128, 128, 128, 3, 1), (2, 64, 64, 256, 256, 3, 1), (2, 64, 64, 1024, 256, 1, 0), (2, 64, 64, 256, 1024, 1, 0),
              (2, 128, 128, 128, 512, 1, 0), (2, 128, 128, 512, 128, 1, 0), (2, 32, 32, 512, 512, 3, 1), (2, 32, 32, 512, 2048, 1, 0),
              (2, 32, 32, 2048, 512, 1, 0), (2, 256, 256, 64, 64, 3, 1), (2, 256, 256, 64, 256, 1, 0), (2, 256, 256, 256, 64, 1, 0),
              (2, 16, 16, 256, 256, 3, 1), (2, 128, 128, 256, 256, 3, 1)]
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
tot = {'fwd': [0, 0], 'bwd_data': [0, 0], 'bwd_filter': [0, 0]}
for (N, H, W, Ci, Co, k, p) in SHAPES:
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, k, k, Ci), device=dev) * 0.05
    b = torch.zeros((Co,), device=dev); gy = torch.randn((N, H, W, Co), device=dev)
    fl = 2.0 * N * H * W * k * k * Ci * Co
    t1 = timeit(lambda: hnn.conv2d_fwd_raw(x, w, b, 1, p, True))
    t2 = timeit(lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, p))
    t3 = timeit(lambda: hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, p, True))
    for kname, tt in (('fwd', t1), ('bwd_data', t2), ('bwd_filter', t3)):
        tot[kname][0] += fl; tot[kname][1] += tt
    print('%4dx%3dx%3d %5d->%5d k%d  fwd %6.1f  bwd_data %6.1f  bwd_filter %6.1f TF/s' % (N, H, W, Ci, Co, k, fl / t1 / 1e12, fl / t2 / 1e12, fl / t3 / 1e12))
print('aggregate', {k: round(v[0] / v[1] / 1e12, 1) for k, v in tot.items()})
