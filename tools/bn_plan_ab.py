"""BatchNorm at the ResNet-50 shapes of the 2 x 1024^2 step, direct C calls (no Python wrapper between the launches), HIP events:
the finalisation layout (mrcnn_debug_bn_plan fin_quads) and the row-block cap of the reduction passes, A/B in one process.
usage: bn_plan_ab.py [cap,quads ...]   default: 1024,4 1024,1 2048,1 4096,1"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import ptr, stream_ptr, check
dev = torch.device('cuda:0')
lib = _hip.lib()
SHAPES = [(524288, 64, 1), (131072, 64, 6), (131072, 256, 4), (32768, 128, 8), (32768, 512, 5), (8192, 256, 12),
          (8192, 1024, 7), (2048, 512, 6), (2048, 2048, 4)]       # (pixels, channels, count per step)
VARIANTS = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]] or [(1024, 4), (1024, 1), (2048, 1), (4096, 1)]


def timed(f, n=50):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {v: [0.0, 0.0, 0.0] for v in VARIANTS}
print('%8s %5s %3s  %s' % ('pixels', 'C', 'n', '   '.join('cap %4d q%d: fwd(2 launches) fwd(3) bwd us' % v for v in VARIANTS)))
for P, C, cnt in SHAPES:
    x = torch.randn((P, C), device=dev); gy = torch.randn((P, C), device=dev)
    y = torch.empty_like(x); gx = torch.empty_like(x)
    g = torch.ones((C,), device=dev); b = torch.zeros((C,), device=dev)
    m = torch.empty((C,), device=dev); s = torch.empty((C,), device=dev); gg = torch.empty((C,), device=dev); gb = torch.empty((C,), device=dev)
    rows = 2 * ((P + 127) // 128)            # the convolution epilogue's partial rows (128-row tiles, two wave rows each)
    part = torch.randn((rows, 2, C), device=dev)
    part[:, 1].abs_().add_(4.0 * 64)         # sums of squares well above mean^2 P: the exact-recompute branch stays off
    line = []
    for v in VARIANTS:
        check(lib.mrcnn_debug_bn_plan(*v))
        nb = lib.mrcnn_bn_workspace_bytes(P, C)
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
        st = stream_ptr()
        f2 = timed(lambda: lib.mrcnn_bn_train_fwd_stats_f32(ptr(x), ptr(part), rows, ptr(g), ptr(b), None, ptr(y), ptr(m), ptr(s), None, None, P, C, 2e-5, 0.9, 1, st))
        f3 = timed(lambda: lib.mrcnn_bn_train_fwd_f32(ptr(x), ptr(g), ptr(b), None, ptr(y), ptr(m), ptr(s), None, None, P, C, 2e-5, 0.9, 1, ptr(ws), nb, st))
        bw = timed(lambda: lib.mrcnn_bn_train_bwd_f32(ptr(gy), ptr(x), None, ptr(g), ptr(b), ptr(m), ptr(s), ptr(gx), None, ptr(gg), ptr(gb), P, C, 1, ptr(ws), nb, st))
        line.append('%22.1f %6.1f %6.1f' % (f2, f3, bw))
        for i, t in enumerate((f2, f3, bw)):
            tot[v][i] += t * cnt
    print('%8d %5d %3d  %s' % (P, C, cnt, '   '.join(line)))
print('per step (ms):       %s' % '   '.join('%22.3f %6.3f %6.3f' % tuple(t / 1e3 for t in tot[v]) for v in VARIANTS))
check(lib.mrcnn_debug_bn_plan(1024, 4))
