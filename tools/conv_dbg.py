import os, sys
R = '/root/repo' if os.path.exists('/root/repo/chainer-maskrcnn_amd') else os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd'))
import torch
from chainer_maskrcnn._hip import nn as hnn, lib, check, ptr, stream_ptr
dev = torch.device('cuda:0')
N, H, W, Ci, Co, k, p = 2, 256, 256, 256, 256, 3, 1
x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, k, k, Ci), device=dev) * 0.05
b = torch.zeros((Co,), device=dev); y = torch.empty((N, H, W, Co), device=dev)
fl = 2.0 * N * H * W * k * k * Ci * Co
def run(flag):
    check(lib().mrcnn_conv2d_fwd_f32(ptr(x), ptr(w), ptr(b), ptr(y), N, H, W, Ci, Co, k, k, 1, p, 1 | (flag << 4), None, 0, stream_ptr()))
for flag, name in ((0, 'full'), (2, 'loads issued, never stored (no wait)'), (1, 'no global loads'), (3, 'no global loads, no LDS stores')):
    for _ in range(3): run(flag)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run(flag)
    e1.record(); torch.cuda.synchronize()
    print('%-50s %.1f TF/s' % (name, fl / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12))
