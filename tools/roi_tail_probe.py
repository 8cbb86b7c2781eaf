"""ROIAlign backward microbench time vs number of 8x8 tiles (is the 850-tile configs[1] map paying a second round?)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import numpy as np, torch
from chainer_maskrcnn import _hip
from tests.util import rand_rois_xy
dev = torch.device('cuda:0')
lib = _hip.lib()
for (H, W) in ((200, 272), (192, 256), (176, 272), (200, 304), (256, 256)):
    rs = np.random.RandomState(2)
    rois = torch.from_numpy(rand_rois_xy(rs, 512, 1, H, W, 0.25)).to(dev)
    gy = torch.randn((512, 7, 7, 256), device=dev)
    gx = torch.empty((1, H, W, 256), device=dev)
    def f():
        _hip.check(lib.mrcnn_roi_align_bwd_f32(_hip.ptr(gy), 1, 1, 256, H, W, _hip.ptr(rois), 512, 7, 7, 0.25, 2, _hip.ptr(gx), _hip.stream_ptr()))
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    tiles = ((H + 7) // 8) * ((W + 7) // 8)
    print('map %3dx%3d  tiles %4d  %.1f us' % (H, W, tiles, e0.elapsed_time(e1) / 200 * 1e3))
