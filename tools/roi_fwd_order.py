"""ROIAlign forward on BASELINE configs[1]: caller order against the map-order walk (ranking kernel + forward kernel), back to back."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
dev = torch.device('cuda:0')
lib = _hip.lib()
x, yx, gy = config2_inputs()
N, C, H, W = x.shape
R, _, PH, PW = gy.shape
xt = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
rois = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
y = torch.empty((R, PH, PW, C), device=dev)
nb = lib.mrcnn_roi_align_fwd_workspace_bytes(R)
ws = torch.empty((nb,), dtype=torch.uint8, device=dev)


def run(on, n=200):
    _hip.check(lib.mrcnn_roi_align_set_fwd_map_order(on))
    f = lambda: _hip.check(lib.mrcnn_roi_align_fwd_ws_f32(_hip.ptr(xt), 1, N, C, H, W, _hip.ptr(rois), R, PH, PW, 0.25, 2, _hip.ptr(y),
                                                          _hip.ptr(ws), nb, _hip.stream_ptr()))
    for _ in range(20):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rnd in range(3):
    print('round %d: caller order %.2f us | map order (ranking kernel + forward kernel) %.2f us' % (rnd, run(0), run(1)))
_hip.check(lib.mrcnn_roi_align_set_fwd_map_order(1))
