"""(test infrastructure: imports oracle/ - hence under tests/, not tools/)
ROIAlign forward on BASELINE configs[1]: time + bit-exact check against the NumPy oracle on a RoI subset."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
from oracle import roi_align as ora
dev = torch.device('cuda:0')
lib = _hip.lib()
x, yx, gy = config2_inputs()
N, C, H, W = x.shape
R, _, PH, PW = gy.shape
xt = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
xy = yx[:, [0, 2, 1, 4, 3]].copy()
rois_xy = torch.from_numpy(xy).to(dev)
y = torch.empty((R, C, PH, PW), device=dev).contiguous(memory_format=torch.channels_last)
algo = 4 * (N * C * H * W + R * C * PH * PW) + 20 * R


def fwd():
    _hip.check(lib.mrcnn_roi_align_fwd_f32(_hip.ptr(xt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(y), _hip.stream_ptr()))


fwd()
torch.cuda.synchronize()
sel = np.arange(0, R, 16)
want = ora.roi_align_fwd(x, xy[sel], 7, 7, 0.25, 2)
got = y.cpu().numpy()[sel]
print('bit-exact on %d RoIs:' % len(sel), np.array_equal(got, want), ' max diff %.3e' % np.abs(got - want).max())
for _ in range(10): fwd()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): fwd()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 50 * 1e3)
print('configs[1] fwd: min %.1f us median %.1f us  %.0f GB/s algorithmic (%.3f of 8 TB/s)' % (min(ts), np.median(ts), algo / min(ts) / 1e3, algo / min(ts) / 1e3 / 8000))

# ---- experiment: the same RoIs pre-sorted spatially on the host (y band major, x minor): what would an XCD-local order buy?
for name, key in (('sorted by y centre', (xy[:, 2] + xy[:, 4])), ('sorted by (32-cell y band, x)', np.floor((xy[:, 2] + xy[:, 4]) / 2 * 0.25 / 32) * 4096 + (xy[:, 1] + xy[:, 3]) / 2 * 0.25)):
    order = np.argsort(key, kind='stable')
    rois_s = torch.from_numpy(xy[order].copy()).to(dev)

    def fwd_s():
        _hip.check(lib.mrcnn_roi_align_fwd_f32(_hip.ptr(xt), 1, N, C, H, W, _hip.ptr(rois_s), R, PH, PW, 0.25, 2, _hip.ptr(y), _hip.stream_ptr()))
    for _ in range(10): fwd_s()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): fwd_s()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print('configs[1] fwd, RoIs %s: min %.1f us' % (name, min(ts)))
