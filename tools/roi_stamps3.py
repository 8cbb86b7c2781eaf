"""Where does a wave of the table-driven ROIAlign backward (variant 3) spend its cycles?  Diagnostic build with s_memtime
stamps (mrcnn_debug_roi_align_bwd3_stamps) on BASELINE configs[1]."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
dev = torch.device('cuda:0')
lib = _hip.lib()
x, yx, gy = config2_inputs()
N, C, H, W = x.shape
R, _, PH, PW = gy.shape
rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
gx = torch.empty((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
tiles = ((H + 7) // 8) * ((W + 7) // 8) * N
nwg = 8 * ((tiles + 7) // 8)
st = torch.zeros((nwg * 4, 8), dtype=torch.int64, device=dev)
nb = lib.mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, PH, PW, 2)
ws = torch.empty((nb,), dtype=torch.uint8, device=dev)


def run():
    _hip.check(lib.mrcnn_debug_roi_align_bwd3_stamps(_hip.ptr(gyt), N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(gx),
                                                     _hip.ptr(ws), nb, _hip.ptr(st), _hip.stream_ptr()))


for _ in range(20):
    run()
st.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
s = s[s[:, 6] > 0]
print('both kernels %.1f us by events; %d waves stamped' % (e0.elapsed_time(e1) * 1e3, len(s)))
t0 = s[:, 0].min()
q = lambda a: 'min %7.0f  p10 %7.0f  med %7.0f  p90 %7.0f  max %7.0f' % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print('wave start  (cycles after first)   ', q(s[:, 0] - t0))
print('wave end    (cycles after first)   ', q(s[:, 6] - t0))
print('wave lifetime                      ', q(s[:, 6] - s[:, 0]))
print('  scan                             ', q(s[:, 1]))
print('  descriptors + prefix sum         ', q(s[:, 2]))
print('  entry generation (waits for tw)  ', q(s[:, 3]))
print('  drain                            ', q(s[:, 4]))
print('  stores issue .. done             ', q(s[:, 6] - s[:, 5]))
print('  before the scan (tile decode)    ', q(s[:, 5] - s[:, 0] - s[:, 1] - s[:, 2] - s[:, 3] - s[:, 4]))
print('  entries                          ', q(s[:, 7]))
heavy = s[np.argsort(s[:, 6] - s[:, 0])[-5:]]
for h in heavy:
    print('  slow wave: life %6.0f scan %5.0f desc %5.0f gen %5.0f drain %6.0f store %5.0f entries %4.0f' % (h[6] - h[0], h[1], h[2], h[3], h[4], h[6] - h[5], h[7]))
