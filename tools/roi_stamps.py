"""Where does a wave of the ROIAlign backward kernel spend its cycles?  Diagnostic build with s_memtime stamps
(mrcnn_debug_roi_align_bwd_stamps) on BASELINE configs[1]."""
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
st = torch.zeros((nwg * 4, 12), dtype=torch.int64, device=dev)


def run():
    _hip.check(lib.mrcnn_debug_roi_align_bwd_stamps(_hip.ptr(gyt), N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(gx),
                                                    _hip.ptr(st), _hip.stream_ptr()))


for _ in range(20):
    run()
st.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record()
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
s = s[s[:, 5] > 0]
print('kernel %.1f us by events; %d waves stamped' % (e0.elapsed_time(e1) * 1e3, len(s)))
t0 = s[:, 0].min()
clk = (s[:, 5] - s[:, 0]).sum() / ((s[:, 7] - s[:, 6]).sum() * 10.0)      # cycles per ns: memrealtime ticks at 100 MHz
print('in-kernel clock ~ %.2f GHz' % clk)
q = lambda a: 'min %7.0f  p10 %7.0f  med %7.0f  p90 %7.0f  max %7.0f' % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print('wave start  (cycles after first)   ', q(s[:, 0] - t0))
print('wave end    (cycles after first)   ', q(s[:, 5] - t0))
print('wave lifetime                      ', q(s[:, 5] - s[:, 0]))
print('  scan (4 groups of 128 RoIs)      ', q(s[:, 1]))
print('  tables + queue build             ', q(s[:, 2]))
print('  drain                            ', q(s[:, 3]))
print('  stores issue .. all stores done  ', q(s[:, 5] - s[:, 4]))
print('  [preamble before the table fill] ', q(s[:, 8]))
print('  [table passes, total]            ', q(s[:, 9]))
print('  [build minus preamble and tables]', q(s[:, 2] - s[:, 8] - s[:, 9]))
print('  units (candidates)               ', q(np.mod(s[:, 10], 65536)))
print('  table passes                     ', q(np.floor(s[:, 10] / 65536)))
print('  entries in the last drain        ', q(s[:, 11]))
rt = (s[:, 7].max() - s[:, 6].min()) * 0.01
print('first wave start .. last wave end: %.1f us (memrealtime)' % rt)
