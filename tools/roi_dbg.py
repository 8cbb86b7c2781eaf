import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import numpy as np, torch, ctypes
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
x, yx, gy = config2_inputs()
dev = torch.device('cuda:0')
N, C, H, W = x.shape; R_, _, PH, PW = gy.shape
rois = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
gx = torch.empty((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last)
lev = torch.zeros((R_,), dtype=torch.int32, device=dev)
lib = _hip.lib()
arr = (ctypes.c_void_p * 1)(gx.data_ptr()); Hs = (ctypes.c_int * 1)(H); Ws = (ctypes.c_int * 1)(W); sc = (ctypes.c_float * 1)(0.25)
def run(flag):
    _hip.check(lib.mrcnn_roi_align_fpn_bwd_f32(_hip.ptr(gyt), arr, Hs, Ws, sc, 1, N, C, _hip.ptr(rois), _hip.ptr(lev), R_, PH, PW, 2, flag, None, 0, _hip.stream_ptr()))
for flag, name in ((0, 'full'), (2, 'no phase B'), (6, 'no list, no phase B'), (14, 'no tables, no list, no phase B')):
    for _ in range(10): run(flag)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run(flag)
    e1.record(); torch.cuda.synchronize()
    print('%-40s %.2f us' % (name, e0.elapsed_time(e1) * 1e3 / 50))
