"""configs[1] ROIAlign for rocprofv3 --pmc passes (few launches, no timing): forward + the SHIPPED backward (the fused wave kernel
k_roi_align_bwd_waves, what roi_align_2d(...).backward() launches).  With a second argument `planned`: the opt-in two-launch form instead (the
entry-list plan k_roi_align_bwd_waves<MODE 1> + k_roi_align_bwd_lean with the plan verified) - the kernel name k_roi_align_bwd_waves then means the
plan builder, so the two forms go to separate output directories."""
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
rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
y = torch.empty((R, C, PH, PW), device=dev).contiguous(memory_format=torch.channels_last)
gx = torch.empty_like(xt)
nbf = lib.mrcnn_roi_align_fwd_workspace_bytes(R)
wsf = torch.empty((max(nbf, 1),), dtype=torch.uint8, device=dev)       # the forward's map-order permutation
import ctypes
fused = not (len(sys.argv) > 2 and sys.argv[2] == 'planned')
Hs, Ws, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
gxp = (ctypes.c_void_p * 1)(gx.data_ptr())
pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, PH, PW, 0)
plan = torch.zeros((pb,), dtype=torch.uint8, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    _hip.check(lib.mrcnn_roi_align_fwd_ws_f32(_hip.ptr(xt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(y), _hip.ptr(wsf), nbf, _hip.stream_ptr()))
    if fused:
        _hip.check(lib.mrcnn_roi_align_bwd_f32(_hip.ptr(gyt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(gx), _hip.stream_ptr()))
        continue
    _hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, PH, PW, 2, 0, _hip.ptr(plan), pb, _hip.stream_ptr()))
    _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), gxp, Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, PH, PW, 2, 0, None, 0,
                                                       _hip.ptr(plan), pb, 1, _hip.stream_ptr()))
torch.cuda.synchronize()
