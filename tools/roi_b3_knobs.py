"""Variant-3 ROIAlign backward on configs[1] under its measurement knobs: LDS pad (resident workgroups per CU) x s_setprio."""
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
nb = lib.mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, PH, PW, 2)
ws = torch.empty((nb,), dtype=torch.uint8, device=dev)


def bwd():
    _hip.check(lib.mrcnn_roi_align_bwd_ws_f32(_hip.ptr(gyt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(gx), _hip.ptr(ws), nb, _hip.stream_ptr()))


def timed(n=50, rounds=5):
    for _ in range(5): bwd()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): bwd()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return min(out)


_hip.check(lib.mrcnn_roi_align_set_bwd_variant(3))
for pad in (0, 12, 20, 33, 44, 60):
    for prio in (0, 1):
        _hip.check(lib.mrcnn_debug_roi_align_bwd3_knobs(pad * 1024, prio))
        print('pad %2d KiB (LDS/wg %2d KiB) prio %d: %.1f us (both kernels)' % (pad, 20 + pad, prio, timed()), flush=True)
_hip.check(lib.mrcnn_debug_roi_align_bwd3_knobs(0, 0))
_hip.check(lib.mrcnn_roi_align_set_bwd_variant(2))
