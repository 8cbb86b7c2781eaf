"""``roi_align_2d`` - drop-in for the operator of the reference's un-vendored
``chainer_maskrcnn/functions/roi_align`` submodule (.gitmodules:1-3), imported by
chainer_maskrcnn/functions/roi_align_2d_yx.py:1.

    roi_align_2d(x, rois, outh, outw, spatial_scale) -> y

x    (N,C,H,W) float32 on a HIP device.  torch.channels_last memory (NHWC) takes the
     fast kernels; contiguous NCHW takes the generic strided kernels.
rois (R,5) float32 rows (batch_idx, x_min, y_min, x_max, y_max) in image pixels.
y    (R,C,outh,outw), same memory format as x.  Differentiable w.r.t. x.
``sampling_ratio`` (not in the reference signature; default 2, 0 = adaptive) is the
deliberate pin recorded in oracle/roi_align.py.
"""
import ctypes

import torch

from chainer_maskrcnn import _hip


def _layout_of(x):
    if x.dim() != 4:
        raise ValueError('roi_align_2d: x must be 4-D (N,C,H,W), got %s' % (tuple(x.shape),))
    if x.is_contiguous(memory_format=torch.channels_last) and not (x.shape[1] == 1 and x.is_contiguous()):
        return _hip.LAYOUT_NHWC
    if x.is_contiguous():
        return _hip.LAYOUT_NCHW
    return None


class _RoIAlign2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rois, outh, outw, spatial_scale, sampling_ratio):
        _hip.require_cuda(x, rois)
        if x.dtype != torch.float32 or rois.dtype != torch.float32:
            raise TypeError('roi_align_2d: float32 only')
        if rois.dim() != 2 or rois.shape[1] != 5:
            raise ValueError('roi_align_2d: rois must be (R,5)')
        layout = _layout_of(x)
        if layout is None:
            x = x.contiguous()
            layout = _hip.LAYOUT_NCHW
        rois = rois.contiguous()
        N, C, H, W = x.shape
        R = rois.shape[0]
        fmt = torch.channels_last if layout == _hip.LAYOUT_NHWC else torch.contiguous_format
        y = torch.empty((R, C, outh, outw), dtype=torch.float32, device=x.device, memory_format=fmt)
        from chainer_maskrcnn._hip.nn import workspace
        nb = _hip.lib().mrcnn_roi_align_fwd_workspace_bytes(R)
        ws = workspace(nb, x.device) if nb else None           # the map-order permutation of the RoIs
        _hip.check(_hip.lib().mrcnn_roi_align_fwd_ws_f32(
            _hip.ptr(x), layout, N, C, H, W, _hip.ptr(rois), R, outh, outw,
            float(spatial_scale), int(sampling_ratio), _hip.ptr(y), _hip.ptr(ws), ws.numel() if ws is not None else 0,
            _hip.stream_ptr()))
        ctx.save_for_backward(rois)
        ctx.meta = (layout, N, C, H, W, outh, outw, float(spatial_scale), int(sampling_ratio))
        return y

    @staticmethod
    def backward(ctx, gy):
        (rois,) = ctx.saved_tensors
        layout, N, C, H, W, outh, outw, scale, sr = ctx.meta
        fmt = torch.channels_last if layout == _hip.LAYOUT_NHWC else torch.contiguous_format
        gy = gy.contiguous(memory_format=fmt)
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=gy.device, memory_format=fmt)
        from chainer_maskrcnn._hip.nn import workspace
        nb = _hip.lib().mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, rois.shape[0], outh, outw, sr)
        ws = workspace(nb, gy.device) if nb else None          # RoI-split slabs of small maps
        _hip.check(_hip.lib().mrcnn_roi_align_bwd_ws_f32(
            _hip.ptr(gy), layout, N, C, H, W, _hip.ptr(rois), rois.shape[0], outh, outw, scale, sr,
            _hip.ptr(gx), _hip.ptr(ws), ws.numel() if ws is not None else 0, _hip.stream_ptr()))
        return gx, None, None, None, None, None


def roi_align_2d(x, rois, outh, outw, spatial_scale, sampling_ratio=2):
    return _RoIAlign2D.apply(x, rois, outh, outw, spatial_scale, sampling_ratio)


def roi_align_sample_tables(rois, H, W, outh, outw, spatial_scale, sampling_ratio, smax):
    """Device-computed sample indices/weights (verification hook, see include/mrcnn_hip.h)."""
    _hip.require_cuda(rois)
    rois = rois.contiguous()
    R = rois.shape[0]
    cnt = torch.zeros((R, 2), dtype=torch.int32, device=rois.device)
    idx = torch.empty((R, 2, smax, 2), dtype=torch.int32, device=rois.device)
    wgt = torch.empty((R, 2, smax, 2), dtype=torch.float32, device=rois.device)
    _hip.check(_hip.lib().mrcnn_roi_align_sample_tables(
        _hip.ptr(rois), R, H, W, outh, outw, float(spatial_scale), int(sampling_ratio), smax,
        _hip.ptr(cnt), _hip.ptr(idx), _hip.ptr(wgt), _hip.stream_ptr()))
    return cnt, idx, wgt
