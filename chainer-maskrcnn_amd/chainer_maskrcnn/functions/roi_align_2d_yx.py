"""The (y, x)-ordered entry point of ROIAlign that the heads call.

ChainerCV keeps boxes as (y_min, x_min, y_max, x_max); the pooling operator takes (x1, y1, x2, y2).  The reference bridges
the two with a column permutation in chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7; this module is that bridge for the
HIP operator (functions/roi_align/roi_align_2d.py -> mrcnn_roi_align_fwd_f32 / _bwd_f32), with the operator's
``sampling_ratio`` exposed (default 2, DESIGN.md section 4 "Deliberate pins")."""
from chainer_maskrcnn.functions.roi_align import roi_align_2d as _operator

_COLUMNS_YX_TO_XY = [0, 2, 1, 4, 3]          # batch index stays first; the two corner pairs swap their members


def _roi_align_2d_yx(x, indices_and_rois, outh, outw, spatial_scale, sampling_ratio=2):
    """x (N, C, H, W); indices_and_rois (R, 5) rows (batch index, y1, x1, y2, x2) in image coordinates -> (R, C, outh, outw)."""
    if indices_and_rois.ndim != 2 or indices_and_rois.shape[1] != 5:
        raise ValueError('indices_and_rois must be (R, 5): (batch index, y1, x1, y2, x2); got %r' % (tuple(indices_and_rois.shape),))
    return _operator.roi_align_2d(x, indices_and_rois[:, _COLUMNS_YX_TO_XY], outh, outw, spatial_scale, sampling_ratio)
