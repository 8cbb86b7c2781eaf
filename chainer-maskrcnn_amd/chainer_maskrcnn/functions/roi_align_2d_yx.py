"""Mirror of chainer_maskrcnn/functions/roi_align_2d_yx.py:1-7 of the reference."""
from .roi_align.roi_align_2d import roi_align_2d


def _roi_align_2d_yx(x, indices_and_rois, outh, outw, spatial_scale, sampling_ratio=2):
    # (idx, y1, x1, y2, x2) -> (idx, x1, y1, x2, y2)
    xy_indices_and_rois = indices_and_rois[:, [0, 2, 1, 4, 3]]
    return roi_align_2d(x, xy_indices_and_rois, outh, outw, spatial_scale, sampling_ratio)
