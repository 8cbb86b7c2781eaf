"""Box + keypoint head over FPN levels on gfx950 kernels.

Mirror of chainer_maskrcnn/model/head/fpn_roi_keypoint_head.py:10-111: the box branch of the mask head
(conv3x3 -> fc -> fc -> cls_loc / score, n_class = 2) and a keypoint branch of ``n_mask_convs`` (default 8,
model/maskrcnn.py:110-111) 3x3 convolutions, the 2x2/2 deconvolution, a 1x1 convolution to ``n_keypoints``
heat maps and a corner-aligned bilinear x2 resize to 56x56 (:76-81).  Same kernels and the same fused
mechanisms as FPNRoIMaskHead.

Waived quirk: when EVERY RoI of a call maps to one pyramid level the reference pools
the box branch from level 0 with spatial_scales[0] whatever that level is (:62-64).  Detecting it needs a
device->host copy of the levels; the batched multi-level ROIAlign always uses each RoI's own level.
"""
from .fpn_roi_mask_head import FPNRoIMaskHead


class FPNRoIKeypointHead(FPNRoIMaskHead):
    mask_size = 56

    def __init__(self, n_class, n_keypoints, roi_size_box, roi_size_mask, n_mask_convs=8, loc_initialW=None,
                 score_initialW=None, mask_initialW=None, ps=None, prefix='head', in_channels=256, fc_channels=1024):
        super().__init__(n_class, roi_size_box, roi_size_mask, loc_initialW=loc_initialW, score_initialW=score_initialW,
                         mask_initialW=mask_initialW, ps=ps, prefix=prefix, in_channels=in_channels,
                         fc_channels=fc_channels, n_mask_convs=n_mask_convs, mask_out_channels=n_keypoints,
                         upsample2x=True, mask_conv_names=['mask_convs/%d' % i for i in range(n_mask_convs)])
        self.n_keypoints = n_keypoints
