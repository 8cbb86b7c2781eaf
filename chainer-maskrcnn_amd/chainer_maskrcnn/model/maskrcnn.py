"""MaskRCNN model assembly on gfx950 kernels.

Mirror of chainer_maskrcnn/model/maskrcnn.py:23-155,261-276 (constructor :26-133, __call__ :135-155,
prepare :261-276) and of the ChainerCV ``FasterRCNN`` base-class attributes the train chain and
train.py rely on (loc_normalize_mean/std, use_preset, n_class).  Only the
combination the reference can actually train is built: backbone 'fpn' with head_arch 'fpn' or
'fpn_keypoint' (README.md:39); the other strings raise ValueError like the reference does for unknown
names.  ``predict`` / ``_suppress`` are SURVEY.md section 8f "next" rows.
"""
import numpy as np
import torch

from chainer_maskrcnn.nn.core import ParamStore
from chainer_maskrcnn._hip import ops
from .extractor.feature_pyramid_network import FeaturePyramidNetwork
from .rpn.multilevel_region_proposal_network import MultilevelRegionProposalNetwork
from .head.fpn_roi_mask_head import FPNRoIMaskHead


class MaskRCNN(object):
    feat_stride = 16

    def __init__(self, n_fg_class, n_keypoints=None, n_mask_convs=None, pretrained_model=None, min_size=600,
                 max_size=1000, ratios=[0.5, 1, 2], anchor_scales=[8], rpn_initialW=None, loc_initialW=None,
                 score_initialW=None, proposal_creator_params={}, backbone='fpn', head_arch='fpn',
                 device='cuda', seed=1234, _test_shrink=None):
        if n_fg_class is None:
            raise ValueError('The n_fg_class needs to be supplied as an argument')
        self.ps = ParamStore()
        shrink = _test_shrink or {}
        if backbone == 'fpn':
            self.extractor = FeaturePyramidNetwork(self.ps, **shrink)
            self.rpn = MultilevelRegionProposalNetwork(
                anchor_scales=self.extractor.anchor_scales, feat_strides=self.extractor.feat_strides,
                in_channels=self.extractor.out_channels, mid_channels=self.extractor.out_channels,
                proposal_creator_params=proposal_creator_params, ps=self.ps)
        elif backbone in ('c4', 'darknet'):
            raise ValueError('backbone %r is outside the MI355X hot path (SURVEY.md section 2.1); only \'fpn\' is built'
                             % backbone)
        else:
            raise ValueError('unknown backbone: {}'.format(backbone))
        c = self.extractor.out_channels
        if head_arch == 'fpn':
            self.head = FPNRoIMaskHead(n_fg_class + 1, roi_size_box=7, roi_size_mask=14, loc_initialW=loc_initialW,
                                       score_initialW=score_initialW, mask_initialW=0.01, ps=self.ps, in_channels=c,
                                       fc_channels=1024 // shrink.get('width_div', 1))
            self.predict_mask = True
        elif head_arch == 'fpn_keypoint':
            if n_keypoints is None:
                raise ValueError('n_keypoints must be set in keypoint detection')
            from .head.fpn_roi_keypoint_head import FPNRoIKeypointHead
            self.head = FPNRoIKeypointHead(2, n_keypoints, roi_size_box=7, roi_size_mask=14,
                                           n_mask_convs=8 if n_mask_convs is None else n_mask_convs,
                                           loc_initialW=loc_initialW, score_initialW=score_initialW, mask_initialW=0.01,
                                           ps=self.ps, in_channels=c, fc_channels=1024 // shrink.get('width_div', 1))
            self.predict_mask = False
        elif head_arch in ('res5', 'light'):
            raise ValueError('head_arch %r is outside the MI355X hot path (SURVEY.md section 2.1)' % head_arch)
        else:
            raise ValueError('unknown head archtecture specified. {}'.format(head_arch))
        # FasterRCNN base-class state
        self.mean = np.array([122.7717, 115.9465, 102.9801], dtype=np.float32)[:, None, None]   # unused (maskrcnn.py:273-274)
        self.min_size, self.max_size = min_size, max_size
        self.loc_normalize_mean = (0., 0., 0., 0.)
        self.loc_normalize_std = (0.1, 0.1, 0.2, 0.2)
        self.use_preset('visualize')
        self.train = True
        self.device = torch.device(device)
        self.ps.materialise(self.device, seed)

    @property
    def n_class(self):
        return self.head.n_class

    def use_preset(self, preset):
        if preset == 'visualize':
            self.nms_thresh, self.score_thresh = 0.3, 0.7
        elif preset == 'evaluate':
            self.nms_thresh, self.score_thresh = 0.3, 0.05
        else:
            raise ValueError('preset must be visualize or evaluate')

    def to_nhwc4(self, x):
        """(N,3,H,W) float32 images on the device -> the extractor's (N,H,W,4) operand."""
        return ops.image_nchw3_to_nhwc4(x.contiguous())

    def __call__(self, x, scale=1.):
        """Reference forward (:135-155).  x (N,3,H,W) on the device."""
        img_size = tuple(x.shape[2:])
        h = self.extractor(self.to_nhwc4(x))
        self.rpn.train = self.train
        rpn_locs, rpn_scores, rois, roi_indices, anchor, levels = self.rpn(h, img_size, scale)
        levels = levels.clamp(0, len(h) - 1)
        indices_and_rois = torch.cat((roi_indices.to(torch.float32)[:, None], rois), dim=1)
        if self.train:
            roi_cls_locs, roi_scores, mask = self.head(h, indices_and_rois, levels, self.extractor.spatial_scales)
            return roi_cls_locs, roi_scores, rois, roi_indices, mask
        roi_cls_locs, roi_scores = self.head(h, indices_and_rois, levels, self.extractor.spatial_scales, train=False)
        return roi_cls_locs, roi_scores, rois, roi_indices, levels

    def prepare_size(self, H, W):
        """Scaled size of maskrcnn.py:261-271 (min side -> min_size unless the max side would exceed max_size)."""
        scale = self.min_size / min(H, W)
        if scale * max(H, W) > self.max_size:
            scale = self.max_size / max(H, W)
        return int(H * scale), int(W * scale)
