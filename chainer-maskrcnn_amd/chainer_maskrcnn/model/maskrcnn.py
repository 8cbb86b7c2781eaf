"""MaskRCNN model assembly on gfx950 kernels.

Mirror of chainer_maskrcnn/model/maskrcnn.py:23-155,261-276 (constructor :26-133, __call__ :135-155,
prepare :261-276) and of the ChainerCV ``FasterRCNN`` base-class attributes the train chain and
train.py rely on (loc_normalize_mean/std, use_preset, n_class).  Only the
combination the reference can actually train is backbone 'fpn' with head_arch 'fpn' or 'fpn_keypoint'
(README.md:39) - that is the hot path.  The legacy variants (backbone 'c4' / 'darknet', head_arch 'res5' / 'light';
SURVEY.md section 8 f-4) construct like in the reference and run forward through ``forward_legacy``.
``predict`` / ``_suppress`` are SURVEY.md section 8f "next" rows.
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
        elif backbone == 'c4':          # legacy (maskrcnn.py:60-69): ResNet-50 C4 + ChainerCV's single-level RPN
            from .extractor.c4_backbone import C4Backbone
            from .rpn.region_proposal_network import RegionProposalNetwork
            self.extractor = C4Backbone(pretrained_model, ps=self.ps, **shrink)
            wd = shrink.get('width_div', 1)
            self.rpn = RegionProposalNetwork(1024 // wd, 516 // wd, ratios=ratios, anchor_scales=anchor_scales,
                                             feat_stride=self.feat_stride, initialW=rpn_initialW,
                                             proposal_creator_params=proposal_creator_params, ps=self.ps)
        elif backbone == 'darknet':     # legacy (maskrcnn.py:70-75)
            from .extractor.darknet import Darknet
            self.extractor = Darknet(ps=self.ps)
            self.rpn = MultilevelRegionProposalNetwork(
                anchor_scales=self.extractor.anchor_scales, feat_strides=self.extractor.feat_strides, in_channels=256,
                proposal_creator_params={'n_test_pre_nms': 50, 'n_test_post_nms': 10}, ps=self.ps)
        else:
            raise ValueError('unknown backbone: {}'.format(backbone))
        c = self.extractor.out_channels
        self.head_arch = head_arch
        if head_arch == 'res5':         # legacy (maskrcnn.py:81-89)
            from .head.resnet_roi_mask_head import ResnetRoIMaskHead
            self.head = ResnetRoIMaskHead(n_fg_class + 1, roi_size=7, spatial_scale=1. / self.feat_stride,
                                          loc_initialW=loc_initialW, score_initialW=score_initialW, mask_initialW=0.01,
                                          ps=self.ps, width_div=shrink.get('width_div', 1))
            self.predict_mask = True
        elif head_arch == 'light':      # legacy (maskrcnn.py:91-98)
            from .head.light_roi_mask_head import LightRoIMaskHead
            self.head = LightRoIMaskHead(n_fg_class + 1, roi_size=7, loc_initialW=loc_initialW, score_initialW=score_initialW,
                                         mask_initialW=0.01, ps=self.ps, in_channels=c)
            self.predict_mask = True
        elif head_arch == 'fpn':
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

    def forward_legacy(self, x, scale=1.):
        """The legacy wiring (backbone 'c4' / 'darknet' with head 'res5' / 'light'): extractor -> RPN -> head with the
        signature those heads have, ``head(h, rois, roi_indices, spatial_scale)``.  The reference's own ``__call__`` cannot
        drive them (it passes the FPN head's arguments, maskrcnn.py:148-154, and unpacks six RPN outputs where ChainerCV's
        RPN returns five) - upstream they were driven by the legacy MaskRCNNTrainChain, which is broken."""
        img_size = tuple(x.shape[2:])
        h = self.extractor(self.to_nhwc4(x))
        self.rpn.train = self.train
        out = self.rpn(h, img_size, scale)
        rpn_locs, rpn_scores, rois, roi_indices = out[:4]
        self.head.train = self.train
        scale_ = self.extractor.spatial_scales[0] if hasattr(self.extractor, 'spatial_scales') else 1. / self.feat_stride
        res = self.head(h, rois, roi_indices, scale_)
        return tuple(res) + (rois, roi_indices)

    def __call__(self, x, scale=1.):
        """Reference forward (:135-155).  x (N,3,H,W) on the device."""
        if self.head_arch in ('res5', 'light'):
            raise TypeError('MaskRCNN.__call__ passes the FPN head signature (maskrcnn.py:148-154), which the %r head does not '
                            'have - the reference fails here too; use forward_legacy()' % self.head_arch)
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

    def prepare(self, img):
        """maskrcnn.py:261-276: resize so that the short side is min_size unless the long side would exceed max_size,
        then scale to [0,1] (no mean subtraction - SURVEY.md App. B-3).  img (3,H,W) float32 tensor with values 0..255
        on the device.  The resize is chainercv.transforms.resize = cv2.resize INTER_LINEAR on float32, here
        ``mrcnn_image_resize_f32`` (the same tap rule as the training Transform's device resize)."""
        _, H, W = img.shape
        oh, ow = self.prepare_size(H, W)
        return ops.image_resize_f32(img.to(self.device, torch.float32).contiguous(), oh, ow, 255.0)

    def predict(self, imgs):
        """maskrcnn.py:157-259: imgs = list of (3,H,W) float32 tensors (0..255).  Returns (masks, labels, scores) -
        lists with one entry per image: masks (D,H,W) bool, labels (D,) int32 in [0, n_fg_class-1], scores (D,) float32 -
        and keeps the boxes in ``self.last_bboxes``.  One device->host copy per image (the per-class keep counts)."""
        from chainer_maskrcnn.nn import core
        masks, labels, scores, bboxes = [], [], [], []
        keep_train, keep_core = self.train, core.TRAIN
        self.train, core.TRAIN = False, False
        try:
            for img in imgs:
                size = tuple(img.shape[1:])
                x = self.prepare(img.to(self.device))
                scale = x.shape[2] / size[1]
                roi_cls_locs, roi_scores, rois, roi_indices, levels = self.__call__(x[None].contiguous(), scale=scale)
                box_out = self.head.last_box_out
                cls_bbox, prob = ops.detect_decode(rois.contiguous(), box_out, self.n_class, self.head.LOC0, scale,
                                                   self.loc_normalize_mean, self.loc_normalize_std, size)
                self.last_rois, self.last_decoded = rois, (cls_bbox, prob)      # parity tests read these
                bbox, label, score, level = self._suppress(cls_bbox, prob, levels)
                D = bbox.shape[0]
                if D > 0 and self.predict_mask:
                    xy5 = torch.cat((torch.zeros((D, 1), device=self.device), bbox[:, [1, 0, 3, 2]] * scale), dim=1).contiguous()
                    m = self.head.mask_branch(self.head.x, xy5, level.to(torch.int32).contiguous(),
                                              self.extractor.spatial_scales)
                    mask = ops.mask_paste(m, label.contiguous(), bbox.contiguous(), size).bool()
                else:
                    mask = torch.zeros((D,) + size, dtype=torch.bool, device=self.device)
                masks.append(mask)
                labels.append(label)
                scores.append(score)
                bboxes.append(bbox)
        finally:
            self.train, core.TRAIN = keep_train, keep_core
        self.last_bboxes = bboxes
        return masks, labels, scores

    def _suppress(self, cls_bbox, prob, levels):
        """maskrcnn.py:278-312 on the device: for every foreground class l (skipping the LAST class when masks are
        predicted - the reference's off-by-one guard, :288-291): prob[:, l] > score_thresh, NMS(nms_thresh) in
        descending score order; results concatenated over classes, labels l-1."""
        l_end = self.n_class - 1 if self.predict_mask else self.n_class
        keep_idx, keep_cnt = ops.class_nms(cls_bbox, prob, 1, l_end, self.score_thresh, self.nms_thresh)
        cnt = keep_cnt.cpu().tolist()                   # the one host sync of predict()
        sel, lab = [], []
        for l in range(1, l_end):
            if cnt[l]:
                sel.append(keep_idx[l, :cnt[l]])
                lab.append(torch.full((cnt[l],), l - 1, dtype=torch.int32, device=prob.device))
        if not sel:
            z = torch.zeros((0,), dtype=torch.long, device=prob.device)
            return cls_bbox[z], z.to(torch.int32), prob[z, 0], levels[z]
        sel = torch.cat(sel).long()
        lab = torch.cat(lab)
        return cls_bbox[sel], lab, prob[sel, (lab + 1).long()], levels[sel]

    def prepare_size(self, H, W):
        """Scaled size of maskrcnn.py:261-271 (min side -> min_size unless the max side would exceed max_size)."""
        scale = self.min_size / min(H, W)
        if scale * max(H, W) > self.max_size:
            scale = self.max_size / max(H, W)
        return int(H * scale), int(W * scale)
