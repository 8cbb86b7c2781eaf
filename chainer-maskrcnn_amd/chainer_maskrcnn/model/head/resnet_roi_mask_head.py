"""ResNet res5 box + mask head (the C4 head of the Mask R-CNN paper) on gfx950 kernels - legacy variant.

Mirror of chainer_maskrcnn/model/head/resnet_roi_mask_head.py:11-73: ROIAlign 7x7 on the res4 map -> ResNet-50 res5
block with stride 1 (:27-29; training-mode BatchNorm over the R*49 pooled pixels) -> ReLU -> conv 3x3 (2048) + ReLU ->
{global average pooling -> ``cls_loc`` (n_class*4, per-class boxes here) and ``score``} and
{deconv 2x2/2 (256) + ReLU -> conv 3x3 (n_class - 1)} = mask (R, n_class-1, 14, 14).  Forward only.
"""
import numpy as np
import torch

from chainer_maskrcnn._hip import ops
from chainer_maskrcnn.nn.core import Conv, Bottleneck, ParamStore, normal
from chainer_maskrcnn.model.head.fpn_roi_mask_head import roi_align_fpn_fwd


class ResnetRoIMaskHead(object):
    mask_size = 14

    def __init__(self, n_class, roi_size, spatial_scale, loc_initialW=None, score_initialW=None, mask_initialW=None, ps=None,
                 prefix='head', width_div=1):
        self.ps = ps if ps is not None else ParamStore()
        self.n_class, self.roi_size, self.spatial_scale = n_class, roi_size, spatial_scale
        p = prefix + '/'
        d = width_div
        cin, mid, cout = 1024 // d, 512 // d, 2048 // d
        self.res5 = [Bottleneck(self.ps, p + 'res5/a', cin, mid, cout, 1, True),           # stride forced to 1 (:28-29)
                     Bottleneck(self.ps, p + 'res5/b1', cout, mid, cout, 1, False),
                     Bottleneck(self.ps, p + 'res5/b2', cout, mid, cout, 1, False)]
        self.conv1 = Conv(self.ps, p + 'conv1', cout, cout, 3, 1, 1, relu=True)
        mi = normal(0.01 if mask_initialW is None else mask_initialW)
        cd = 256 // d
        self.deconv1 = Conv(self.ps, p + 'deconv1', cout, 4 * cd, 1, bias=False, init=mi)
        self.ps.register(p + 'deconv1/b', (cd,), lambda rs: np.zeros((cd,), np.float32))
        self.deconv_b = p + 'deconv1/b'
        self.conv2 = Conv(self.ps, p + 'conv2', cd, n_class - 1, 3, 1, 1, init=mi)
        self.cls_loc = Conv(self.ps, p + 'cls_loc', cout, n_class * 4, init=normal(0.001 if loc_initialW is None else loc_initialW))
        self.score = Conv(self.ps, p + 'score', cout, n_class, init=normal(0.01 if score_initialW is None else score_initialW))

    def __call__(self, x, rois, roi_indices, spatial_scale):
        x = x[0] if isinstance(x, (tuple, list)) else x
        xy5 = torch.cat((roi_indices.to(torch.float32)[:, None], rois[:, [1, 0, 3, 2]]), dim=1).contiguous()
        lv = torch.zeros((rois.shape[0],), dtype=torch.int32, device=rois.device)
        h = roi_align_fpn_fwd([x], xy5, lv, self.roi_size, [spatial_scale])
        for b in self.res5:
            h, _ = b.fwd(h)                       # a bottleneck ends in ReLU: the extra F.relu of :62 is the identity
        h, _ = self.conv1.fwd(h)
        R = h.shape[0]
        gap = ops.global_avg_pool(h).view(R, 1, 1, -1)
        locs, _ = self.cls_loc.fwd(gap)
        scores, _ = self.score.fwd(gap)
        d, _ = self.deconv1.fwd(h)
        up = ops.relu(ops.pixel_shuffle2x(d, bias=self.ps.p(self.deconv_b)))
        m, _ = self.conv2.fwd(up)
        return (locs.view(R, -1)[:, :self.n_class * 4], scores.view(R, -1)[:, :self.n_class],
                m[..., :self.n_class - 1].permute(0, 3, 1, 2))
