"""Single-level RegionProposalNetwork (the 'c4' backbone's RPN) on gfx950 kernels - legacy variant.

The reference builds ChainerCV's ``RegionProposalNetwork(1024, 516, ratios, anchor_scales, feat_stride=16, ...)`` for
``backbone='c4'`` (chainer_maskrcnn/model/maskrcnn.py:60-69): conv 3x3 (in -> mid) + ReLU, ``loc`` 1x1 (4A) and ``score``
1x1 (2A), anchors of ``generate_anchor_base(ratios, anchor_scales)`` at stride 16, proposals ranked by the SOFTMAX
foreground probability (the multi-level RPN of this repo ranks by the raw logit).  Returns ChainerCV's 5-tuple
``(rpn_locs, rpn_scores, rois, roi_indices, anchor)``.  Built on the multi-level implementation with one level.
"""
import torch

from chainer_maskrcnn._hip import ops
from chainer_maskrcnn.utils.anchors import generate_anchor_base
from .multilevel_region_proposal_network import MultilevelRegionProposalNetwork


class RegionProposalNetwork(MultilevelRegionProposalNetwork):
    def __init__(self, in_channels=512, mid_channels=512, ratios=[0.5, 1, 2], anchor_scales=[8, 16, 32], feat_stride=16,
                 initialW=None, proposal_creator_params=dict(), ps=None, prefix='rpn'):
        if len(anchor_scales) != 1:
            # the fused (loc, score) head is sized for A = len(ratios) anchors per position; MaskRCNN passes [8]
            raise ValueError('RegionProposalNetwork: one anchor scale per position on this path (got %r)' % (anchor_scales,))
        super().__init__(anchor_scales=[anchor_scales[0]], feat_strides=[feat_stride], in_channels=in_channels,
                         mid_channels=mid_channels, ratios=ratios, initialW=initialW,
                         proposal_creator_params=proposal_creator_params, ps=ps, prefix=prefix)
        self.anchor_bases = [generate_anchor_base(anchor_scales=anchor_scales, ratios=ratios)]
        self.feat_stride = feat_stride

    def __call__(self, x, img_size, scale=1.):
        """x: (N,H,W,C) NHWC feature map (or a one-element tuple of it)."""
        xs = x if isinstance(x, (tuple, list)) else (x,)
        N = xs[0].shape[0]
        dev = xs[0].device
        anchors = self.anchors_for([(xs[0].shape[1], xs[0].shape[2])], dev)
        A_tot = anchors.shape[0]
        locs = torch.empty((N, A_tot, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((N, A_tot, 2), dtype=torch.float32, device=dev)
        h, _ = self.conv.fwd(xs[0])
        o, _ = self.head.fwd(h)
        ops.rpn_pack(o, self.n_anchor, locs, scores, 0)
        pl = self.proposal_layer
        n_pre = pl.n_train_pre_nms if self.train else pl.n_test_pre_nms
        n_post = pl.n_train_post_nms if self.train else pl.n_test_post_nms
        out = ops.rpn_proposals(locs, ops.softmax2(scores), anchors, img_size, pl.min_size * scale, n_pre, n_post, pl.nms_thresh)
        valid = out['roi_indices'] >= 0
        return locs, scores, out['rois'][valid], out['roi_indices'][valid], anchors
