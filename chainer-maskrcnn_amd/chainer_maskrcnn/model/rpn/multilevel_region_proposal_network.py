"""Multi-level RPN on gfx950 kernels: shared 3x3 conv + fused 1x1 (loc, score) head over the five
pyramid levels, then the whole proposal path on the device.

Mirror of chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:16-31 (level map),
:63-88 (constructor), :90-166 (forward).  Differences in mechanism, not in results:
  * the two 1x1 heads (:85-86) are ONE convolution with 6A (padded to 32) output channels written
    NHWC, so the transposes of :134,137 disappear; ``mrcnn_rpn_pack_f32`` produces the concatenated
    (N, A_total, 4) / (N, A_total, 2) arrays of :149-151;
  * ChainerCV's ProposalCreator + NMS (:156-158) run in ``mrcnn_rpn_proposals_f32`` without a
    device->host copy; outputs are padded to n_post rows per image with device-side counts.
    ``__call__`` returns exact-size arrays like the reference (one host sync); ``forward_padded``
    is the sync-free form the train chain uses.
"""
import contextlib

import numpy as np
import torch

from chainer_maskrcnn.nn.core import Conv, ParamStore, normal
from chainer_maskrcnn._hip import ops, nn as hnn

PACK_LEVELS_IN_ONE_LAUNCH = True        # (r6) one pack / unpack launch for all pyramid levels (A/B switch; same bits)
SMALL_LEVELS_BESIDE_P2 = False      # A/B switch, see forward_padded
from chainer_maskrcnn.utils.anchors import generate_anchor_base, enumerate_shifted_anchor


def map_rois_to_fpn_levels(rois, k_min=0, k_max=4):
    """rois (R,4) yx on a HIP device -> float32 levels, floor(4 + log2(sqrt(area)/224 + 1e-6)) clipped."""
    return ops.map_rois_to_fpn_levels(rois.contiguous(), k_min, k_max)


class ProposalCreatorParams(object):
    """Defaults of ChainerCV's ProposalCreator."""

    def __init__(self, nms_thresh=0.7, n_train_pre_nms=12000, n_train_post_nms=2000, n_test_pre_nms=6000,
                 n_test_post_nms=300, force_cpu_nms=False, min_size=16):
        self.nms_thresh, self.min_size = nms_thresh, min_size
        self.n_train_pre_nms, self.n_train_post_nms = n_train_pre_nms, n_train_post_nms
        self.n_test_pre_nms, self.n_test_post_nms = n_test_pre_nms, n_test_post_nms


class MultilevelRegionProposalNetwork(object):
    def __init__(self, anchor_scales, feat_strides, in_channels=256, mid_channels=256, ratios=[0.5, 1, 2],
                 initialW=None, proposal_creator_params=dict(), ps=None, prefix='rpn'):
        if len(anchor_scales) != len(feat_strides):
            raise ValueError('length of anchor_scales and feat_strides should be same!')
        self.anchor_bases = [generate_anchor_base(anchor_scales=[s], ratios=ratios) for s in anchor_scales]
        self.feat_strides = feat_strides
        self.proposal_layer = ProposalCreatorParams(**proposal_creator_params)
        self.n_anchor = A = self.anchor_bases[0].shape[0]
        self.ps = ps if ps is not None else ParamStore()
        init = None if initialW is None else normal(initialW)
        self.conv = Conv(self.ps, prefix + '/conv', in_channels, mid_channels, 3, 1, 1, relu=True, init=init, fwd_tile=0)
        # channels [0,4A) = loc (a*4+k), [4A,6A) = score (a*2+c): the reference's `loc` and `score` links fused
        self.head = Conv(self.ps, prefix + '/loc_score', mid_channels, 6 * A, 1, 1, 0, init=init)
        self.train = True
        self._anchor_cache = {}

    def anchors_for(self, shapes, device):
        key = (tuple(shapes), str(device))
        if key not in self._anchor_cache:
            a = [enumerate_shifted_anchor(self.anchor_bases[i], self.feat_strides[i], hh, ww)
                 for i, (hh, ww) in enumerate(shapes)]
            self._anchor_cache[key] = torch.from_numpy(np.concatenate(a, axis=0)).to(device)
        return self._anchor_cache[key]

    @staticmethod
    def per_image_params(N, img_size, scale, min_size, dev):
        """(N,3) f32 device tensor (h, w, min_size * scale) when the images of the batch differ in size or scale, else None.
        img_size: (H, W) or an (N,2) array / tensor of each image's own size inside the padded batch; scale: a number or N
        numbers (host or device).  No device->host copy is made."""
        sizes = img_size if torch.is_tensor(img_size) else np.asarray(img_size, np.float32)
        per_size = sizes.ndim == 2
        per_scale = (torch.is_tensor(scale) and scale.numel() > 1) or (not torch.is_tensor(scale) and np.ndim(scale) > 0 and np.size(scale) > 1)
        if not per_size and not per_scale:
            return None
        out = torch.empty((N, 3), dtype=torch.float32, device=dev)
        if per_size:
            out[:, :2] = (sizes if torch.is_tensor(sizes) else torch.from_numpy(np.ascontiguousarray(sizes))).to(dev, torch.float32)
        else:
            out[:, 0], out[:, 1] = float(sizes[0]), float(sizes[1])
        if per_scale:
            sc = scale if torch.is_tensor(scale) else torch.from_numpy(np.asarray(scale, np.float32))
            out[:, 2] = sc.to(dev, torch.float32).reshape(N) * float(min_size)
        else:
            out[:, 2] = float(min_size) * float(scale)
        return out

    def forward_padded(self, xs, img_size, scale=1., debug=False, after_heads=None, batch_size=None):
        """xs: NHWC pyramid levels.  Returns a dict: locs (N,A,4), scores (N,A,2), anchors (A,4) and the
        padded proposal outputs of ops.rpn_proposals (rois, roi_indices, levels, n_rois).
        ``after_heads(locs, scores, anchors)`` is called once the head outputs are enqueued and BEFORE the proposal
        kernels are: work that only needs the head outputs (anchor targets, RPN losses) can be put on another stream
        there and overlaps the latency-bound proposal chain (decode, radix sort, NMS: ~0.8 ms of tiny kernels)."""
        N = xs[0].shape[0]
        dev = xs[0].device
        shapes = [(x.shape[1], x.shape[2]) for x in xs]
        anchors = self.anchors_for(shapes, dev)
        Atot = anchors.shape[0]
        locs = torch.empty((N, Atot, 4), dtype=torch.float32, device=dev)
        scores = torch.empty((N, Atot, 2), dtype=torch.float32, device=dev)
        tape, a_off = [], 0
        # SMALL_LEVELS_BESIDE_P2 (off by default): the levels are independent and p3 .. p6 are latency-bound launches
        # that do not fill the chip - they run on the weight-gradient stream (idle in the forward pass) beside p2's two large kernels
        beside = SMALL_LEVELS_BESIDE_P2 and dev.type == 'cuda' and hnn.PROFILE is None and len(xs) > 1
        if beside:
            main, side = torch.cuda.current_stream(dev), hnn.side_stream(dev)
            side.wait_stream(main)
        outs = []
        one_pack = PACK_LEVELS_IN_ONE_LAUNCH and not beside and len(xs) <= 8
        for i, x in enumerate(xs):
            with (torch.cuda.stream(side) if (beside and i > 0) else contextlib.nullcontext()):
                h, c1 = self.conv.fwd(x)
                o, c2 = self.head.fwd(h)
                if one_pack:
                    outs.append(o)
                else:
                    ops.rpn_pack(o, self.n_anchor, locs, scores, a_off)
            if beside and i > 0:
                for t_ in tuple(c1) + tuple(c2):        # saved for the backward pass, which runs on other streams
                    if torch.is_tensor(t_):
                        t_.record_stream(main)
            tape.append((c1, c2, tuple(o.shape), a_off))
            a_off += x.shape[1] * x.shape[2] * self.n_anchor
        if beside:
            main.wait_stream(side)
        if one_pack:            # the concat of rpn/...:143-152 for all levels in one launch (five dependent 6-us launches before)
            ops.rpn_pack_levels(outs, self.n_anchor, locs, scores)
        self.tape = tape
        if after_heads is not None:
            after_heads(locs, scores, anchors)
        pl = self.proposal_layer
        n_pre = pl.n_train_pre_nms if self.train else pl.n_test_pre_nms
        n_post = pl.n_train_post_nms if self.train else pl.n_test_post_nms
        per_image = self.per_image_params(N, img_size, scale, pl.min_size, dev)
        if per_image is not None:       # the scalar arguments are unused then; pass the padded size
            img_size, scale = (batch_size or (xs[0].shape[1] * self.feat_strides[0], xs[0].shape[2] * self.feat_strides[0])), 1.0
        out = ops.rpn_proposals(locs, scores, anchors, img_size, pl.min_size * float(scale), n_pre, n_post, pl.nms_thresh,
                                debug=debug, per_image=per_image)
        out.update(locs=locs, scores=scores, anchors=anchors, n_post=n_post)
        return out

    def __call__(self, xs, img_size, scale=1.):
        """Reference signature (:90-166): (rpn_locs, rpn_scores, rois, roi_indices, anchor, levels)."""
        o = self.forward_padded(xs, img_size, scale)
        valid = o['roi_indices'] >= 0            # host sync: exact-size outputs like the reference
        return (o['locs'], o['scores'], o['rois'][valid], o['roi_indices'][valid], o['anchors'], o['levels'][valid])

    def backward(self, g_locs, g_scores, g_feats):
        """g_locs (N,A,4), g_scores (N,A,2): loss gradients.  g_feats: per-level gradient tensors that are
        accumulated into (they already hold the ROIAlign gradients); None = the RPN's own contribution is RETURNED as a
        list of new per-level tensors (the train chain runs this branch early, beside the proposal chain, and adds the
        contributions where this call used to be)."""
        first = True
        out = []
        g_os = None
        if PACK_LEVELS_IN_ONE_LAUNCH and len(self.tape) <= 8:
            g_os = ops.rpn_unpack_grad_levels(g_locs, g_scores, [t[2] for t in self.tape], self.n_anchor)
        for i, (c1, c2, oshape, a_off) in enumerate(self.tape):
            g_o = g_os[i] if g_os is not None else ops.rpn_unpack_grad(g_locs, g_scores, oshape, self.n_anchor, a_off)
            g_h = self.head.bwd(c2, g_o, accumulate_params=not first, mask_gx=True)     # + ReLU backward of self.conv
            out.append(self.conv.bwd(c1, g_h, gx_acc=None if g_feats is None else g_feats[i], accumulate_params=not first, gy_masked=True))
            first = False
        self.tape = None
        return out if g_feats is None else None
