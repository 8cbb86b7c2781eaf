"""Oracle: the whole FPN Mask R-CNN training step on the CPU (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates, in float64 PyTorch-CPU autograd + the NumPy ROIAlign oracle, the forward of
  model/extractor/feature_pyramid_network.py:46-71 (+ Chainer ResNet50Layers),
  model/rpn/multilevel_region_proposal_network.py:126-152, model/head/fpn_roi_mask_head.py:55-88,
and the five losses of model/fpn_maskrcnn_train_chain.py:81-106 / train.py:50-58, so that the
device step's losses AND every parameter gradient can be checked (convolutions are floating-point
kernels => a torch reference is the checker, per the task statement; tolerance 1e-3 relative).

Inputs that are not differentiated (proposals -> sampled RoIs, labels, regression / mask targets,
anchor labels) are passed in: their device implementations are pinned separately against
oracle/proposal.py and oracle/targets.py (tests/test_rpn_gpu.py, tests/test_targets_gpu.py).

PINNED (wiring): tests/test_step_reference_cpu.py compares this module - five losses to 1e-6, p2-p6, RPN outputs and head
outputs to <= 1e-6 - with a training-step forward executed by the reference's own MaskRCNN / FeaturePyramidNetwork /
MultilevelRegionProposalNetwork / FPNRoIMaskHead / FPNMaskRCNNTrainChain code on float64 stand-ins of the Chainer primitives
(tests/golden/make_step_reference.py -> tests/golden/step_reference.npz).  The third-party arithmetic stays unpinned.

Weights use the device's storage convention: conv weight (Cout, KH, KW, Cin) with zero-padded
channels, activations NHWC; names are the ParamStore names of the product.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import roi_align as ora

D = torch.float64


def set_dtype(dt):
    """float64 for parity checks (default); float32 when timed as the CPU baseline (the reference computes in fp32)."""
    global D
    D = dt


def conv(x, w, b=None, stride=1, pad=0):
    """x (N,H,W,Cin) NHWC, w (Cout,KH,KW,Cin) -> (N,Ho,Wo,Cout)."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1)


def bn_train(x, gamma, beta, eps=2e-5):
    C = x.shape[-1]
    xr = x.reshape(-1, C)
    mean = xr.mean(0)
    var = xr.var(0, unbiased=False)
    return gamma * (x - mean) / torch.sqrt(var + eps) + beta


def maxpool_cover_all(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)


def upsample_add(top, lat):
    H, W = lat.shape[1:3]
    return top.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, :H, :W] + lat


def resize_images_x2(x):
    """Chainer F.resize_images to twice the size [3P-RECALL, SURVEY.md App. A-8]: corner-aligned bilinear,
    u = linspace(0, H-1, 2H), u0 = clip(floor(u), 0, H-2), u1 = u0+1, weights (u1-u), (u-u0).  x NHWC."""
    N, H, W, C = x.shape

    def axis(n):
        u = np.linspace(0, n - 1, 2 * n)
        u0 = np.clip(np.floor(u).astype(np.int64), 0, max(n - 2, 0))
        u1 = np.minimum(u0 + 1, n - 1)
        return torch.from_numpy(u0), torch.from_numpy(u1), torch.from_numpy(u0 + 1 - u).to(x.dtype), torch.from_numpy(u - u0).to(x.dtype)
    y0, y1, wy0, wy1 = axis(H)
    x0, x1, wx0, wx1 = axis(W)
    rows = x[:, y0] * wy0[None, :, None, None] + x[:, y1] * wy1[None, :, None, None]
    return rows[:, :, x0] * wx0[None, None, :, None] + rows[:, :, x1] * wx1[None, None, :, None]


class _RoIAlignNp(torch.autograd.Function):
    """oracle.roi_align on one level (NCHW NumPy, float32 arithmetic as specified) inside autograd."""

    @staticmethod
    def forward(ctx, x, rois, P, scale):
        xn = x.detach().permute(0, 3, 1, 2).numpy().astype(np.float32)
        y = ora.roi_align_fwd(xn, rois, P, P, scale, 2)
        ctx.meta = (rois, tuple(xn.shape), scale)
        return torch.from_numpy(y).to(D).permute(0, 2, 3, 1)

    @staticmethod
    def backward(ctx, gy):
        rois, shape, scale = ctx.meta
        g = ora.roi_align_bwd(gy.permute(0, 3, 1, 2).numpy().astype(np.float32), rois, shape, scale, 2)
        return torch.from_numpy(g).to(D).permute(0, 2, 3, 1), None, None, None


def roi_align_fpn(feats, rois_xy5, levels, P, scales):
    """Multi-level pooling = the per-RoI loops of fpn_roi_mask_head.py:59-61,75-77."""
    R = rois_xy5.shape[0]
    C = feats[0].shape[-1]
    out = torch.zeros((R, P, P, C), dtype=D)
    for l, f in enumerate(feats):
        idx = np.nonzero(levels == l)[0]
        if idx.size:
            out[torch.from_numpy(idx)] = _RoIAlignNp.apply(f, rois_xy5[idx], P, scales[l])
    return out


def smooth_l1_loss(x, t, label, sigma):
    """ChainerCV _fast_rcnn_loc_loss."""
    w = (label > 0).to(D)[:, None]
    s2 = sigma ** 2
    d = w * (x - t)
    ad = d.abs()
    flag = (ad < 1.0 / s2).to(D)
    y = flag * (s2 / 2.0) * d * d + (1 - flag) * (ad - 0.5 / s2)
    return y.sum() / max(float((label >= 0).sum()), 1.0)


def softmax_ce(x, t):
    valid = t != -1
    if valid.sum() == 0:
        return x.sum() * 0
    return F.cross_entropy(x[valid], t[valid].long(), reduction='sum') / float(valid.sum())


class OracleStep(object):
    """params: dict name -> float64 torch tensor (requires_grad) in the product's storage convention."""

    def __init__(self, params, stage_blocks, n_class, loc0, feat_strides=(4, 8, 16, 32, 64), n_anchor=3,
                 mask_conv_names=('mask1', 'mask2', 'mask3', 'mask4'), n_keypoints=None, bn_buffers=None, tap=None):
        """bn_buffers: None = training-mode BatchNorm (batch statistics, chainer.config.train True); a dict
        name -> tensor of the running statistics ('.../avg_mean', '.../avg_var') = inference mode (maskrcnn.py:171-172).
        tap: name of a bottleneck ('extractor/resnet/res5/b2'): its 3x3 convolution's input and output are kept in
        ``self.taps`` (the output with retain_grad) so a test can feed them to the device's filter-gradient kernel."""
        self.p = params
        self.bn_buffers, self.tap, self.taps = bn_buffers, tap, {}
        self.mask_conv_names, self.n_keypoints = mask_conv_names, n_keypoints
        self.stage_blocks = stage_blocks
        self.n_class, self.loc0, self.n_anchor = n_class, loc0, n_anchor
        self.scales = [1.0 / s for s in feat_strides]

    def _bn(self, x, pre, eps=2e-5):
        p = self.p
        if self.bn_buffers is None:
            return bn_train(x, p[pre + '/gamma'], p[pre + '/beta'], eps)
        mean, var = self.bn_buffers[pre + '/avg_mean'].to(D), self.bn_buffers[pre + '/avg_var'].to(D)
        return p[pre + '/gamma'] * (x - mean) / torch.sqrt(var + eps) + p[pre + '/beta']

    def _bottleneck(self, x, pre, stride, project):
        p = self.p
        h1 = F.relu(self._bn(conv(x, p[pre + '/conv1/W'], None, stride), pre + '/bn1'))
        y2 = conv(h1, p[pre + '/conv2/W'], None, 1, 1)
        if self.tap == pre:
            if y2.requires_grad:
                y2.retain_grad()
            self.taps[pre + '/conv2'] = (h1, y2)
        h = F.relu(self._bn(y2, pre + '/bn2'))
        h = self._bn(conv(h, p[pre + '/conv3/W']), pre + '/bn3')
        if project:
            r = self._bn(conv(x, p[pre + '/conv4/W'], None, stride), pre + '/bn4')
        else:
            r = x
        return F.relu(h + r)

    def extractor(self, img4):
        p = self.p
        e = 'extractor/'
        h = F.relu(self._bn(conv(img4, p[e + 'resnet/conv1/W'], p[e + 'resnet/conv1/b'], 2, 3), e + 'resnet/bn1'))
        h = maxpool_cover_all(h)
        cs = []
        for name, n, stride in zip(('res2', 'res3', 'res4', 'res5'), self.stage_blocks, (1, 2, 2, 2)):
            h = self._bottleneck(h, e + 'resnet/%s/a' % name, stride, True)
            for i in range(1, n):
                h = self._bottleneck(h, e + 'resnet/%s/b%d' % (name, i), 1, False)
            cs.append(h)
        c2, c3, c4, c5 = cs
        cv = lambda n, x, s=1, pd=0: conv(x, p[e + n + '/W'], p[e + n + '/b'], s, pd)
        p5 = cv('toplayer', c5)
        p4 = cv('conv_p4', upsample_add(p5, cv('lat_p4', c4)), 1, 1)
        p3 = cv('conv_p3', upsample_add(p4, cv('lat_p3', c3)), 1, 1)
        p2 = cv('conv_p2', upsample_add(p3, cv('lat_p2', c2)), 1, 1)
        p6 = cv('conv_p6', p5, 2)
        return [p2, p3, p4, p5, p6]

    def rpn(self, feats):
        p, A = self.p, self.n_anchor
        locs, scores = [], []
        for f in feats:
            h = F.relu(conv(f, p['rpn/conv/W'], p['rpn/conv/b'], 1, 1))
            o = conv(h, p['rpn/loc_score/W'], p['rpn/loc_score/b'])
            n = o.shape[0]
            locs.append(o[..., :4 * A].reshape(n, -1, 4))
            scores.append(o[..., 4 * A:6 * A].reshape(n, -1, 2))
        return torch.cat(locs, 1), torch.cat(scores, 1)

    def head_box(self, feats, rois_xy5, levels):
        p = self.p
        pool = roi_align_fpn(feats, rois_xy5, levels, 7, self.scales)
        h = F.relu(conv(pool, p['head/conv1/W'], p['head/conv1/b'], 1, 1))
        R = h.shape[0]
        h = h.reshape(R, 1, 1, -1)
        h = F.relu(conv(h, p['head/fc1/W'], p['head/fc1/b']))
        h = F.relu(conv(h, p['head/fc2/W'], p['head/fc2/b']))
        return conv(h, p['head/score_cls_loc/W'], p['head/score_cls_loc/b']).reshape(R, -1)

    def head_mask(self, feats, rois_xy5, levels):
        p = self.p
        h = roi_align_fpn(feats, rois_xy5, levels, 14, self.scales)
        for nm in self.mask_conv_names:
            h = F.relu(conv(h, p['head/%s/W' % nm], p['head/%s/b' % nm], 1, 1))
        d = conv(h, p['head/deconv1/W'])
        N, H, W, C4 = d.shape
        C = C4 // 4
        up = d.reshape(N, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * H, 2 * W, C) + p['head/deconv1/b']
        m = conv(up, p['head/conv2/W'], p['head/conv2/b'])
        return resize_images_x2(m) if self.n_keypoints else m

    def losses(self, img4, t):
        """t: dict of NumPy targets taken from the device step (see tests/test_step_gpu.py)."""
        feats = self.extractor(img4)
        locs, scores = self.rpn(feats)
        n, A = locs.shape[:2]
        rl = torch.from_numpy(t['gt_rpn_label'].reshape(-1))
        l_rpn_loc = smooth_l1_loss(locs.reshape(n * A, 4), torch.from_numpy(t['gt_rpn_loc'].reshape(-1, 4)).to(D), rl, 3.0)
        l_rpn_cls = softmax_ce(scores.reshape(n * A, 2), rl)
        box = self.head_box(feats, t['rois_xy5'], t['sample_levels'])
        lab = torch.from_numpy(t['gt_roi_label'])
        l_roi_loc = smooth_l1_loss(box[:, self.loc0:self.loc0 + 4], torch.from_numpy(t['gt_roi_loc']).to(D), lab, 1.0)
        l_roi_cls = softmax_ce(box[:, :self.n_class], lab)
        m = self.head_mask(feats, t['mask_rois_xy5'], t['mask_levels'])
        ml = t['mask_label']
        gt = t['gt_roi_mask']
        rows = np.nonzero(ml > 0)[0]
        if self.n_keypoints:           # train_keypoints.py:21-27: softmax CE over the H*W positions of each (RoI, keypoint)
            K = self.n_keypoints
            x = m[torch.from_numpy(rows)][..., :K].permute(0, 3, 1, 2).reshape(len(rows) * K, -1)
            l_mask = softmax_ce(x, torch.from_numpy(gt[rows].reshape(-1)))
            return dict(rpn_loc_loss=l_rpn_loc, rpn_cls_loss=l_rpn_cls, roi_loc_loss=l_roi_loc, roi_cls_loss=l_roi_cls,
                        mask_loss=l_mask, feats=feats, locs=locs, scores=scores, box=box, mask=m)
        sel = m[torch.from_numpy(rows), :, :, torch.from_numpy(ml[rows] - 1)]        # calc_mask_loss: channel label-1
        tt = torch.from_numpy(gt[rows]).to(D)
        valid = torch.from_numpy(gt[rows] != -1)
        l_mask = F.binary_cross_entropy_with_logits(sel[valid], tt[valid], reduction='sum') / max(float(valid.sum()), 1.0)
        return dict(rpn_loc_loss=l_rpn_loc, rpn_cls_loss=l_rpn_cls, roi_loc_loss=l_roi_loc, roi_cls_loss=l_roi_cls,
                    mask_loss=l_mask, feats=feats, locs=locs, scores=scores, box=box, mask=m)
