"""Light-Head R-CNN style box + mask head on gfx950 kernels - legacy variant.

Mirror of chainer_maskrcnn/model/head/light_roi_mask_head.py:11-127: two separable large-kernel paths
(15x1 -> 1x15 and 1x15 -> 15x1, 256 then 490 channels, NO activation, :97-99) summed into a thin feature map, ROIAlign
7x7, one FC (2048) + ReLU, ``cls_loc`` (4, class-agnostic) and ``score``; mask = ``deconv1_(pool)`` (2x2 / 2 to
n_class - 1 channels).  Quirk kept (:110-113, :124-127): the three 3x3 mask convolutions are evaluated and their result is
DISCARDED in the reference (``mask = self.deconv1_(pool)`` overwrites it) - their parameters exist here too (checkpoint
compatibility) and are not evaluated.  ``__call__(x, rois, roi_indices, spatial_scale)`` with rois (R,4) yx,
roi_indices (R,); training mode returns (roi_cls_locs, roi_scores, mask), inference mode (roi_cls_locs, roi_scores) and
keeps the thin map for ``predict_mask``.  Forward only.
"""
import numpy as np
import torch

from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr, ops
from chainer_maskrcnn._hip.nn import workspace, conv_out
from chainer_maskrcnn.nn.core import Conv, ParamStore, normal, lecun_normal, pad_to
from chainer_maskrcnn.model.head.fpn_roi_mask_head import roi_align_fpn_fwd


class RectConv(object):
    """Convolution2D with a (kh, kw) kernel and (ph, pw) padding, forward only (mrcnn_conv2d_fwd_rect_f32)."""

    def __init__(self, ps, name, cin, cout, ksize, pad):
        self.ps, self.name, self.cin, self.cout = ps, name, cin, cout
        (self.kh, self.kw), (self.ph, self.pw) = ksize, pad
        self.cin_p, self.cout_p = pad_to(cin, 32), pad_to(cout, 32)
        gen = lecun_normal(cin * self.kh * self.kw)((cout, self.kh, self.kw, cin))

        def w_init(rs):
            w = np.zeros((self.cout_p, self.kh, self.kw, self.cin_p), np.float32)
            w[:cout, :, :, :cin] = gen(rs)
            return w
        ps.register(name + '/W', (self.cout_p, self.kh, self.kw, self.cin_p), w_init)
        ps.register(name + '/b', (self.cout_p,), lambda rs: np.zeros((self.cout_p,), np.float32))

    def __call__(self, x):
        N, H, W, Cin = x.shape
        assert Cin == self.cin_p and x.is_contiguous()
        Ho, Wo = conv_out(H, self.kh, 1, self.ph), conv_out(W, self.kw, 1, self.pw)
        y = torch.empty((N, Ho, Wo, self.cout_p), dtype=torch.float32, device=x.device)
        nb = lib().mrcnn_conv2d_workspace_bytes(N, H, W, Cin, self.cout_p, self.kh, self.kw, 1, max(self.ph, self.pw))
        ws = workspace(nb, x.device) if nb else None
        check(lib().mrcnn_conv2d_fwd_rect_f32(ptr(x), ptr(self.ps.p(self.name + '/W')), ptr(self.ps.p(self.name + '/b')), ptr(y),
                                              N, H, W, Cin, self.cout_p, self.kh, self.kw, 1, self.ph, self.pw, 0, ptr(ws),
                                              ws.numel() if ws is not None else 0, stream_ptr()))
        return y


class LightRoIMaskHead(object):
    mask_size = 14

    def __init__(self, n_class, roi_size, loc_initialW=None, score_initialW=None, mask_initialW=None, ps=None, prefix='head',
                 in_channels=1024, k=15, c_mid=256, c_out=490, fc_channels=2048):
        self.ps = ps if ps is not None else ParamStore()
        self.n_class, self.roi_size = n_class, roi_size
        p = prefix + '/'
        q = k // 2
        self.conv_ul = RectConv(self.ps, p + 'conv_ul', in_channels, c_mid, (k, 1), (q, 0))
        self.conv_bl = RectConv(self.ps, p + 'conv_bl', c_mid, c_out, (1, k), (0, q))
        self.conv_ur = RectConv(self.ps, p + 'conv_ur', in_channels, c_mid, (1, k), (0, q))
        self.conv_br = RectConv(self.ps, p + 'conv_br', c_mid, c_out, (k, 1), (q, 0))
        self.c_out, self.c_out_p = c_out, pad_to(c_out, 32)
        # fc: Chainer flattens (R, C, 7, 7); here the pooled tensor is (R, 7, 7, C_p): weight columns in (h, w, c_p) order
        self.fc = Conv(self.ps, p + 'fc', self.c_out_p * roi_size * roi_size, fc_channels, relu=True,
                       init=lecun_normal(c_out * roi_size * roi_size))
        li = normal(0.001 if loc_initialW is None else loc_initialW)
        si = normal(0.01 if score_initialW is None else score_initialW)
        self.cls_loc = Conv(self.ps, p + 'cls_loc', fc_channels, 4, init=li)
        self.score = Conv(self.ps, p + 'score', fc_channels, n_class, init=si)
        mi = normal(0.01 if mask_initialW is None else mask_initialW)
        # the reference's dead mask convolutions (parameters only, see the module docstring)
        self.conv2 = Conv(self.ps, p + 'conv2', c_out, 256, 3, 1, 1, init=mi)
        self.conv3_ = Conv(self.ps, p + 'conv3_', 256, 256, 3, 1, 1, init=mi)
        self.conv4 = Conv(self.ps, p + 'conv4', 256, 256, 3, 1, 1, init=mi)
        # deconv1_: Deconvolution2D(c_out -> n_class - 1, 2x2 / 2) = 1x1 conv to 4 blocks of K_p channels + pixel shuffle
        K = n_class - 1
        self.K, self.K_p = K, pad_to(K, 32)
        idx = [ab * self.K_p + o for ab in range(4) for o in range(K)]
        self.deconv1_ = Conv(self.ps, p + 'deconv1_', c_out, 4 * K, 1, bias=False, init=mi, cout_p=4 * self.K_p, cout_index=idx)
        self.ps.register(p + 'deconv1_/b', (self.K_p,), lambda rs: np.zeros((self.K_p,), np.float32))
        self.deconv_b = p + 'deconv1_/b'
        self.train = True
        self.tfp = None

    def thin_feature_map(self, x):
        left = self.conv_bl(self.conv_ul(x))
        right = self.conv_br(self.conv_ur(x))
        return ops.add(left, right)

    def _pool(self, tfp, rois, roi_indices, spatial_scale):
        xy5 = torch.cat((roi_indices.to(torch.float32)[:, None], rois[:, [1, 0, 3, 2]]), dim=1).contiguous()
        lv = torch.zeros((rois.shape[0],), dtype=torch.int32, device=rois.device)
        return roi_align_fpn_fwd([tfp], xy5, lv, self.roi_size, [spatial_scale])

    def _mask(self, pool):
        d, _ = self.deconv1_.fwd(pool)
        return ops.pixel_shuffle2x(d, bias=self.ps.p(self.deconv_b))        # (R, 14, 14, K_p)

    def __call__(self, x, rois, roi_indices, spatial_scale):
        """x (N,H,W,C) NHWC (or the backbone's one-element tuple); rois (R,4) yx; roi_indices (R,)."""
        x = x[0] if isinstance(x, (tuple, list)) else x
        tfp = self.thin_feature_map(x)
        pool = self._pool(tfp, rois, roi_indices, spatial_scale)
        R = pool.shape[0]
        h, _ = self.fc.fwd(pool.view(R, 1, 1, -1))
        locs, _ = self.cls_loc.fwd(h)
        scores, _ = self.score.fwd(h)
        roi_cls_locs, roi_scores = locs.view(R, -1)[:, :4], scores.view(R, -1)[:, :self.n_class]
        if self.train:
            return roi_cls_locs, roi_scores, self._mask(pool)[..., :self.K].permute(0, 3, 1, 2)
        self.tfp = tfp
        return roi_cls_locs, roi_scores

    def predict_mask(self, rois, roi_indices, spatial_scale):
        pool = self._pool(self.tfp, rois, roi_indices, spatial_scale)
        return self._mask(pool)[..., :self.K].permute(0, 3, 1, 2)
