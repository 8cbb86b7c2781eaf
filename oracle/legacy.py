"""Oracle: the legacy model variants (TEST INFRASTRUCTURE - see oracle/__init__.py), forward only, float64 torch-CPU +
the NumPy ROIAlign oracle.  Restates
  model/extractor/c4_backbone.py:7-26   ResNet-50 conv1..res4, pool1 = max_pooling_2d(3, stride 2) [cover_all], (res4,)
  model/extractor/darknet.py:6-60       5 x (conv3x3 + BN(train) + ReLU), 2x2/2 max pooling (cover_all) after the first 4
  model/head/light_roi_mask_head.py:11-127   separable 15x1/1x15 pairs (no activation), ROIAlign 7x7, fc + ReLU,
                                        cls_loc / score, mask = deconv1_(pool) (the three mask convs are dead code upstream)
  model/head/resnet_roi_mask_head.py:11-73   ROIAlign 7x7 -> res5 (stride 1) -> conv3x3 + ReLU -> GAP -> cls_loc / score;
                                        mask = conv2(relu(deconv1(h)))
Third-party pieces ("parity unpinned"): Chainer ResNet50Layers / BuildingBlock, F.max_pooling_2d cover_all output size
(size + 2p - k + s - 1) // s + 1, L.Deconvolution2D, _global_average_pooling_2d.
Parameters: dict name -> float64 tensor in the product's storage convention (conv weight (Cout_p, KH, KW, Cin_p))."""
import numpy as np
import torch
import torch.nn.functional as F

from . import model as om


def conv_rect(x, w, b, pad_hw):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=1, padding=pad_hw)
    return y.permute(0, 2, 3, 1)


def maxpool_cover_all(x, k, s):
    """F.max_pooling_2d(x, k, stride=s), cover_all=True, pad 0: windows may hang over the bottom / right edge."""
    return F.max_pool2d(x.permute(0, 3, 1, 2), k, s, ceil_mode=True).permute(0, 2, 3, 1)


def bottleneck(p, x, pre, stride, project):
    h = F.relu(om.bn_train(om.conv(x, p[pre + '/conv1/W'], None, stride), p[pre + '/bn1/gamma'], p[pre + '/bn1/beta']))
    h = F.relu(om.bn_train(om.conv(h, p[pre + '/conv2/W'], None, 1, 1), p[pre + '/bn2/gamma'], p[pre + '/bn2/beta']))
    h = om.bn_train(om.conv(h, p[pre + '/conv3/W']), p[pre + '/bn3/gamma'], p[pre + '/bn3/beta'])
    r = om.bn_train(om.conv(x, p[pre + '/conv4/W'], None, stride), p[pre + '/bn4/gamma'], p[pre + '/bn4/beta']) if project else x
    return F.relu(h + r)


def c4_backbone(p, img4, stage_blocks, prefix='extractor'):
    e = prefix + '/'
    h = F.relu(om.bn_train(om.conv(img4, p[e + 'conv1/W'], p[e + 'conv1/b'], 2, 3), p[e + 'bn1/gamma'], p[e + 'bn1/beta']))
    h = maxpool_cover_all(h, 3, 2)
    for name, n, stride in zip(('res2', 'res3', 'res4'), stage_blocks, (1, 2, 2)):
        h = bottleneck(p, h, e + '%s/a' % name, stride, True)
        for i in range(1, n):
            h = bottleneck(p, h, e + '%s/b%d' % (name, i), 1, False)
    return h


def darknet(p, img4, prefix='extractor'):
    h = img4
    for i in range(5):
        pre = '%s/conv%d' % (prefix, i + 1)
        h = F.relu(om.bn_train(om.conv(h, p[pre + '/c/W'], p[pre + '/c/b'], 1, 1), p[pre + '/bn/gamma'], p[pre + '/bn/beta']))
        if i < 4:
            h = maxpool_cover_all(h, 2, 2)
    return h


def _pool(x, rois_yx, roi_indices, P, scale):
    xy5 = np.concatenate([np.asarray(roi_indices, np.float32)[:, None], np.asarray(rois_yx, np.float32)[:, [1, 0, 3, 2]]], 1)
    return om.roi_align_fpn([x], xy5, np.zeros(len(xy5), np.int64), P, [scale])


def _deconv2x2(h, w, b):
    d = om.conv(h, w)
    N, H, W, C4 = d.shape
    C = C4 // 4
    return d.reshape(N, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * H, 2 * W, C) + b


def light_head(p, x, rois_yx, roi_indices, scale, n_class, k=15, prefix='head'):
    h_ = prefix + '/'
    q = k // 2
    left = conv_rect(conv_rect(x, p[h_ + 'conv_ul/W'], p[h_ + 'conv_ul/b'], (q, 0)), p[h_ + 'conv_bl/W'], p[h_ + 'conv_bl/b'], (0, q))
    right = conv_rect(conv_rect(x, p[h_ + 'conv_ur/W'], p[h_ + 'conv_ur/b'], (0, q)), p[h_ + 'conv_br/W'], p[h_ + 'conv_br/b'], (q, 0))
    tfp = left + right
    pool = _pool(tfp, rois_yx, roi_indices, 7, scale)
    R = pool.shape[0]
    h = F.relu(om.conv(pool.reshape(R, 1, 1, -1), p[h_ + 'fc/W'], p[h_ + 'fc/b']))
    locs = om.conv(h, p[h_ + 'cls_loc/W'], p[h_ + 'cls_loc/b']).reshape(R, -1)[:, :4]
    scores = om.conv(h, p[h_ + 'score/W'], p[h_ + 'score/b']).reshape(R, -1)[:, :n_class]
    mask = _deconv2x2(pool, p[h_ + 'deconv1_/W'], p[h_ + 'deconv1_/b'])[..., :n_class - 1].permute(0, 3, 1, 2)
    return locs, scores, mask


def res5_head(p, x, rois_yx, roi_indices, scale, n_class, prefix='head'):
    h_ = prefix + '/'
    h = _pool(x, rois_yx, roi_indices, 7, scale)
    h = bottleneck(p, h, h_ + 'res5/a', 1, True)
    h = bottleneck(p, h, h_ + 'res5/b1', 1, False)
    h = bottleneck(p, h, h_ + 'res5/b2', 1, False)
    h = F.relu(om.conv(h, p[h_ + 'conv1/W'], p[h_ + 'conv1/b'], 1, 1))
    R = h.shape[0]
    gap = h.mean(dim=(1, 2)).reshape(R, 1, 1, -1)
    locs = om.conv(gap, p[h_ + 'cls_loc/W'], p[h_ + 'cls_loc/b']).reshape(R, -1)[:, :n_class * 4]
    scores = om.conv(gap, p[h_ + 'score/W'], p[h_ + 'score/b']).reshape(R, -1)[:, :n_class]
    up = F.relu(_deconv2x2(h, p[h_ + 'deconv1/W'], p[h_ + 'deconv1/b']))
    mask = om.conv(up, p[h_ + 'conv2/W'], p[h_ + 'conv2/b'], 1, 1)[..., :n_class - 1].permute(0, 3, 1, 2)
    return locs, scores, mask
