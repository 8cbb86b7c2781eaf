#!/usr/bin/env python3
"""Golden vectors of the LEGACY model variants made by executing the reference's own classes (build container only).

Run from the repo root:   python tests/golden/make_legacy_reference.py        (needs /root/reference; ~1 min)

Executed from /root/reference, unmodified, in this process (forward only - the reference's train chain for these variants,
model/maskrcnn_train_chain.py:74, is broken upstream):
    C4Backbone.__init__ / functions / __call__          chainer_maskrcnn/model/extractor/c4_backbone.py:7-26
    ConvBatch, Darknet.__init__ / __call__              chainer_maskrcnn/model/extractor/darknet.py:6-60
    LightRoIMaskHead.__init__ / __call__ (train mode)   chainer_maskrcnn/model/head/light_roi_mask_head.py:11-113
    ResnetRoIMaskHead.__init__ / __call__               chainer_maskrcnn/model/head/resnet_roi_mask_head.py:11-73
    _roi_align_2d_yx                                    chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7

on the float64 stand-ins of the Chainer primitives (tests/golden/mini_chainer.py: Convolution2D with rectangular kernels,
Linear, Deconvolution2D, training-mode BatchNormalization, ResNet50Layers / BuildingBlock incl. the ``functions`` /
``links()`` / ``__call__(x, layers)`` protocol C4Backbone relies on, max_pooling_2d with cover_all,
_global_average_pooling_2d) and this repo's ROIAlign oracle in place of the absent submodule.  What the fixture pins is the
WIRING of the four classes: the 3x3 / stride-2 cover_all stem pooling of C4Backbone and its stop after res4, Darknet's
conv -> BN -> ReLU units with four poolings, the two separable 15x1 / 1x15 paths without activation and the discarded mask
convolutions of the light head (mask = deconv1_(pool)), res5 at stride 1 -> ReLU -> conv1 + ReLU -> {GAP -> cls_loc, score}
and conv2(relu(deconv1(h))) of the res5 head.  Weights: tests/golden/weights.py (seeded, Chainer layouts, assigned by
snapshot key).  Only data is stored: legacy_reference.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = '/root/reference'

import make_step_reference as ms        # noqa: E402  (install(): the stand-in packages; assign(): load_npz semantics)
import mini_chainer as mc               # noqa: E402
from weights import legacy_chainer_weights          # noqa: E402
from oracle import roi_align as oroi    # noqa: E402

SEEDS = {'c4': 31001, 'darknet': 31002, 'light': 31003, 'res5': 31004}
N_CLASS = 6
LIGHT_IN = 256          # the light head on a Darknet-sized (256-channel) map
ROIS = np.array([[8.0, 12.0, 120.0, 150.0], [0.0, 0.0, 159.0, 191.0], [40.5, 60.25, 90.0, 100.0], [100.0, 20.0, 158.0, 80.0],
                 [30.0, 100.0, 70.0, 180.0], [64.0, 64.0, 96.0, 96.0]], np.float32)       # (y1, x1, y2, x2) on a 160 x 192 image


def main():
    ms.install()
    sys.path.insert(0, REF)
    from chainer_maskrcnn.model.extractor.c4_backbone import C4Backbone
    from chainer_maskrcnn.model.extractor.darknet import Darknet
    from chainer_maskrcnn.model.head.light_roi_mask_head import LightRoIMaskHead
    from chainer_maskrcnn.model.head.resnet_roi_mask_head import ResnetRoIMaskHead
    import chainer_maskrcnn.functions.roi_align_2d_yx as ref_yx
    for cls in (C4Backbone, Darknet, LightRoIMaskHead, ResnetRoIMaskHead):
        assert sys.modules[cls.__module__].__file__.startswith(REF)

    def roi_align_2d(x, rois_xy, outh, outw, spatial_scale):       # the absent submodule's operator: this repo's oracle
        return mc.V(oroi.roi_align_fwd(np.asarray(x, np.float32), np.asarray(rois_xy, np.float32), outh, outw, spatial_scale, 2))
    ref_yx.roi_align_2d = roi_align_2d

    out = {}
    t0 = time.time()
    mc.config.train = True
    # ---- C4Backbone: one 96 x 128 image
    img = np.random.RandomState(41).rand(1, 3, 96, 128).astype(np.float32)
    m = C4Backbone('auto')
    assert not hasattr(m, 'res5') and not hasattr(m, 'fc6')
    ms.assign(m, legacy_chainer_weights('c4', SEEDS['c4']))
    assert all(not l.update_enabled for l in m.links() if isinstance(l, mc.BatchNormalization))
    (res4,) = m(mc.V(img))
    assert res4.shape == (1, 1024, 6, 8)
    out['c4_img'], out['c4_res4'] = img, np.asarray(res4, np.float32)
    # ---- Darknet: one 80 x 112 image
    img = np.random.RandomState(42).rand(1, 3, 80, 112).astype(np.float32)
    m = Darknet()
    ms.assign(m, legacy_chainer_weights('darknet', SEEDS['darknet']))
    (h,) = m(mc.V(img))
    assert h.shape == (1, 256, 5, 7) and Darknet.feat_strides == [16] and m.anchor_scales == [4.0]
    out['darknet_img'], out['darknet_out'] = img, np.asarray(h, np.float32)
    # ---- heads: a 10 x 12 map of stride 16, six RoIs of one image
    idx = np.zeros(len(ROIS), np.int32)
    x = np.random.RandomState(43).standard_normal((1, LIGHT_IN, 10, 12)).astype(np.float32)
    m = LightRoIMaskHead(N_CLASS, 7)
    ms.assign(m, legacy_chainer_weights('light', SEEDS['light'], n_class=N_CLASS, in_channels=LIGHT_IN))
    locs, scores, mask = m(mc.V(x), ROIS, idx, 1. / 16)
    assert locs.shape == (6, 4) and scores.shape == (6, N_CLASS) and mask.shape == (6, N_CLASS - 1, 14, 14)
    out.update(light_x=x, light_locs=np.asarray(locs, np.float32), light_scores=np.asarray(scores, np.float32),
               light_mask=np.asarray(mask, np.float32))
    x = np.random.RandomState(44).standard_normal((1, 1024, 10, 12)).astype(np.float32)
    m = ResnetRoIMaskHead(N_CLASS, 7, 1. / 16)
    assert m.res5.a.conv1.stride == (1, 1) and m.res5.a.conv4.stride == (1, 1)
    ms.assign(m, legacy_chainer_weights('res5', SEEDS['res5'], n_class=N_CLASS))
    locs, scores, mask = m(mc.V(x), ROIS[:4], idx[:4], 1. / 16)
    assert locs.shape == (4, N_CLASS * 4) and scores.shape == (4, N_CLASS) and mask.shape == (4, N_CLASS - 1, 14, 14)
    out.update(res5_x=x, res5_locs=np.asarray(locs, np.float32), res5_scores=np.asarray(scores, np.float32),
               res5_mask=np.asarray(mask, np.float32))
    out['rois_yx'] = ROIS
    out['seeds'] = np.array([SEEDS[k] for k in ('c4', 'darknet', 'light', 'res5')], np.int64)
    out['n_class'], out['light_in'] = np.int64(N_CLASS), np.int64(LIGHT_IN)
    path = os.path.join(HERE, 'legacy_reference.npz')
    np.savez_compressed(path, **out)
    print('legacy_reference.npz %.2f MB in %.0f s' % (os.path.getsize(path) / 2 ** 20, time.time() - t0))


if __name__ == '__main__':
    main()
