"""CPU pin of oracle/legacy.py (the float64 restatement the device's legacy variants are compared with in
tests/test_legacy_gpu.py) against forward passes EXECUTED BY THE REFERENCE'S OWN CLASSES - C4Backbone, Darknet,
LightRoIMaskHead, ResnetRoIMaskHead (tests/golden/make_legacy_reference.py -> legacy_reference.npz; float64 stand-ins of the
Chainer primitives, this repo's ROIAlign oracle for the absent submodule).  The weights are the seeded Chainer-layout arrays
of tests/golden/weights.py brought into the product's storage convention by ``legacy_native`` with the padded shapes of the
product's own layer objects, so the layout mapping of the legacy layers (rectangular kernels, the (c,h,w) -> (h,w,c_p) FC
re-ordering, the 2x2 deconvolution as a 1x1 convolution + pixel shuffle) is pinned by the same comparison."""
import os
import sys

import numpy as np
import torch

from chainer_maskrcnn.nn.core import ParamStore
from oracle import legacy as ol

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from weights import legacy_chainer_weights, legacy_native      # noqa: E402

D = torch.float64


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


def legacy_part(kind, n_class=6, in_channels=256):
    """(product layer object registered in a fresh ParamStore, padded shapes by name, name prefix)."""
    ps = ParamStore()
    if kind == 'c4':
        from chainer_maskrcnn.model.extractor.c4_backbone import C4Backbone
        part = C4Backbone(ps=ps)
    elif kind == 'darknet':
        from chainer_maskrcnn.model.extractor.darknet import Darknet
        part = Darknet(ps=ps)
    elif kind == 'light':
        from chainer_maskrcnn.model.head.light_roi_mask_head import LightRoIMaskHead
        part = LightRoIMaskHead(n_class, 7, ps=ps, in_channels=in_channels)
    else:
        from chainer_maskrcnn.model.head.resnet_roi_mask_head import ResnetRoIMaskHead
        part = ResnetRoIMaskHead(n_class, 7, 1. / 16, ps=ps)
    prefix = 'extractor/' if kind in ('c4', 'darknet') else 'head/'
    return part, ps, {n: s for n, (_, s) in ps.offsets.items()}, prefix


def legacy_params(kind, seed, n_class=6, in_channels=256):
    part, ps, shapes, prefix = legacy_part(kind, n_class, in_channels)
    kw = dict(n_class=n_class, in_channels=in_channels) if kind == 'light' else (dict(n_class=n_class) if kind == 'res5' else {})
    native = legacy_native(kind, legacy_chainer_weights(kind, seed, **kw), shapes, prefix, **kw)
    assert set(native) == set(shapes), set(native) ^ set(shapes)          # every product parameter is covered, nothing else
    return part, ps, native


def _img4(img):
    t = torch.from_numpy(img)
    return torch.cat([t.permute(0, 2, 3, 1), torch.zeros((1,) + t.shape[2:] + (1,))], -1).to(D)


def _nhwc_padded(x, cp):
    t = torch.from_numpy(x).permute(0, 2, 3, 1).to(D)
    return torch.cat([t, torch.zeros(t.shape[:3] + (cp - t.shape[3],), dtype=D)], -1) if cp > t.shape[3] else t


def test_oracle_legacy_equals_reference_executed_forward(golden_dir):
    d = np.load(os.path.join(golden_dir, 'legacy_reference.npz'))
    seeds = dict(zip(('c4', 'darknet', 'light', 'res5'), [int(v) for v in d['seeds']]))
    n_class, light_in = int(d['n_class']), int(d['light_in'])
    P = lambda native: {k: torch.from_numpy(v).to(D) for k, v in native.items()}
    with torch.no_grad():
        _, _, nat = legacy_params('c4', seeds['c4'])
        res4 = ol.c4_backbone(P(nat), _img4(d['c4_img']), (3, 4, 6))
        assert _rel(res4.permute(0, 3, 1, 2).numpy(), d['c4_res4']) < 1e-6
        _, _, nat = legacy_params('darknet', seeds['darknet'])
        h = ol.darknet(P(nat), _img4(d['darknet_img']))
        assert _rel(h.permute(0, 3, 1, 2).numpy(), d['darknet_out']) < 1e-6
        rois, idx = d['rois_yx'], np.zeros(len(d['rois_yx']), np.int32)
        _, _, nat = legacy_params('light', seeds['light'], n_class, light_in)
        locs, scores, mask = ol.light_head(P(nat), _nhwc_padded(d['light_x'], light_in), rois, idx, 1. / 16, n_class)
        assert _rel(locs.numpy(), d['light_locs']) < 1e-6 and _rel(scores.numpy(), d['light_scores']) < 1e-6
        assert _rel(mask.numpy(), d['light_mask']) < 1e-6
        _, _, nat = legacy_params('res5', seeds['res5'], n_class)
        locs, scores, mask = ol.res5_head(P(nat), _nhwc_padded(d['res5_x'], 1024), rois[:4], idx[:4], 1. / 16, n_class)
        assert _rel(locs.numpy(), d['res5_locs']) < 1e-6 and _rel(scores.numpy(), d['res5_scores']) < 1e-6
        assert _rel(mask.numpy(), d['res5_mask']) < 1e-6
