"""The DEVICE legacy variants against forward passes executed by the reference's own classes (tests/golden/legacy_reference.npz,
see tests/test_legacy_reference_cpu.py for how it is made and how oracle/legacy.py is pinned by it): C4Backbone, Darknet,
LightRoIMaskHead and ResnetRoIMaskHead at FULL width on the fixture's inputs and seeded weights, conv activations within
1e-3 of the tensor scale (BASELINE.json north_star)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.nn import core  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_legacy_reference_cpu import legacy_params  # noqa: E402

DEV = 'cuda:0'


def _rel(got, want):
    want = np.asarray(want, np.float64)
    return float(np.abs(got.detach().double().cpu().numpy() - want).max()) / max(float(np.abs(want).max()), 1e-30)


def _materialise(kind, seed, **kw):
    part, ps, native = legacy_params(kind, seed, **kw)
    ps.materialise(torch.device(DEV), 1)
    for k, v in native.items():
        ps.p(k).copy_(torch.from_numpy(v))
    return part


def _nhwc(x, cp):
    t = torch.from_numpy(x).permute(0, 2, 3, 1).contiguous()
    if cp > t.shape[3]:
        t = torch.cat([t, torch.zeros(t.shape[:3] + (cp - t.shape[3],))], -1)
    return t.contiguous().to(DEV)


def test_device_legacy_variants_match_the_reference_executed_forward(golden_dir):
    d = np.load(os.path.join(golden_dir, 'legacy_reference.npz'))
    seeds = dict(zip(('c4', 'darknet', 'light', 'res5'), [int(v) for v in d['seeds']]))
    n_class, light_in = int(d['n_class']), int(d['light_in'])
    core.TRAIN = True
    m = _materialise('c4', seeds['c4'])
    (res4,) = m(_nhwc(d['c4_img'], 4))
    assert tuple(res4.shape) == (1, 6, 8, 1024) and _rel(res4.permute(0, 3, 1, 2), d['c4_res4']) < 1e-3
    m = _materialise('darknet', seeds['darknet'])
    (h,) = m(_nhwc(d['darknet_img'], 4))
    assert tuple(h.shape) == (1, 5, 7, 256) and _rel(h.permute(0, 3, 1, 2), d['darknet_out']) < 1e-3
    rois = torch.from_numpy(d['rois_yx']).to(DEV)
    idx = torch.zeros((rois.shape[0],), dtype=torch.int32, device=DEV)
    m = _materialise('light', seeds['light'], n_class=n_class, in_channels=light_in)
    m.train = True
    locs, scores, mask = m(_nhwc(d['light_x'], light_in), rois, idx, 1. / 16)
    assert _rel(locs, d['light_locs']) < 1e-3 and _rel(scores, d['light_scores']) < 1e-3 and _rel(mask, d['light_mask']) < 1e-3
    m = _materialise('res5', seeds['res5'], n_class=n_class)
    locs, scores, mask = m(_nhwc(d['res5_x'], 1024), rois[:4], idx[:4], 1. / 16)
    assert _rel(locs, d['res5_locs']) < 1e-3 and _rel(scores, d['res5_scores']) < 1e-3 and _rel(mask, d['res5_mask']) < 1e-3
