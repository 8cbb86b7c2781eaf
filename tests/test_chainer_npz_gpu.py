"""SURVEY.md section 8 f-2 on the device: a Chainer-NPZ snapshot (train.py:99-101,134-137) drives the HIP model.

(1) a snapshot written from one device model makes a differently-initialised device model compute bitwise the same
    function (save_npz -> load_npz(strict=True));
(2) arrays in CHAINER's layouts - Convolution2D W (Cout,Cin,KH,KW), Linear W (out, C*H*W flattened channel-major),
    Deconvolution2D W (Cin,Cout,2,2) - filled with fresh random numbers and loaded with strict=False give, on the device,
    what Chainer's own semantics give: the head of head/fpn_roi_mask_head.py:55-85 restated here with plain torch-CPU
    NCHW operators (float64) on those same arrays + the NumPy ROIAlign oracle.  This is the check that the layout
    changes of the mapping (NHWC flattening of fc1, fused score/cls_loc, deconvolution as 1x1 convolution + pixel
    shuffle, composed deconv1*conv2) are right on the kernels that consume them, not only on a CPU round trip."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import roi_align as ora
from oracle import boxes as ob

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.nn import core  # noqa: E402
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap, save_npz, load_npz  # noqa: E402

DEV = 'cuda:0'
SHRINK = dict(stages=(2, 1, 1, 1), width_div=2)


def _model(seed):
    return MaskRCNN(n_fg_class=80, device=DEV, seed=seed, _test_shrink=SHRINK)


def test_snapshot_transfers_the_function_between_device_models(tmp_path):
    a, b = _model(1), _model(2)
    x = torch.from_numpy(np.random.RandomState(0).rand(1, 3, 128, 160).astype(np.float32)).to(DEV)
    core.TRAIN = True
    for m in (a, b):
        m.train = True
    ya = a(x, 1.0)
    yb = b(x, 1.0)
    assert not torch.equal(ya[1], yb[1])
    path = str(tmp_path / 'model_5000.npz')
    save_npz(path, a)
    assert len(load_npz(path, b, strict=True)) > 100
    yb = b(x, 1.0)
    for u, v in zip(ya, yb):
        assert torch.equal(u, v)


def test_chainer_layout_arrays_give_chainer_semantics_on_the_device():
    m = _model(3)
    h = m.head
    rs = np.random.RandomState(5)
    inventory = ChainerNpzMap(m).to_chainer()
    arrays = {k: (rs.standard_normal(v.shape) * (0.5 / np.sqrt(max(np.prod(v.shape[1:]), 1)))).astype(np.float32)
              for k, v in inventory.items() if k.startswith('head/')}
    for k in arrays:
        if k.endswith('/b'):
            arrays[k] = (rs.standard_normal(arrays[k].shape) * 0.1).astype(np.float32)
    loaded = ChainerNpzMap(m).from_chainer(arrays, strict=False)
    assert set(loaded) == set(arrays)
    # a small pyramid and RoIs on several levels
    c = h.channels
    shapes = [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32., 1 / 64.]
    feats = [rs.standard_normal((1, c) + s).astype(np.float32) for s in shapes]
    rois = np.array([[4, 6, 40, 50], [0, 0, 127, 159], [30, 20, 100, 140], [60, 80, 75, 99], [10, 100, 120, 150], [5, 5, 20, 18]], np.float32)
    iar = np.concatenate([np.zeros((len(rois), 1), np.float32), rois], 1)          # (idx, y1, x1, y2, x2)
    levels = np.clip(ob.map_rois_to_fpn_levels(rois), 0, 4).astype(np.int32)
    assert len(set(levels.tolist())) >= 2
    xs = [torch.from_numpy(f).to(DEV).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) for f in feats]
    core.TRAIN = True
    locs, scores, mask = h(xs, torch.from_numpy(iar).to(DEV), torch.from_numpy(levels).to(DEV), scales, train=True)

    # ---- Chainer semantics on the same arrays (NCHW, float64)
    D = torch.float64
    t = lambda k: torch.from_numpy(arrays[k]).to(D)

    def pool(size):
        out = []
        for l, r in zip(levels, iar):
            xy5 = r[[0, 2, 1, 4, 3]][None]
            out.append(ora.roi_align_fwd(feats[l], xy5, size, size, scales[l], 2))
        return torch.from_numpy(np.concatenate(out, 0)).to(D)

    pb = pool(h.roi_size_box)
    g = F.relu(F.conv2d(pb, t('head/conv1/W'), t('head/conv1/b'), padding=1))
    g = F.relu(F.linear(g.reshape(g.shape[0], -1), t('head/fc1/W'), t('head/fc1/b')))          # (C, H, W)-major flattening
    g = F.relu(F.linear(g, t('head/fc2/W'), t('head/fc2/b')))
    want_locs = F.linear(g, t('head/cls_loc/W'), t('head/cls_loc/b'))
    want_scores = F.linear(g, t('head/score/W'), t('head/score/b'))
    pm = pool(h.roi_size_mask)
    for i in range(1, 5):
        pm = F.relu(F.conv2d(pm, t('head/mask%d/W' % i), t('head/mask%d/b' % i), padding=1))
    up = F.conv_transpose2d(pm, t('head/deconv1/W'), t('head/deconv1/b'), stride=2)
    want_mask = F.conv2d(up, t('head/conv2/W'), t('head/conv2/b'))

    def rel(got, want):
        return float((got.detach().double().cpu() - want).abs().max()) / float(want.abs().max())

    assert locs.shape == want_locs.shape and scores.shape == want_scores.shape and mask.shape == want_mask.shape
    assert rel(locs, want_locs) <= 2e-5, rel(locs, want_locs)
    assert rel(scores, want_scores) <= 2e-5, rel(scores, want_scores)
    assert rel(mask, want_mask) <= 2e-5, rel(mask, want_mask)
