"""BASELINE.json configs[0] at FULL size: "FPN MaskRCNN forward on one 800x800 COCO image" - the real ResNet-50-FPN
(80 classes), not the reduced test network - through the reference's plumbing entry points

    MaskRCNN.prepare  (maskrcnn.py:261-276)   vs oracle/predict.py cv2_resize_linear_f32        bit-exact
    MaskRCNN.__call__ (maskrcnn.py:135-155)   vs oracle/model.py (float64)                      <= 1e-3 of tensor scale
    MaskRCNN.predict  (maskrcnn.py:157-259)   vs oracle/model.py + oracle/predict.py            index outputs bit-exact

Index work (proposal selection, per-class suppression) is checked by feeding the DEVICE's floating-point inputs of that
stage to the oracle, as everywhere else in this suite: box decoding differs by <= 2 ulp of exp() between NumPy and the
device, which could flip a tie in a keep list without being an error."""
import os

import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import predict as op
from oracle import proposal as opr
from oracle import boxes as ob

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.nn import core  # noqa: E402

DEV = 'cuda:0'
D = torch.float64
_cache = {}


def _model():
    if 'm' not in _cache:
        _cache['m'] = MaskRCNN(n_fg_class=80, device=DEV, seed=3)
    return _cache['m']


def _rel(got, want):
    want = want.detach().double()
    return float((got.detach().double().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def _oracle(m, bn_buffers=None):
    ps = m.ps
    params = {n: ps.p(n).detach().cpu().to(D) for n in ps.names()}
    return om.OracleStep(params, tuple(len(s) for s in m.extractor.stages), m.head.n_class, m.head.LOC0, bn_buffers=bn_buffers)


def test_prepare_matches_cv2_float_resize():
    """480x640 -> 600x800 (short side to min_size) and 500x1400 -> 357x1000 (long side capped at max_size)."""
    m = _model()
    rs = np.random.RandomState(0)
    for H, W in ((480, 640), (500, 1400), (800, 800)):
        img = (rs.rand(3, H, W) * 255).astype(np.float32)
        got = m.prepare(torch.from_numpy(img).to(DEV)).cpu().numpy()
        scale = 600 / min(H, W)
        if scale * max(H, W) > 1000:
            scale = 1000 / max(H, W)
        oh, ow = int(H * scale), int(W * scale)
        assert got.shape == (3, oh, ow)
        want = np.stack([op.cv2_resize_linear_f32(img[c], (ow, oh)) for c in range(3)]).astype(np.float32) / 255
        np.testing.assert_array_equal(got, want.astype(np.float32))


def test_call_800x800_train_mode_matches_oracle():
    """train.py's forward (chainer.config.train True: batch-statistics BN, 12000 -> 2000 proposals, box AND mask head on
    every proposal): A = 159,882 anchors."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    m = _model()
    m.train, core.TRAIN = True, True
    rs = np.random.RandomState(1)
    x = rs.rand(1, 3, 800, 800).astype(np.float32)
    xd = torch.from_numpy(x).to(DEV)
    roi_cls_locs, roi_scores, rois, roi_indices, mask = m(xd, 1.0)
    R = rois.shape[0]
    assert 0 < R <= 2000 and roi_cls_locs.shape == (R, 4) and roi_scores.shape == (R, 81) and mask.shape == (R, 80, 28, 28)
    assert int(roi_indices.abs().max()) == 0
    # the same call's intermediate tensors (deterministic kernels: a second evaluation is bitwise the same)
    feats = m.extractor(m.to_nhwc4(xd))
    assert [tuple(f.shape[1:3]) for f in feats] == [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    rpn_locs, rpn_scores, rois2, idx2, anchor, levels = m.rpn(feats, (800, 800), 1.0)
    assert anchor.shape[0] == 159882 and torch.equal(rois2, rois)
    with torch.no_grad():
        o = _oracle(m)
        img4 = torch.cat([torch.from_numpy(x).permute(0, 2, 3, 1), torch.zeros((1, 800, 800, 1))], -1).to(D)
        wf = o.extractor(img4)
        for l, (f, w) in enumerate(zip(feats, wf)):
            assert _rel(f, w) <= 1e-3, ('p%d' % (l + 2), _rel(f, w))
        wl, wsc = o.rpn(wf)
        assert _rel(rpn_locs, wl) <= 1e-3 and _rel(rpn_scores, wsc) <= 1e-3
        # proposals: the oracle's ProposalCreator on the DEVICE's head outputs (index work => exact)
        lv = ob.map_rois_to_fpn_levels(rois.cpu().numpy())
        np.testing.assert_array_equal(lv, levels.cpu().numpy())
        want_rois = opr.ProposalCreator()(rpn_locs[0].cpu().numpy(), rpn_scores[0, :, 1].cpu().numpy(), anchor.cpu().numpy(),
                                          (800, 800), 1.0, train=True)
        assert abs(want_rois.shape[0] - R) <= 3         # exp() differs by <= 2 ulp: a box on the IoU threshold may flip
        k = min(R, len(want_rois), 50)                  # the head of the list is stable under ulp-level differences
        np.testing.assert_allclose(rois.cpu().numpy()[:k], want_rois[:k], rtol=1e-5, atol=1e-3)
        # heads on the device's RoIs: box head on all of them, mask head on the first 48 (per-RoI independent)
        xy5 = torch.cat((roi_indices.to(torch.float32)[:, None], rois[:, [1, 0, 3, 2]]), 1).cpu().numpy()
        lvi = np.clip(lv, 0, 4).astype(np.int32)
        box = o.head_box(wf, xy5, lvi)
        assert _rel(roi_scores, box[:, :81]) <= 1e-3 and _rel(roi_cls_locs, box[:, m.head.LOC0:m.head.LOC0 + 4]) <= 1e-3
        k = min(48, R)
        wm = o.head_mask(wf, xy5[:k], lvi[:k])[..., :80].permute(0, 3, 1, 2)
        assert _rel(mask[:k], wm) <= 1e-3


def test_predict_800x800_matches_oracle():
    """MaskRCNN.predict on one 600x600 uint8-valued image (prepare resizes it to 800x800): inference-mode BN (running
    statistics), 6000 -> 300 proposals, decode, per-class suppression, mask head on the detections, paste."""
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    m = _model()
    m.use_preset('evaluate')
    keep = (m.score_thresh, m.min_size, m.max_size)
    m.score_thresh = 0.0135            # random weights: class probabilities sit around 1/81 = 0.0123
    m.min_size, m.max_size = 800, 1333  # the network input of configs[0] is 800x800: prepare scales 600 -> 800
    try:
        rs = np.random.RandomState(2)
        img = np.floor(rs.rand(3, 600, 600) * 256).astype(np.float32)
        masks, labels, scores = m.predict([torch.from_numpy(img)])
        bbox = m.last_bboxes[0].cpu().numpy()
        label, score, mk = labels[0].cpu().numpy(), scores[0].cpu().numpy(), masks[0].cpu().numpy()
        Dn = len(label)
        assert Dn > 0 and mk.shape == (Dn, 600, 600)
        with torch.no_grad():
            x = np.stack([op.cv2_resize_linear_f32(img[c], (800, 800)) for c in range(3)]).astype(np.float32) / 255
            bufs = {k: v.cpu() for k, v in m.ps.buffers.items()}
            o = _oracle(m, bn_buffers=bufs)
            img4 = torch.cat([torch.from_numpy(x[None]).permute(0, 2, 3, 1), torch.zeros((1, 800, 800, 1))], -1).to(D)
            wf = o.extractor(img4)
            # device side of the same forward (predict() keeps the pyramid in head.x and the box output in last_box_out)
            for l, (f, w) in enumerate(zip(m.head.x, wf)):
                assert _rel(f, w) <= 1e-3, ('p%d' % (l + 2), _rel(f, w))
            scale = 800 / 600
            rois, box_out = m.last_rois, m.head.last_box_out
            R = rois.shape[0]
            assert 0 < R <= 300
            lv = np.clip(ob.map_rois_to_fpn_levels(rois.cpu().numpy()), 0, 4).astype(np.int32)
            xy5 = np.concatenate([np.zeros((R, 1), np.float32), rois.cpu().numpy()[:, [1, 0, 3, 2]]], 1)
            wbox = o.head_box(wf, xy5, lv)
            nc, l0 = 81, m.head.LOC0
            assert _rel(box_out[:, :nc], wbox[:, :nc]) <= 1e-3 and _rel(box_out[:, l0:l0 + 4], wbox[:, l0:l0 + 4]) <= 1e-3
            # decode + suppress on the device's box-head output
            bo = box_out.cpu().numpy()
            cls_bbox, prob = op.decode(rois.cpu().numpy(), bo[:, l0:l0 + 4], bo[:, :nc], scale, (600, 600), nc)
            dev_bbox, dev_prob = m.last_decoded
            np.testing.assert_allclose(dev_bbox.cpu().numpy(), cls_bbox, rtol=1e-5, atol=1e-3)
            np.testing.assert_allclose(dev_prob.cpu().numpy(), prob, rtol=1e-5, atol=1e-7)
            idx, lab = op.suppress(dev_bbox.cpu().numpy(), dev_prob.cpu().numpy(), nc, m.nms_thresh, m.score_thresh, predict_mask=True)
            np.testing.assert_array_equal(label, lab)
            np.testing.assert_array_equal(bbox, dev_bbox.cpu().numpy()[idx])
            np.testing.assert_array_equal(score, dev_prob.cpu().numpy()[idx, lab + 1])
            # masks of (up to) the first 24 detections: oracle mask head on the detection boxes, oracle paste
            k = min(24, Dn)
            dxy5 = np.concatenate([np.zeros((k, 1), np.float32), (bbox[:k] * np.float32(scale))[:, [1, 0, 3, 2]]], 1).astype(np.float32)
            wm = o.head_mask(wf, dxy5, lv[idx[:k]])[..., :80].permute(0, 3, 1, 2).numpy()
            want = op.paste_masks(wm, lab[:k], bbox[:k], (600, 600))
            diff = (want != mk[:k]).sum() / max(int(want.sum()), 1)
            assert diff <= 2e-3, diff           # pixels whose sigmoid sits within float32 rounding of the 127/255 threshold
    finally:
        m.score_thresh, m.min_size, m.max_size = keep
        m.use_preset('visualize')
