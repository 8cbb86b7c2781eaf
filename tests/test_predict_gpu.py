"""GPU parity tests of the inference path: BN in inference mode, box decode + softmax,
per-class suppression and mask paste against oracle/predict.py; MaskRCNN.predict end to end on a reduced network."""
import numpy as np
import pytest
import torch

from oracle import predict as op
from oracle import boxes as ob

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402
from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402

DEV = 'cuda:0'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_bn_inference_mode():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 5, 7, 64), generator=g, dtype=torch.float64)
    gamma, beta, mean = (torch.randn((64,), generator=g, dtype=torch.float64) for _ in range(3))
    var = torch.rand((64,), generator=g, dtype=torch.float64) + 0.1
    res = torch.randn(x.shape, generator=g, dtype=torch.float64)
    want = (gamma * (x - mean) / torch.sqrt(var + 2e-5) + beta + res).clamp_min(0)
    got = ops.bn_infer_fwd(*(t.float().to(DEV) for t in (x, gamma, beta, mean, var)), residual=res.float().to(DEV), relu=True)
    assert (got.double().cpu() - want).abs().max() < 1e-5


def _case(seed, R=300, n_class=81, ld=96, loc0=88):
    rs = np.random.RandomState(seed)
    c = rs.uniform(50, 550, (R, 2)); hw = np.exp(rs.uniform(np.log(20), np.log(300), (R, 2)))
    rois = np.concatenate([c - hw / 2, c + hw / 2], 1).astype(np.float32)
    box = np.zeros((R, ld), np.float32)
    box[:, :n_class] = rs.standard_normal((R, n_class)) * 3
    box[:, loc0:loc0 + 4] = rs.standard_normal((R, 4)) * 0.5
    return rois, box


def test_decode_softmax_and_suppress_match_oracle():
    rois, box = _case(1)
    scale, size = 1.25, (480, 500)
    cls_bbox, prob = ops.detect_decode(_t(rois), _t(box), 81, 88, scale, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), size)
    wb, wp = op.decode(rois, box[:, 88:92], box[:, :81], scale, size, 81)
    np.testing.assert_allclose(cls_bbox.cpu().numpy(), wb, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(prob.cpu().numpy(), wp, rtol=1e-5, atol=1e-7)
    # suppression is index work: feed the DEVICE boxes / probabilities to the oracle => bit-exact selections
    for thresh in (0.05, 0.3):
        keep_idx, keep_cnt = ops.class_nms(cls_bbox, prob, 1, 80, thresh, 0.3)
        idx, lab = op.suppress(cls_bbox.cpu().numpy(), prob.cpu().numpy(), 81, 0.3, thresh, predict_mask=True)
        cnt = keep_cnt.cpu().numpy()
        got_idx = np.concatenate([keep_idx[l, :cnt[l]].cpu().numpy() for l in range(1, 80)])
        got_lab = np.concatenate([np.full(cnt[l], l - 1) for l in range(1, 80)])
        np.testing.assert_array_equal(got_idx, idx)
        np.testing.assert_array_equal(got_lab, lab)
        assert cnt[80] == 0 and cnt[0] == 0                 # background and the skipped last class


def test_mask_paste_matches_cv2_restatement():
    rs = np.random.RandomState(2)
    D, S, Cm, n_fg = 9, 28, 96, 80
    logits = (rs.standard_normal((D, S, S, Cm)) * 2).astype(np.float32)
    label = rs.randint(0, n_fg - 1, D).astype(np.int32)
    H, W = 97, 131
    y0 = rs.uniform(0, H - 30, D); x0 = rs.uniform(0, W - 30, D)
    bbox = np.stack([y0, x0, np.minimum(y0 + rs.uniform(5, 80, D), H), np.minimum(x0 + rs.uniform(5, 90, D), W)], 1).astype(np.float32)
    bbox[0] = [3.2, 4.7, 3.9, 60.0]          # zero-height box
    got = ops.mask_paste(_t(logits), _t(label), _t(bbox), (H, W)).cpu().numpy().astype(bool)
    want = op.paste_masks(logits[..., :n_fg].transpose(0, 3, 1, 2), label, bbox, (H, W))
    assert got.shape == want.shape
    assert (got != want).mean() < 1e-4        # exp() ulp differences can flip a pixel sitting exactly on the threshold
    assert got[1:].any() and not got[0].any()


def test_predict_end_to_end_on_reduced_network():
    m = MaskRCNN(n_fg_class=80, device=DEV, seed=5, _test_shrink=dict(stages=(1, 1, 1, 1), width_div=2), min_size=160, max_size=260)
    m.use_preset('evaluate')
    m.score_thresh = 0.0125                         # random weights: ~uniform class probabilities (1/81 = 0.0123)
    rs = np.random.RandomState(0)
    imgs = [torch.from_numpy((rs.rand(3, 120, 150) * 255).astype(np.float32)), torch.from_numpy((rs.rand(3, 100, 100) * 255).astype(np.float32))]
    masks, labels, scores = m.predict(imgs)
    assert len(masks) == len(labels) == len(scores) == 2
    for img, mk, lb, sc, bb in zip(imgs, masks, labels, scores, m.last_bboxes):
        D = lb.shape[0]
        assert mk.shape == (D,) + tuple(img.shape[1:]) and mk.dtype == torch.bool
        assert sc.shape == (D,) and bb.shape == (D, 4)
        if D:
            assert int(lb.min()) >= 0 and int(lb.max()) <= 78       # the last class is skipped (maskrcnn.py:288-291)
            assert float(sc.min()) > m.score_thresh
            b = bb.cpu().numpy()
            assert (b[:, 0] >= 0).all() and (b[:, 2] <= img.shape[1]).all() and (b[:, 3] <= img.shape[2]).all()
            # masks live inside their boxes
            ys = mk.any(dim=2).cpu().numpy(); xs = mk.any(dim=1).cpu().numpy()
            for i in range(min(D, 20)):
                if ys[i].any():
                    assert np.nonzero(ys[i])[0].min() >= int(b[i, 0]) and np.nonzero(ys[i])[0].max() < int(b[i, 0]) + max(int(b[i, 2] - b[i, 0]), 1)
    assert m.train is True                                           # training mode restored
    assert sum(l.shape[0] for l in labels) > 0


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_suppress_equals_reference_golden(ci):
    """tests/golden/suppress_reference.npz: MaskRCNN._suppress of the REFERENCE (maskrcnn.py:278-312) executed in the build
    container.  The device's per-class threshold + NMS (mrcnn_class_nms_f32) and the host concatenation reproduce its
    boxes, labels, scores and levels exactly."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'suppress_reference.npz'))
    g = lambda k: d['c%d_in_%s' % (ci, k)]
    n_class = int(g('n_class'))
    m = MaskRCNN.__new__(MaskRCNN)              # _suppress only reads these attributes
    import types
    m.head = types.SimpleNamespace(n_class=n_class)
    m.predict_mask, m.score_thresh, m.nms_thresh = bool(g('predict_mask')), float(g('score_thresh')), float(g('nms_thresh'))
    bbox, label, score, level = m._suppress(_t(g('box')), _t(g('prob')), _t(g('level')))
    np.testing.assert_array_equal(bbox.cpu().numpy(), d['c%d_out_bbox' % ci])
    np.testing.assert_array_equal(label.cpu().numpy(), d['c%d_out_label' % ci])
    np.testing.assert_array_equal(score.cpu().numpy(), d['c%d_out_score' % ci])
    np.testing.assert_array_equal(level.cpu().numpy(), d['c%d_out_level' % ci])


@pytest.mark.parametrize('ci', [0, 1, 2, 3, 4])
def test_prepare_equals_reference_golden(ci):
    """tests/golden/prepare_reference.npz: MaskRCNN.prepare of the reference (maskrcnn.py:261-276) executed in the build
    container (resize = this repo's cv2 restatement): size rule, interpolation and /255 on the device, bit for bit."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'prepare_reference.npz'))
    m = MaskRCNN.__new__(MaskRCNN)
    m.min_size, m.max_size = (int(v) for v in d['c%d_in_min_max' % ci])
    m.device = torch.device(DEV)
    got = m.prepare(torch.from_numpy(d['c%d_in_img' % ci]).to(DEV))
    np.testing.assert_array_equal(got.cpu().numpy(), d['c%d_out' % ci])
