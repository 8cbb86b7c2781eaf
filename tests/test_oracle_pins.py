"""CPU tests: the oracle against the golden vectors produced by the reference's own code
(tests/golden/make_reference_vectors.py) and against its own literal scalar forms."""
import os

import numpy as np
import pytest

from oracle import boxes, roi_align as ora, targets, proposal, losses


def test_levels_match_reference_function(golden_dir):
    d = np.load(os.path.join(golden_dir, 'levels_reference.npz'))
    got = boxes.map_rois_to_fpn_levels(d['rois'])
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, d['levels'])
    # the values quoted in SURVEY.md section 8c
    special = np.array([[0, 0, 224, 224], [0, 0, 112, 112], [0, 0, 56, 56], [0, 0, 28, 28],
                        [0, 0, 14, 14], [0, 0, 0, 0], [0, 0, 1000, 1000]], np.float32)
    np.testing.assert_array_equal(boxes.map_rois_to_fpn_levels(special), [4, 3, 2, 1, 0, 0, 4])


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_proposal_target_creator_control_flow_matches_reference(golden_dir, ci):
    d = np.load(os.path.join(golden_dir, 'ptc_reference.npz'))
    shape = tuple(d['c%d_in_mask_shape' % ci])
    mask = np.unpackbits(d['c%d_in_mask' % ci], axis=-1)[..., :shape[-1]].reshape(shape)
    rng = np.random.RandomState(int(d['c%d_in_np_seed' % ci]))
    out = targets.ProposalTargetCreator([32, 64, 128, 256, 512])(
        d['c%d_in_roi' % ci], d['c%d_in_bbox' % ci], d['c%d_in_label' % ci], mask,
        d['c%d_in_levels' % ci], mask_size=28, binary_mask=True, rng=rng)
    for k, v in zip(('sample_roi', 'sample_levels', 'gt_roi_loc', 'gt_roi_label', 'gt_roi_mask'), out):
        np.testing.assert_array_equal(np.asarray(v), d['c%d_out_%s' % (ci, k)], err_msg=k)


@pytest.mark.parametrize('ci', [0, 1])
def test_proposal_target_creator_keypoints_matches_reference(golden_dir, ci):
    d = np.load(os.path.join(golden_dir, 'ptc_keypoint_reference.npz'))
    kp = d['c%d_in_kp' % ci].copy()
    rng = np.random.RandomState(int(d['c%d_in_np_seed' % ci]))
    out = targets.ProposalTargetCreator([32, 64, 128, 256, 512])(
        d['c%d_in_roi' % ci], d['c%d_in_bbox' % ci], d['c%d_in_label' % ci], kp,
        d['c%d_in_levels' % ci], mask_size=56, binary_mask=False, rng=rng)
    for k, v in zip(('sample_roi', 'sample_levels', 'gt_roi_loc', 'gt_roi_label', 'gt_roi_mask'), out):
        np.testing.assert_array_equal(np.asarray(v), d['c%d_out_%s' % (ci, k)], err_msg=k)
    np.testing.assert_array_equal(kp, d['c%d_out_kp_after' % ci])   # in-place mutation quirk


def _rand_rois(rs, R, N, H, W, scale):
    h = np.exp(rs.uniform(np.log(2), np.log(H / scale * 1.2), R))
    w = np.exp(rs.uniform(np.log(2), np.log(W / scale * 1.2), R))
    cy = rs.uniform(-0.1 * H / scale, 1.1 * H / scale, R)
    cx = rs.uniform(-0.1 * W / scale, 1.1 * W / scale, R)
    idx = rs.randint(0, N, R)
    return np.stack([idx, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)


@pytest.mark.parametrize('sr', [1, 2, 0])
def test_roi_align_vectorised_equals_scalar_formula(sr):
    rs = np.random.RandomState(3)
    x = rs.standard_normal((2, 3, 9, 11)).astype(np.float32)
    rois = _rand_rois(rs, 7, 2, 9, 11, 0.5)
    rois[0] = [0, -30, -30, -20, -20]          # fully outside
    rois[1] = [1, 4, 4, 4, 4]                  # zero area
    rois[2] = [0, 0, 0, 22, 18]                # whole map
    a = ora.roi_align_fwd(x, rois, 3, 2, 0.5, sr)
    b = ora.roi_align_fwd_scalar(x, rois, 3, 2, 0.5, sr)
    np.testing.assert_array_equal(a, b)
    assert np.all(a[0] == 0)


@pytest.mark.parametrize('sr', [2, 0])
def test_roi_align_backward_is_adjoint(sr):
    rs = np.random.RandomState(4)
    x = rs.standard_normal((2, 4, 12, 10)).astype(np.float32)
    rois = _rand_rois(rs, 9, 2, 12, 10, 0.25)
    y = ora.roi_align_fwd(x, rois, 4, 4, 0.25, sr)
    gy = rs.standard_normal(y.shape).astype(np.float32)
    gx = ora.roi_align_bwd(gy, rois, x.shape, 0.25, sr)
    lhs = np.sum(y.astype(np.float64) * gy)
    rhs = np.sum(gx.astype(np.float64) * x)
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


def test_nms_crafted_ties_and_threshold_equal():
    # b1 has IoU exactly 0.5 with b0 -> suppressed at thresh 0.5 (>=), kept at 0.5000001
    b = np.array([[0, 0, 10, 10], [0, 5, 10, 20], [0, 0, 10, 10], [50, 50, 60, 60]], np.float32)
    iou01 = (10 * 5) / (100 + 150 - 50.)
    assert iou01 == 0.25
    b[1] = [0, 0, 10, 20]                       # inter 100, union 200 -> 0.5
    np.testing.assert_array_equal(boxes.nms(b, 0.5), [0, 3])
    np.testing.assert_array_equal(boxes.nms(b, np.float32(0.50001)), [0, 1, 3])


def test_argsort_pin_is_score_desc_index_desc():
    s = np.array([1, 3, 3, 0, 3, 1], np.float32)
    np.testing.assert_array_equal(boxes.argsort_desc_pinned(s), [4, 2, 1, 5, 0, 3])


def test_cv2_resize_known_answers():
    # constant image stays constant; identity size copies; a half/half step image rounds at .5
    a = np.ones((9, 13), np.uint8)
    assert np.all(targets.cv2_resize_linear_u8(a, (28, 28)) == 1)
    a = np.zeros((4, 4), np.uint8); a[:, 2:] = 1
    out = targets.cv2_resize_linear_u8(a, (8, 8))
    assert out.shape == (8, 8) and set(np.unique(out)) <= {0, 1}
    np.testing.assert_array_equal(out[0], [0, 0, 0, 0, 1, 1, 1, 1])
    with pytest.raises(ValueError):
        targets.cv2_resize_linear_u8(np.zeros((0, 3), np.uint8), (28, 28))


def test_anchor_enumeration_counts():
    shapes = [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    a = boxes.fpn_anchors(shapes)
    assert a.shape == (159882, 4)
    base = boxes.generate_anchor_base(anchor_scales=[2.], ratios=[0.5, 1, 2])
    np.testing.assert_allclose(base[1], [-8, -8, 24, 24])


def test_losses_gradients_numeric():
    rs = np.random.RandomState(5)
    x = rs.standard_normal((6, 5)).astype(np.float32)
    t = np.array([0, 4, -1, 2, 2, 1])
    l, g = losses.softmax_cross_entropy(x, t)
    eps = 1e-2
    xp = x.copy(); xp[1, 3] += eps
    xm = x.copy(); xm[1, 3] -= eps
    num = (losses.softmax_cross_entropy(xp, t)[0] - losses.softmax_cross_entropy(xm, t)[0]) / (2 * eps)
    assert abs(num - g[1, 3]) < 1e-3
    assert np.all(g[2] == 0)
    xs = rs.standard_normal((3, 4, 4)).astype(np.float32)
    ts = rs.randint(0, 2, (3, 4, 4))
    l, g = losses.sigmoid_cross_entropy(xs, ts)
    xp = xs.copy(); xp[0, 1, 1] += eps
    xm = xs.copy(); xm[0, 1, 1] -= eps
    num = (losses.sigmoid_cross_entropy(xp, ts)[0] - losses.sigmoid_cross_entropy(xm, ts)[0]) / (2 * eps)
    assert abs(num - g[0, 1, 1]) < 1e-3


def test_proposal_creator_shapes_and_order():
    rs = np.random.RandomState(6)
    shapes = [(16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    anchors = boxes.fpn_anchors(shapes)
    loc = (rs.standard_normal((anchors.shape[0], 4)) * 0.2).astype(np.float32)
    score = rs.standard_normal(anchors.shape[0]).astype(np.float32)
    pc = proposal.ProposalCreator(n_train_pre_nms=300, n_train_post_nms=50)
    roi, dbg = pc(loc, score, anchors, (64, 64), return_debug=True)
    assert roi.shape[0] <= 50 and roi.shape[1] == 4
    s = score[dbg['anchor_index']]
    assert np.all(np.diff(s) <= 0)
    assert np.all(roi[:, 2] - roi[:, 0] >= 16) and np.all(roi[:, 3] - roi[:, 1] >= 16)


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_proposal_creator_control_flow_matches_reference(golden_dir, ci):
    """pc_reference.npz = outputs of the REFERENCE's in-tree ProposalCreator (utils/proposal_creator.py:108-169) executed in
    the build container (tests/golden/make_reference_vectors.py): clip, min_size * scale filter, descending sort, top n_pre,
    NMS, top n_post - train and test presets.  The oracle's ProposalCreator (what every device test is compared with) must
    give the same RoIs bit for bit."""
    from oracle import proposal as opr
    d = np.load(os.path.join(golden_dir, 'pc_reference.npz'))
    g = lambda k: d['c%d_in_%s' % (ci, k)]
    n_pre, n_post, train = int(g('n_pre')), int(g('n_post')), bool(g('train'))
    pc = opr.ProposalCreator(n_train_pre_nms=n_pre, n_train_post_nms=n_post, n_test_pre_nms=n_pre, n_test_post_nms=n_post,
                             min_size=int(g('min_size')))
    got = pc(g('loc'), g('score'), g('anchor'), tuple(int(v) for v in g('img')), scale=float(g('scale')), train=train)
    np.testing.assert_array_equal(got, d['c%d_out_roi' % ci])


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_suppress_control_flow_matches_reference(golden_dir, ci):
    """suppress_reference.npz = MaskRCNN._suppress of the reference (maskrcnn.py:278-312) executed in the build container with
    this oracle's NMS as ChainerCV's: per-class loop, score threshold, the skipped last class when masks are predicted, label
    offsets, concatenation order."""
    from oracle import predict as op
    d = np.load(os.path.join(golden_dir, 'suppress_reference.npz'))
    g = lambda k: d['c%d_in_%s' % (ci, k)]
    box, prob, level = g('box'), g('prob'), g('level')
    idx, lab = op.suppress(box, prob, int(g('n_class')), float(g('nms_thresh')), float(g('score_thresh')), predict_mask=bool(g('predict_mask')))
    np.testing.assert_array_equal(box[idx], d['c%d_out_bbox' % ci])
    np.testing.assert_array_equal(lab, d['c%d_out_label' % ci])
    np.testing.assert_array_equal(prob[idx, lab + 1], d['c%d_out_score' % ci])
    np.testing.assert_array_equal(level[idx], d['c%d_out_level' % ci])


@pytest.mark.parametrize('ci', [0, 1, 2, 3, 4])
def test_prepare_size_rule_matches_reference(golden_dir, ci):
    from oracle import predict as op
    d = np.load(os.path.join(golden_dir, 'prepare_reference.npz'))
    mn, mx = (int(v) for v in d['c%d_in_min_max' % ci])
    np.testing.assert_array_equal(op.prepare(d['c%d_in_img' % ci], mn, mx), d['c%d_out' % ci])
