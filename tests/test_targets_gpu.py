"""GPU parity tests: device target creators (targets.hip) against the NumPy oracle
(oracle/targets.py).  Index outputs, labels and mask targets bit-exact; box regression targets
involve log() and are compared to 1e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle import targets as ot

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402

DEV = 'cuda:0'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
i32 = torch.int32


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype else t).to(DEV)


def _scene(seed, N, n_roi, G, H, W, roi_cap, gt_cap):
    rs = np.random.RandomState(seed)
    gt = np.zeros((N, gt_cap, 4), np.float32)
    lab = np.zeros((N, gt_cap), np.int32)
    rois = np.zeros((N, roi_cap, 4), np.float32)
    lev = np.zeros((N, roi_cap), np.float32)
    n_gt = np.zeros(N, np.int32)
    n_r = np.zeros(N, np.int32)
    for i in range(N):
        g = G - i
        c = rs.uniform(0.2, 0.8, (g, 2)) * [H, W]
        hw = np.exp(rs.uniform(np.log(24), np.log(min(H, W) / 2), (g, 2)))
        b = np.concatenate([np.maximum(c - hw / 2, 0), np.minimum(c + hw / 2, [H, W])], 1).astype(np.float32)
        gt[i, :g] = b
        lab[i, :g] = rs.randint(0, 80, g)
        n_gt[i] = g
        nr = n_roi - 7 * i
        # half the proposals are jittered gt boxes (so positives exist), half random
        j = b[rs.randint(0, g, nr // 2)] + rs.uniform(-12, 12, (nr // 2, 4)).astype(np.float32)
        c2 = rs.uniform(0, 1, (nr - nr // 2, 2)) * [H, W]
        hw2 = np.exp(rs.uniform(np.log(16), np.log(min(H, W) / 2), (nr - nr // 2, 2)))
        r = np.concatenate([j, np.concatenate([c2 - hw2 / 2, c2 + hw2 / 2], 1)], 0)
        r[:, 0::2] = np.clip(r[:, 0::2], 0, H)
        r[:, 1::2] = np.clip(r[:, 1::2], 0, W)
        r = r.astype(np.float32)
        ok = ((r[:, 2] - r[:, 0]) >= 4) & ((r[:, 3] - r[:, 1]) >= 4)
        r = r[ok]
        rois[i, :len(r)] = r
        lev[i, :len(r)] = ob.map_rois_to_fpn_levels(r)
        n_r[i] = len(r)
    keys = rs.randint(0, 2 ** 32, (N, roi_cap + gt_cap), dtype=np.uint64).astype(np.uint32)
    keys[:, 3:9] = keys[:, 3:4]        # equal keys: ties resolved by index
    return rois, lev, n_r, gt, lab, n_gt, keys


@pytest.mark.parametrize('seed,n_roi,G', [(0, 2000, 8), (1, 300, 3), (2, 40, 2)])
def test_proposal_target_matches_key_driven_oracle(seed, n_roi, G):
    N, H, W, roi_cap, gt_cap = 2, 512, 640, 2000, 16
    rois, lev, n_r, gt, lab, n_gt, keys = _scene(seed, N, n_roi, G, H, W, roi_cap, gt_cap)
    o = ops.proposal_target(_t(rois.reshape(-1, 4)), _t(lev.reshape(-1)), _t(n_r), _t(gt), _t(lab), _t(n_gt),
                            _t(keys.view(np.int32)))
    o = {k: v.cpu().numpy() for k, v in o.items()}
    for i in range(N):
        r, g = rois[i, :n_r[i]], gt[i, :n_gt[i]]
        kk = np.concatenate([keys[i, :n_r[i]], keys[i, roi_cap:roi_cap + n_gt[i]]])
        keep, n_pos, assign, _ = ot.proposal_targets_from_keys(r, g, lab[i, :n_gt[i]], kk)
        S = len(keep)
        rows = slice(i * 256, i * 256 + S)
        assert o['n_pos'][i] == n_pos and o['n_sampled'][i] == S
        np.testing.assert_array_equal(o['sample_src'][rows], keep)
        allb = np.concatenate([r, g], 0)
        np.testing.assert_array_equal(o['sample_roi'][rows], allb[keep])
        want_label = lab[i][assign[keep]] + 1
        want_label[n_pos:] = 0
        np.testing.assert_array_equal(o['gt_roi_label'][rows], want_label)
        np.testing.assert_array_equal(o['gt_assign'][rows], assign[keep])
        want_lev = np.concatenate([lev[i, :n_r[i]], ob.map_rois_to_fpn_levels(g)])[keep]
        np.testing.assert_array_equal(o['sample_levels'][rows], want_lev.astype(np.int32))
        want_loc = (ob.bbox2loc(allb[keep], g[assign[keep]]) - np.zeros(4, np.float32)) / np.array([.1, .1, .2, .2], np.float32)
        np.testing.assert_allclose(o['gt_roi_loc'][rows], want_loc, rtol=1e-5, atol=1e-5)
        xy5 = o['rois_xy5'][rows]
        np.testing.assert_array_equal(xy5[:, 0], i)
        np.testing.assert_array_equal(xy5[:, 1:], allb[keep][:, [1, 0, 3, 2]])
        pad = slice(i * 256 + S, (i + 1) * 256)
        assert np.all(o['gt_roi_label'][pad] == -1)


def test_proposal_target_reference_golden_candidate_sets():
    """The golden vectors (produced by the REFERENCE class with np.random) pin the candidate sets and
    the output bookkeeping: with any keys the device must sample from the same pos/neg sets and the
    same counts; labels / loc targets of the rows both picked must agree."""
    d = np.load(os.path.join(GOLDEN, 'ptc_reference.npz'))
    for ci in range(4):
        roi, bbox, label = d['c%d_in_roi' % ci], d['c%d_in_bbox' % ci], d['c%d_in_label' % ci]
        R, G = len(roi), len(bbox)
        rs = np.random.RandomState(ci)
        keys = rs.randint(0, 2 ** 32, (1, R + G), dtype=np.uint64).astype(np.uint32)
        o = ops.proposal_target(_t(roi), _t(d['c%d_in_levels' % ci]), _t(np.array([R], np.int32)), _t(bbox[None]),
                                _t(label[None]), _t(np.array([G], np.int32)), _t(keys.view(np.int32)))
        o = {k: v.cpu().numpy() for k, v in o.items()}
        want_roi, want_label = d['c%d_out_sample_roi' % ci], d['c%d_out_gt_roi_label' % ci]
        S, n_pos = len(want_roi), int((want_label > 0).sum())
        assert o['n_sampled'][0] == S and o['n_pos'][0] == n_pos
        ref = {tuple(r): (l, tuple(np.round(loc, 4))) for r, l, loc in
               zip(want_roi, want_label, d['c%d_out_gt_roi_loc' % ci])}
        hits = 0
        for r, l, loc in zip(o['sample_roi'][:S], o['gt_roi_label'][:S], o['gt_roi_loc'][:S]):
            if tuple(r) in ref:
                hits += 1
                assert ref[tuple(r)][0] == l
                np.testing.assert_allclose(loc, ref[tuple(r)][1], atol=2e-4)
        # fewer than 64 foreground candidates: both samplers take them all; otherwise the random subsets overlap
        assert hits >= (n_pos if n_pos < 64 else 1)


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_proposal_target_reference_order_equals_reference_golden(ci):
    """Reference-order mode: the host draws np.random.choice exactly like utils/proposal_target_creator.py:63-78 (same
    NumPy seed as the golden run), the device consumes the draw order => ALL FIVE outputs equal the reference's own,
    row for row (sample_roi, sample_levels, gt_roi_label, gt_roi_mask bit-exact; gt_roi_loc involves log())"""
    from chainer_maskrcnn.utils.proposal_target_creator import ProposalTargetCreator
    d = np.load(os.path.join(GOLDEN, 'ptc_reference.npz'))
    shape = tuple(d['c%d_in_mask_shape' % ci])
    mask = np.unpackbits(d['c%d_in_mask' % ci], axis=-1)[..., :shape[-1]].reshape(shape)
    roi, bbox, label = d['c%d_in_roi' % ci], d['c%d_in_bbox' % ci], d['c%d_in_label' % ci]
    ptc = ProposalTargetCreator([32, 64, 128, 256, 512])
    rs = np.random.RandomState(int(d['c%d_in_np_seed' % ci]))
    out = ptc(_t(roi), _t(bbox), _t(label), _t(mask.astype(np.uint8)), _t(d['c%d_in_levels' % ci]), mask_size=28,
              binary_mask=True, random_state=rs)
    sample_roi, sample_levels, gt_roi_loc, gt_roi_label, gt_roi_mask = [o.cpu().numpy() for o in out]
    np.testing.assert_array_equal(sample_roi, d['c%d_out_sample_roi' % ci])
    np.testing.assert_array_equal(sample_levels, d['c%d_out_sample_levels' % ci])
    np.testing.assert_array_equal(gt_roi_label, d['c%d_out_gt_roi_label' % ci])
    np.testing.assert_array_equal(gt_roi_mask, d['c%d_out_gt_roi_mask' % ci])
    np.testing.assert_allclose(gt_roi_loc, d['c%d_out_gt_roi_loc' % ci], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('ci', [0, 1])
def test_keypoint_proposal_target_reference_order_with_inplace_quirk(ci):
    """Same for the keypoint variant, including the reference's in-place mutation of the gt keypoints (App. B-11):
    gt_roi_mask (flat 56x56 indices / -1) equals the reference's output row for row."""
    from chainer_maskrcnn.utils.proposal_target_creator import ProposalTargetCreator
    d = np.load(os.path.join(GOLDEN, 'ptc_keypoint_reference.npz'))
    ptc = ProposalTargetCreator([32, 64, 128, 256, 512])
    rs = np.random.RandomState(int(d['c%d_in_np_seed' % ci]))
    out = ptc(_t(d['c%d_in_roi' % ci]), _t(d['c%d_in_bbox' % ci]), _t(d['c%d_in_label' % ci]), _t(d['c%d_in_kp' % ci]),
              _t(d['c%d_in_levels' % ci]), mask_size=56, binary_mask=False, random_state=rs, inplace_kp_quirk=True)
    sample_roi, sample_levels, gt_roi_loc, gt_roi_label, gt_roi_mask = [o.cpu().numpy() for o in out]
    np.testing.assert_array_equal(sample_roi, d['c%d_out_sample_roi' % ci])
    np.testing.assert_array_equal(gt_roi_label, d['c%d_out_gt_roi_label' % ci])
    np.testing.assert_array_equal(gt_roi_mask, d['c%d_out_gt_roi_mask' % ci])


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_mask_targets_bit_exact_vs_reference_golden(ci):
    """Feed the reference's own sampled RoIs (golden) to the device crop+resize kernel: its outputs must
    equal the golden gt_roi_mask (produced by the reference loop with the oracle's cv2 restatement)."""
    d = np.load(os.path.join(GOLDEN, 'ptc_reference.npz'))
    shape = tuple(d['c%d_in_mask_shape' % ci])
    mask = np.unpackbits(d['c%d_in_mask' % ci], axis=-1)[..., :shape[-1]].reshape(shape)
    roi, bbox = d['c%d_in_roi' % ci], d['c%d_in_bbox' % ci]
    sample_roi = d['c%d_out_sample_roi' % ci]
    want = d['c%d_out_gt_roi_mask' % ci]
    n_pos = want.shape[0]
    iou = ob.bbox_iou(sample_roi[:n_pos], bbox)
    assign = iou.argmax(1).astype(np.int32)
    S = 256
    sr = np.zeros((S, 4), np.float32)
    sr[:len(sample_roi)] = sample_roi
    ga = np.full((S,), -1, np.int32)
    ga[:n_pos] = assign
    got = ops.mask_target(_t(mask[None].astype(np.uint8)), _t(sr), _t(ga), _t(np.array([n_pos], np.int32)), S, 64, 28)
    got = got.cpu().numpy()
    np.testing.assert_array_equal(got[:n_pos], want)
    assert np.all(got[n_pos:] == -1)


def test_mask_target_resize_random_shapes_vs_cv2_restatement():
    rs = np.random.RandomState(9)
    H, W, G = 120, 90, 6
    masks = (rs.uniform(0, 1, (1, G, H, W)) > 0.5).astype(np.uint8)
    S, n_pos = 64, 40
    y0 = rs.uniform(0, H - 2, n_pos); x0 = rs.uniform(0, W - 2, n_pos)
    y1 = np.minimum(y0 + np.exp(rs.uniform(np.log(1.5), np.log(H), n_pos)), H + 3.7)
    x1 = np.minimum(x0 + np.exp(rs.uniform(np.log(1.5), np.log(W), n_pos)), W + 2.2)
    sr = np.zeros((S, 4), np.float32)
    sr[:n_pos] = np.stack([y0, x0, y1, x1], 1)
    ga = np.full((S,), -1, np.int32)
    ga[:n_pos] = rs.randint(0, G, n_pos)
    for msz in (28, 14):
        got = ops.mask_target(_t(masks), _t(sr), _t(ga), _t(np.array([n_pos], np.int32)), S, S, msz).cpu().numpy()
        for j in range(n_pos):
            b = sr[j]
            A = masks[0, ga[j], max(int(b[0]), 0):min(int(b[2]), H), max(int(b[1]), 0):min(int(b[3]), W)]
            np.testing.assert_array_equal(got[j], ot.cv2_resize_linear_u8(A, (msz, msz)).astype(np.int32), err_msg=str(j))


def test_keypoint_targets_vs_oracle_without_inplace_mutation():
    d = np.load(os.path.join(GOLDEN, 'ptc_keypoint_reference.npz'))
    for ci in range(2):
        kp = d['c%d_in_kp' % ci]
        bbox = d['c%d_in_bbox' % ci]
        sample_roi = d['c%d_out_sample_roi' % ci]
        n_pos = d['c%d_out_gt_roi_mask' % ci].shape[0]
        assign = ob.bbox_iou(sample_roi[:n_pos], bbox).argmax(1).astype(np.int32)
        # oracle statement of the per-RoI transform on a COPY of the gt keypoints
        want = np.zeros((n_pos, kp.shape[1]), np.int32)
        for i in range(n_pos):
            y0, x0, y1, x1 = [int(v) for v in sample_roi[i]]
            k = kp[assign[i]].copy()
            k[:, :2] = (k[:, :2] - [y0, x0]) / [max(y1 - y0, 1), max(x1 - x0, 1)] * 56
            for j, r in enumerate(k):
                y, x, v = [int(t) for t in r]
                want[i, j] = y * 56 + x if (v == 2 and 0 <= y < 56 and 0 <= x < 56) else -1
        S = 256
        sr = np.zeros((S, 4), np.float32); sr[:len(sample_roi)] = sample_roi
        ga = np.full((S,), -1, np.int32); ga[:n_pos] = assign
        got = ops.keypoint_target(_t(kp[None]), _t(sr), _t(ga), _t(np.array([n_pos], np.int32)), S, 64, 56).cpu().numpy()
        np.testing.assert_array_equal(got[:n_pos], want)
        assert np.all(got[n_pos:] == -1)


@pytest.mark.parametrize('seed,G', [(0, 8), (1, 1), (2, 30)])
def test_anchor_target_labels_and_sampling(seed, G):
    rs = np.random.RandomState(seed)
    feat = [(48, 64), (24, 32), (12, 16), (6, 8), (3, 4)]
    H, W = 192, 256
    anchors = ob.fpn_anchors(feat)
    A = anchors.shape[0]
    N, gt_cap = 2, 32
    gt = np.zeros((N, gt_cap, 4), np.float32)
    n_gt = np.array([G, max(G - 1, 1)], np.int32)
    for i in range(N):
        c = rs.uniform(0.1, 0.9, (n_gt[i], 2)) * [H, W]
        hw = np.exp(rs.uniform(np.log(16), np.log(150), (n_gt[i], 2)))
        gt[i, :n_gt[i]] = np.concatenate([np.maximum(c - hw / 2, 0), np.minimum(c + hw / 2, [H, W])], 1)
    gt[0, 0] = anchors[np.where((anchors[:, 0] >= 0) & (anchors[:, 1] >= 0) & (anchors[:, 2] <= H) & (anchors[:, 3] <= W))[0][7]]
    keys = rs.randint(0, 2 ** 32, (N, A), dtype=np.uint64).astype(np.uint32)
    # 1. labels before sampling (bit-exact), loc targets
    loc, label = ops.anchor_target(_t(anchors), _t(gt), _t(n_gt), (H, W), keys=None)
    atc = ot.AnchorTargetCreator()
    for i in range(N):
        inside, a, argmax, _, lab = atc.labels_before_sampling(gt[i, :n_gt[i]], anchors, (H, W))
        full = np.full(A, -1, np.int32); full[inside] = lab
        np.testing.assert_array_equal(label[i].cpu().numpy(), full)
        want_loc = np.zeros((A, 4), np.float32); want_loc[inside] = ob.bbox2loc(a, gt[i, :n_gt[i]][argmax])
        np.testing.assert_allclose(loc[i].cpu().numpy(), want_loc, rtol=1e-5, atol=1e-5)
    # 2. key-driven subsampling
    loc, label = ops.anchor_target(_t(anchors), _t(gt), _t(n_gt), (H, W), keys=_t(keys.view(np.int32)))
    for i in range(N):
        wl, wlab = ot.anchor_targets_from_keys(gt[i, :n_gt[i]], anchors, (H, W), keys[i])
        got = label[i].cpu().numpy()
        np.testing.assert_array_equal(got, wlab)
        assert (got == 1).sum() <= 128 and (got >= 0).sum() <= 256


def test_anchor_target_per_image_sizes_equal_separate_calls():
    """Padded batch, images of different sizes: the 'anchor inside the image' test uses each image's OWN size
    (AnchorTargetCreator is called with that image's img_size in the reference, fpn_maskrcnn_train_chain.py:81)."""
    rs = np.random.RandomState(3)
    feat = [(48, 64), (24, 32), (12, 16), (6, 8), (3, 4)]
    anchors = ob.fpn_anchors(feat)
    A = anchors.shape[0]
    sizes = np.array([[192, 256], [150, 201]], np.float32)
    N, gt_cap = 2, 8
    gt = np.zeros((N, gt_cap, 4), np.float32)
    n_gt = np.array([5, 3], np.int32)
    for i in range(N):
        c = rs.uniform(0.2, 0.8, (n_gt[i], 2)) * sizes[i]
        hw = np.exp(rs.uniform(np.log(16), np.log(90), (n_gt[i], 2)))
        gt[i, :n_gt[i]] = np.concatenate([np.maximum(c - hw / 2, 0), np.minimum(c + hw / 2, sizes[i])], 1)
    keys = rs.randint(0, 2 ** 32, (N, A), dtype=np.uint64).astype(np.uint32)
    loc, label = ops.anchor_target(_t(anchors), _t(gt), _t(n_gt), (192, 256), keys=_t(keys.view(np.int32)), per_image_hw=_t(sizes))
    for i in range(N):
        wl, wlab = ot.anchor_targets_from_keys(gt[i, :n_gt[i]], anchors, tuple(sizes[i]), keys[i])
        np.testing.assert_array_equal(label[i].cpu().numpy(), wlab)
        np.testing.assert_allclose(loc[i].cpu().numpy(), wl, rtol=1e-5, atol=1e-5)
    padded = ops.anchor_target(_t(anchors), _t(gt), _t(n_gt), (192, 256), keys=_t(keys.view(np.int32)))[1]
    assert not np.array_equal(padded[1].cpu().numpy(), label[1].cpu().numpy())
