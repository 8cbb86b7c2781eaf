"""Device model and loss kernels against a training-step forward EXECUTED BY THE REFERENCE'S OWN MODEL CODE
(tests/golden/make_step_reference.py -> tests/golden/step_reference.npz; see tests/test_step_reference_cpu.py for the
oracle's pin on the same fixture).  Full-width ResNet-50 FPN, the seeded Chainer-layout weights of tests/golden/weights.py
loaded through ChainerNpzMap.from_chainer, the reference's image; then with the REFERENCE-SAMPLED targets (its
np.random draws cannot be reproduced by the device sampler, so they are fed in):

    extractor features, RPN locs / scores          vs the reference's activations
    proposals + FPN levels from the device RPN     vs the reference's RoIs
    box / mask head on the reference's sampled RoIs  vs its roi_cls_locs / roi_scores / mask logits
    the five losses (the kernels the train chain launches)  vs chainer.reporter's values

Tolerance 1e-3 relative (BASELINE.json north_star) on activations and losses - float32 device against a float64 run."""
import os
import sys

import numpy as np
import pytest
import torch

from chainer_maskrcnn._hip import ops
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.nn import core
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from weights import chainer_weights                       # noqa: E402
from test_step_reference_cpu import load_step_golden      # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-3


def _rel(got, want):
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


@pytest.fixture(scope='module')
def step():
    d = load_step_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    m = MaskRCNN(n_fg_class=80, device=DEV, seed=1)
    weights = chainer_weights(int(d['in_weight_seed']))
    assert set(ChainerNpzMap(m).from_chainer(weights, strict=False)) == set(weights)
    core.TRAIN = True
    m.rpn.train = True
    img = torch.from_numpy(d['in_img']).to(DEV)
    feats = m.extractor(m.to_nhwc4(img))
    r = m.rpn.forward_padded(feats, tuple(img.shape[2:]), 1.0)
    return d, m, feats, r


def test_features_and_rpn_outputs(step):
    d, m, feats, r = step
    p2, p3, p4, p5, p6 = (f.permute(0, 3, 1, 2) for f in feats)
    errs = dict(p6=_rel(p6, d['p6']), p5=_rel(p5, d['p5']), p4=_rel(p4[:, ::4], d['p4_sub']), p3=_rel(p3[:, ::8, ::2, ::2], d['p3_sub']),
                p2=_rel(p2[:, ::8, ::4, ::4], d['p2_sub']), locs=_rel(r['locs'], d['rpn_locs']), scores=_rel(r['scores'], d['rpn_scores']))
    print('step reference, device vs reference-executed activations:', {k: '%.2e' % v for k, v in errs.items()})
    assert max(errs.values()) < TOL, errs
    np.testing.assert_array_equal(r['anchors'].cpu().numpy(), d['anchor'])


def test_proposals_and_levels(step):
    """The device's proposal chain on ITS float32 RPN outputs against the RoIs of the reference's float64 run: the same
    boxes except where a score order or an IoU-threshold decision sits inside the float32 difference of the two runs."""
    d, m, feats, r = step
    n = int(r['n_rois'][0])
    got = r['rois'][0, :n].cpu().numpy() if r['rois'].dim() == 3 else r['rois'][:n].cpu().numpy()
    lev = r['levels'].reshape(-1)[:n].cpu().numpy()
    want, want_lev = d['rois'], d['levels']
    dist = np.abs(got[:, None, :] - want[None, :, :]).max(-1)
    j = dist.argmin(1)
    hit = dist[np.arange(n), j] < 0.05
    print('proposals: device %d, reference %d, matched %d' % (n, len(want), int(hit.sum())))
    assert abs(n - len(want)) <= 0.05 * len(want) and hit.mean() >= 0.95
    np.testing.assert_array_equal(lev[hit], want_lev[j[hit]])
    # rank order (descending score through NMS): the matched boxes appear in the same relative order
    assert np.all(np.diff(j[hit]) > 0) or np.mean(np.diff(j[hit]) > 0) > 0.98


def test_heads_and_losses_on_the_reference_sampled_targets(step):
    d, m, feats, r = step
    head = m.head
    dev = torch.device(DEV)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt).contiguous()
    xy5 = t(d['indices_and_rois'][:, [0, 2, 1, 4, 3]], torch.float32)            # roi_align_2d_yx.py:4-7
    levels = t(d['sample_levels'], torch.int32)
    label = t(d['gt_roi_label'], torch.int32)
    R = xy5.shape[0]
    n_pos = d['gt_roi_mask'].shape[0]
    scales = m.extractor.spatial_scales
    losses = torch.zeros((5, 2), dtype=torch.float32, device=dev)
    # ---- RPN losses (fpn_maskrcnn_train_chain.py:81-85)
    A = d['anchor'].shape[0]
    rl = t(d['gt_rpn_label'], torch.int32)
    ops.smooth_l1(r['locs'].view(A, 4), 4, t(d['gt_rpn_loc'], torch.float32), rl, A, 3.0, out=losses[0])
    ops.softmax_ce(r['scores'].view(A, 2), rl, A, 2, (1, 2, 0, 1), out=losses[1])
    # ---- box head + its losses (:88-101)
    box = head.box_branch(feats, xy5, levels, scales)
    ld = head.out_p
    g_box = torch.empty_like(box)
    ops.softmax_ce(box, label, R, head.n_class, (1, ld, 0, 1), Kfill=head.LOC0, gx=g_box, out=losses[3])
    ops.smooth_l1(box, ld, t(d['gt_roi_loc'], torch.float32), label, R, 1.0, gfill=ld - head.LOC0, col0=head.LOC0, gx=g_box, out=losses[2])
    assert _rel(box[:, :head.n_class], d['roi_scores']) < TOL
    assert _rel(box[:, head.LOC0:head.LOC0 + 4], d['roi_cls_locs']) < TOL
    # ---- mask head on every sampled RoI (the reference's head does that, fpn_roi_mask_head.py:72-84) + calc_mask_loss (train.py:49-57)
    mask = head.mask_branch(feats, xy5, levels, scales)                           # (R, 28, 28, Cp)
    assert _rel(mask[..., :80:16][:, ::2, ::2].permute(0, 3, 1, 2), d['roi_mask_sub']) < TOL
    sel = mask[torch.arange(n_pos, device=dev), :, :, (label[:n_pos] - 1).long()]
    assert _rel(sel, d['roi_mask_pos']) < TOL
    gt = torch.full((R, 28, 28), -1, dtype=torch.int32, device=dev)
    gt[:n_pos] = t(d['gt_roi_mask'], torch.int32)
    m_label = label.clone()
    m_label[n_pos:] = 0                                                           # rows past the positives carry no mask target
    ops.mask_bce(mask, gt, m_label, out=losses[4])
    got = losses[:, 0].cpu().numpy()
    names = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
    print('step reference losses: device', dict(zip(names, got.round(6))), 'reference', {k: round(float(d['loss_' + k]), 6) for k in names})
    for k, v in zip(names, got):
        want = float(d['loss_' + k])
        assert abs(float(v) - want) <= TOL * max(abs(want), 1e-3), (k, float(v), want)
    assert abs(float(ops.loss_total(losses)[0]) - float(d['loss_loss'])) <= TOL * float(d['loss_loss'])


def test_keypoint_model_on_the_reference_executed_keypoint_step():
    """train_keypoints.py's model (FPNRoIKeypointHead: 8 convolutions, deconvolution, 17 heat maps resized to 56x56; one class)
    with the reference-sampled targets of tests/golden/step_keypoint_reference.npz: RPN outputs, box head, keypoint logits of the
    positive rows and the five losses - the keypoint loss through the same soft-max cross-entropy launch the chain makes."""
    d = load_step_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'), 'step_keypoint_reference.npz')
    K = 17
    m = MaskRCNN(n_fg_class=1, n_keypoints=K, head_arch='fpn_keypoint', device=DEV, seed=1)
    weights = chainer_weights(int(d['in_weight_seed']), n_fg_class=1, n_keypoints=K, n_mask_convs=8)
    assert set(ChainerNpzMap(m).from_chainer(weights, strict=False)) == set(weights)
    core.TRAIN = True
    m.rpn.train = True
    dev = torch.device(DEV)
    img = torch.from_numpy(d['in_img']).to(dev)
    feats = m.extractor(m.to_nhwc4(img))
    r = m.rpn.forward_padded(feats, tuple(img.shape[2:]), 1.0)
    assert _rel(r['locs'], d['rpn_locs']) < TOL and _rel(r['scores'], d['rpn_scores']) < TOL
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt).contiguous()
    head = m.head
    xy5 = t(d['indices_and_rois'][:, [0, 2, 1, 4, 3]], torch.float32)
    levels = t(d['sample_levels'], torch.int32)
    label = t(d['gt_roi_label'], torch.int32)
    R, n_pos = xy5.shape[0], d['gt_roi_mask'].shape[0]
    scales = m.extractor.spatial_scales
    losses = torch.zeros((5, 2), dtype=torch.float32, device=dev)
    A = d['anchor'].shape[0]
    rl = t(d['gt_rpn_label'], torch.int32)
    ops.smooth_l1(r['locs'].view(A, 4), 4, t(d['gt_rpn_loc'], torch.float32), rl, A, 3.0, out=losses[0])
    ops.softmax_ce(r['scores'].view(A, 2), rl, A, 2, (1, 2, 0, 1), out=losses[1])
    box = head.box_branch(feats, xy5, levels, scales)
    ld = head.out_p
    g_box = torch.empty_like(box)
    ops.softmax_ce(box, label, R, head.n_class, (1, ld, 0, 1), Kfill=head.LOC0, gx=g_box, out=losses[3])
    ops.smooth_l1(box, ld, t(d['gt_roi_loc'], torch.float32), label, R, 1.0, gfill=ld - head.LOC0, col0=head.LOC0, gx=g_box, out=losses[2])
    assert _rel(box[:, :head.n_class], d['roi_scores']) < TOL and _rel(box[:, head.LOC0:head.LOC0 + 4], d['roi_cls_locs']) < TOL
    # keypoint branch on the positive rows (calc_mask_loss of train_keypoints.py:21-27 uses roi_cls_mask[:n_pos])
    mk = head.mask_branch(feats, xy5[:n_pos].contiguous(), levels[:n_pos].contiguous(), scales)            # (n_pos, 56, 56, Cp)
    Rm, Hm, Wm, Cm = mk.shape
    assert (Rm, Hm, Wm) == (n_pos, 56, 56)
    nchw = mk[..., :K].permute(0, 3, 1, 2)
    assert _rel(nchw[:, :, ::4, ::4], d['roi_mask_sub']) < TOL
    gt = t(d['gt_roi_mask'], torch.int32)
    at = nchw.reshape(n_pos, K, -1).gather(2, gt.clamp(min=0).long()[..., None])[..., 0]
    assert _rel(at, d['roi_mask_at_label']) < TOL
    ops.softmax_ce(mk, gt.view(-1), Rm * K, Hm * Wm, (K, Hm * Wm * Cm, 1, Cm), gx=torch.zeros_like(mk), out=losses[4])
    got = losses[:, 0].cpu().numpy()
    names = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
    print('keypoint step reference losses: device', dict(zip(names, got.round(6))), 'reference', {k: round(float(d['loss_' + k]), 6) for k in names})
    for k, v in zip(names, got):
        want = float(d['loss_' + k])
        assert abs(float(v) - want) <= TOL * max(abs(want), 1e-3), (k, float(v), want)
