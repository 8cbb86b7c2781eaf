"""CPU pin of oracle/model.py (the float64 restatement the device step is compared with in tests/test_step_gpu.py and
tests/test_full_width_gpu.py) against a training-step forward EXECUTED BY THE REFERENCE'S OWN MODEL CODE
(tests/golden/make_step_reference.py -> tests/golden/step_reference.npz: MaskRCNN.__init__, FeaturePyramidNetwork,
MultilevelRegionProposalNetwork, FPNRoIMaskHead, FPNMaskRCNNTrainChain.__call__, ProposalTargetCreator, calc_mask_loss,
run on float64 stand-ins of the third-party Chainer / ChainerCV primitives).

The weights are the seeded Chainer-layout arrays of tests/golden/weights.py, brought into the product's storage by
ChainerNpzMap.from_chainer (utils/chainer_npz.py) - so the mapping is pinned by the same comparison."""
import os
import sys

import numpy as np
import torch

from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap
from oracle import boxes as oboxes
from oracle import proposal as oproposal
from oracle.model import OracleStep, D

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from weights import chainer_weights      # noqa: E402


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


def load_step_golden(golden_dir, name='step_reference.npz'):
    d = dict(np.load(os.path.join(golden_dir, name)))
    if 'in_mask' in d:
        shape = tuple(d['in_mask_shape'])
        d['in_mask'] = np.unpackbits(d['in_mask'], axis=-1)[..., :shape[-1]].reshape(shape)
    return d


def oracle_targets(d):
    """The reference-sampled targets in the form OracleStep.losses takes (see tests/test_step_gpu.py)."""
    n_pos = d['gt_roi_mask'].shape[0]
    xy5 = np.ascontiguousarray(d['indices_and_rois'][:, [0, 2, 1, 4, 3]])         # roi_align_2d_yx.py:4-7
    return dict(gt_rpn_label=d['gt_rpn_label'][None], gt_rpn_loc=d['gt_rpn_loc'][None], rois_xy5=xy5, sample_levels=d['sample_levels'],
                gt_roi_label=d['gt_roi_label'].astype(np.int64), gt_roi_loc=d['gt_roi_loc'],
                mask_rois_xy5=xy5[:n_pos], mask_levels=d['sample_levels'][:n_pos], mask_label=d['gt_roi_label'][:n_pos].astype(np.int64),
                gt_roi_mask=d['gt_roi_mask'].astype(np.int64))


def test_oracle_step_equals_reference_executed_step(golden_dir):
    d = load_step_golden(golden_dir)
    m = MaskRCNN(n_fg_class=80, device='cpu', seed=1)
    weights = chainer_weights(int(d['in_weight_seed']))
    loaded = ChainerNpzMap(m).from_chainer(weights, strict=False)
    assert set(loaded) == set(weights), set(weights) ^ set(loaded)              # every array reached the model
    ps = m.ps
    params = {n: ps.p(n).detach().to(D) for n in ps.names()}
    step = OracleStep(params, (3, 4, 6, 3), m.head.n_class, m.head.LOC0)
    img = torch.from_numpy(d['in_img'])
    img4 = torch.cat([img.permute(0, 2, 3, 1), torch.zeros((1,) + img.shape[2:] + (1,))], -1).to(D)
    with torch.no_grad():
        out = step.losses(img4, oracle_targets(d))
    # ---- the five losses of fpn_maskrcnn_train_chain.py:86-112.  Float64 on both sides except the float32 ROIAlign both go through.
    for k in ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss'):
        want = float(d['loss_' + k])
        assert abs(float(out[k]) - want) <= 1e-6 * max(abs(want), 1e-3), (k, float(out[k]), want)
    # ---- activations: FPN levels (stored subsampled), RPN outputs in the reference's anchor order, head outputs per sampled RoI
    p2, p3, p4, p5, p6 = (f.permute(0, 3, 1, 2).numpy() for f in out['feats'])
    assert _rel(p6, d['p6']) < 1e-6 and _rel(p5, d['p5']) < 1e-6
    assert _rel(p4[:, ::4], d['p4_sub']) < 1e-6 and _rel(p3[:, ::8, ::2, ::2], d['p3_sub']) < 1e-6 and _rel(p2[:, ::8, ::4, ::4], d['p2_sub']) < 1e-6
    assert _rel(out['locs'].numpy(), d['rpn_locs']) < 1e-6 and _rel(out['scores'].numpy(), d['rpn_scores']) < 1e-6
    box = out['box'].numpy()
    assert _rel(box[:, :m.head.n_class], d['roi_scores']) < 1e-5
    assert _rel(box[:, m.head.LOC0:m.head.LOC0 + 4], d['roi_cls_locs']) < 1e-5
    n_pos = d['gt_roi_mask'].shape[0]
    mk = out['mask'].numpy()                                                       # (n_pos, 28, 28, classes)
    sel = mk[np.arange(n_pos), :, :, d['gt_roi_label'][:n_pos] - 1]
    assert _rel(sel, d['roi_mask_pos']) < 1e-5
    assert _rel(mk[..., :80:16][:, ::2, ::2].transpose(0, 3, 1, 2), d['roi_mask_sub'][:n_pos]) < 1e-5


def test_oracle_keypoint_step_equals_reference_executed_step(golden_dir):
    """train_keypoints.py's model: FPNRoIKeypointHead (8 convolutions in a ChainList, deconvolution, 17 heat maps, F.resize_images
    to 56x56), one foreground class, its calc_mask_loss (soft-max cross-entropy over the positions of every labelled keypoint
    of the positive rows), binary_mask=False targets from the reference's ProposalTargetCreator (incl. its in-place quirk)."""
    d = load_step_golden(golden_dir, 'step_keypoint_reference.npz')
    K, NMC = 17, 8
    m = MaskRCNN(n_fg_class=1, n_keypoints=K, head_arch='fpn_keypoint', device='cpu', seed=1)
    weights = chainer_weights(int(d['in_weight_seed']), n_fg_class=1, n_keypoints=K, n_mask_convs=NMC)
    assert set(ChainerNpzMap(m).from_chainer(weights, strict=False)) == set(weights)
    ps = m.ps
    params = {n: ps.p(n).detach().to(D) for n in ps.names()}
    step = OracleStep(params, (3, 4, 6, 3), m.head.n_class, m.head.LOC0, mask_conv_names=['mask_convs/%d' % i for i in range(NMC)], n_keypoints=K)
    img = torch.from_numpy(d['in_img'])
    img4 = torch.cat([img.permute(0, 2, 3, 1), torch.zeros((1,) + img.shape[2:] + (1,))], -1).to(D)
    t = oracle_targets(d)
    n_pos = d['gt_roi_mask'].shape[0]
    assert d['gt_roi_mask'].shape == (n_pos, K) and int((d['gt_roi_label'][:n_pos] == 1).all())
    with torch.no_grad():
        out = step.losses(img4, t)
    for k in ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss'):
        want = float(d['loss_' + k])
        assert abs(float(out[k]) - want) <= 1e-6 * max(abs(want), 1e-3), (k, float(out[k]), want)
    assert _rel(out['locs'].numpy(), d['rpn_locs']) < 1e-6 and _rel(out['scores'].numpy(), d['rpn_scores']) < 1e-6
    box = out['box'].numpy()
    assert _rel(box[:, :2], d['roi_scores']) < 1e-5 and _rel(box[:, m.head.LOC0:m.head.LOC0 + 4], d['roi_cls_locs']) < 1e-5
    mk = out['mask'].numpy()[..., :K].transpose(0, 3, 1, 2)                           # (n_pos, K, 56, 56)
    assert mk.shape == (n_pos, K, 56, 56)
    assert _rel(mk[:, :, ::4, ::4], d['roi_mask_sub']) < 1e-5
    at = mk.reshape(n_pos, K, -1)[np.arange(n_pos)[:, None], np.arange(K)[None, :], np.maximum(d['gt_roi_mask'], 0)]
    assert _rel(at, d['roi_mask_at_label']) < 1e-5


def test_proposals_and_levels_of_the_reference_step(golden_dir):
    """The RoIs the reference's RPN produced from ITS locs / scores (multilevel_region_proposal_network.py:154-169) = the
    oracle's anchors + ProposalCreator + level map on the stored RPN outputs."""
    d = load_step_golden(golden_dir)
    H, W = d['in_img'].shape[2:]
    half = lambda n: (n - 1) // 2 + 1                     # a stride-2 convolution (7x7 pad 3, 1x1 pad 0)
    pool = lambda n: -(-(n - 2) // 2) + 1                 # max_pooling_2d(2), cover_all
    feat = [(pool(half(H)), pool(half(W)))]
    for _ in range(4):                                    # res3, res4, res5, conv_p6
        feat.append((half(feat[-1][0]), half(feat[-1][1])))
    anchor = oboxes.fpn_anchors(feat)
    np.testing.assert_array_equal(anchor, d['anchor'])
    A = anchor.shape[0]
    fg = d['rpn_scores'][0].reshape(A, 2)[:, 1]
    roi = oproposal.ProposalCreator()(d['rpn_locs'][0], fg, anchor, (H, W), scale=1.0, train=True)
    # the reference ran on float64 locs / scores; here they are the float32 copies: identical unless a score tie or a
    # threshold sits within float32 rounding - the stored case has neither
    np.testing.assert_allclose(roi, d['rois'], rtol=0, atol=1e-3)
    np.testing.assert_array_equal(oboxes.map_rois_to_fpn_levels(d['rois']).astype(np.int32), d['levels'])
