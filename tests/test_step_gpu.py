"""GPU parity test of the WHOLE training step: device losses and every parameter gradient against the CPU
oracle (oracle/model.py: float64 torch autograd + NumPy ROIAlign) on a width/depth-reduced network, with the
device's sampled targets fed to the oracle (the samplers are pinned in test_targets_gpu.py / test_rpn_gpu.py).
Tolerance: 1e-3 relative (BASELINE.json north_star); observed ~1e-5."""
import numpy as np
import pytest
import torch

from oracle.model import OracleStep, D

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss, calc_keypoint_loss  # noqa: E402
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay  # noqa: E402
from chainer_maskrcnn.utils.synthetic import make_batch  # noqa: E402

DEV = 'cuda:0'
STAGES = (2, 1, 1, 1)


def _build(mask_rows, seed=7):
    m = MaskRCNN(n_fg_class=80, device=DEV, seed=seed, _test_shrink=dict(stages=STAGES, width_div=2))
    chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss, mask_rows=mask_rows)
    return m, chain


def _batch(N=2, H=128, W=160, G=3):
    b = make_batch(3, N, H, W, G=G)
    b['bboxes'][:, :, 2:] = np.minimum(b['bboxes'][:, :, 2:], [H, W])
    return {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}


@pytest.mark.parametrize('mask_rows', ['positives', 'all'])
def test_step_losses_and_gradients_match_oracle(mask_rows):
    m, chain = _build(mask_rows)
    b = _batch()
    loss = chain(b['imgs'], b['bboxes'], b['labels'], b['masks'], 1.0)
    loss.backward()
    obs = {k: float(v) for k, v in chain.observation.items()}
    assert abs(float(loss.detach()) - sum(obs[k] for k in obs if k != 'loss')) < 1e-5
    # ---- oracle on the same weights and the device's targets
    ps = m.ps
    params = {n: ps.p(n).detach().cpu().to(D).requires_grad_(True) for n in ps.names()}
    t = {k: v.cpu().numpy() for k, v in chain.targets.items() if torch.is_tensor(v)}
    t['gt_rpn_loc'], t['gt_rpn_label'] = (x.cpu().numpy() for x in chain.rpn_targets)
    t['mask_rois_xy5'], t['mask_levels'], t['mask_label'] = (x.cpu().numpy() for x in chain.mask_inputs)
    assert (t['gt_roi_label'] > 0).sum() >= 4 and (t['gt_rpn_label'] == 1).sum() >= 2      # the case exercises every loss
    oracle = OracleStep(params, STAGES, m.head.n_class, m.head.LOC0)
    img4 = torch.cat([b['imgs'].cpu().permute(0, 2, 3, 1), torch.zeros((2, 128, 160, 1))], -1).to(D)
    out = oracle.losses(img4, t)
    names = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
    for k in names:
        ok_ = float(out[k].detach())
        assert abs(obs[k] - ok_) <= 1e-4 * max(abs(ok_), 1e-3), (k, obs[k], ok_)
    sum(out[k] for k in names).backward()
    worst = 0.0
    gmax = max(float(params[n].grad.abs().max()) for n in ps.names() if params[n].grad is not None)
    for n in ps.names():
        want = params[n].grad
        want = torch.zeros_like(params[n]) if want is None else want
        got = ps.g(n).cpu().to(D)
        # biases in front of a BatchNorm have an exactly-zero gradient: floor the scale at 1e-3 of the largest gradient
        scale = max(float(want.abs().max()), 1e-3 * gmax)
        err = float((got - want).abs().max()) / scale
        worst = max(worst, err)
        assert err < 1e-3, (n, err, scale)
    print('worst relative gradient error', worst)


def test_keypoint_step_matches_oracle():
    """BASELINE config 5 head (train_keypoints.py): 17 keypoints, 56x56 heat maps, softmax CE over positions."""
    K, NMC = 17, 2
    m = MaskRCNN(n_fg_class=1, n_keypoints=K, n_mask_convs=NMC, head_arch='fpn_keypoint', device=DEV, seed=11,
                 _test_shrink=dict(stages=(1, 1, 1, 1), width_div=2))
    chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_keypoint_loss, binary_mask=False)
    b = make_batch(5, 2, 128, 160, G=3, n_fg_class=1, n_keypoints=K)
    b['bboxes'][:, :, 2:] = np.minimum(b['bboxes'][:, :, 2:], [128, 160])
    b = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    loss = chain(b['imgs'], b['bboxes'], b['labels'], b['keypoints'], 1.0)
    loss.backward()
    obs = {k: float(v) for k, v in chain.observation.items()}
    assert m.head.mask_size == 56 and chain.targets['gt_roi_mask'].shape[1] == K
    ps = m.ps
    params = {n: ps.p(n).detach().cpu().to(D).requires_grad_(True) for n in ps.names()}
    t = {k: v.cpu().numpy() for k, v in chain.targets.items() if torch.is_tensor(v)}
    t['gt_rpn_loc'], t['gt_rpn_label'] = (x.cpu().numpy() for x in chain.rpn_targets)
    t['mask_rois_xy5'], t['mask_levels'], t['mask_label'] = (x.cpu().numpy() for x in chain.mask_inputs)
    assert (t['gt_roi_mask'] >= 0).sum() > 10
    oracle = OracleStep(params, (1, 1, 1, 1), m.head.n_class, m.head.LOC0,
                        mask_conv_names=['mask_convs/%d' % i for i in range(NMC)], n_keypoints=K)
    img4 = torch.cat([b['imgs'].cpu().permute(0, 2, 3, 1), torch.zeros((2, 128, 160, 1))], -1).to(D)
    out = oracle.losses(img4, t)
    names = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
    for k in names:
        assert abs(obs[k] - float(out[k].detach())) <= 1e-4 * max(abs(float(out[k].detach())), 1e-3), (k, obs[k], float(out[k].detach()))
    sum(out[k] for k in names).backward()
    # fp32 noise floor of this (tiny-batch BatchNorm) network: the same oracle evaluated in float32
    from oracle import model as om
    om.set_dtype(torch.float32)
    try:
        p32 = {n: ps.p(n).detach().cpu().requires_grad_(True) for n in ps.names()}
        o32 = OracleStep(p32, (1, 1, 1, 1), m.head.n_class, m.head.LOC0,
                         mask_conv_names=['mask_convs/%d' % i for i in range(NMC)], n_keypoints=K)
        out32 = o32.losses(img4.float(), t)
        sum(out32[k] for k in names).backward()
    finally:
        om.set_dtype(torch.float64)
    gmax = max(float(params[n].grad.abs().max()) for n in ps.names() if params[n].grad is not None)
    for n in ps.names():
        want = params[n].grad if params[n].grad is not None else torch.zeros_like(params[n])
        w32 = p32[n].grad if p32[n].grad is not None else torch.zeros_like(p32[n])
        scale = max(float(want.abs().max()), 1e-3 * gmax)
        err = float((ps.g(n).cpu().to(D) - want).abs().max()) / scale
        floor = float((w32.to(D) - want).abs().max()) / scale
        assert err < max(1e-3, 3 * floor), (n, err, floor)


def test_step_is_bit_reproducible_and_sgd_updates():
    m, chain = _build('positives')
    b = _batch()
    chain.sampler_keys = None
    chain.proposal_target_creator.set_seed(5)
    chain.anchor_target_creator.set_seed(9)
    chain(b['imgs'], b['bboxes'], b['labels'], b['masks'], 1.0).backward()
    g1 = m.ps.grads.clone()
    l1 = float(chain.observation['loss'])
    chain.proposal_target_creator.set_seed(5)
    chain.anchor_target_creator.set_seed(9)
    chain(b['imgs'], b['bboxes'], b['labels'], b['masks'], 1.0).backward()
    assert float(chain.observation['loss']) == l1
    assert torch.equal(g1, m.ps.grads)                      # no atomics anywhere on the path
    opt = MomentumSGD(lr=1e-2, momentum=0.9).setup(chain)
    opt.add_hook(WeightDecay(0.0005))
    p0 = m.ps.params.clone()
    first = None
    for _ in range(12):
        opt.update(chain, b['imgs'], b['bboxes'], b['labels'], b['masks'], 1.0)
        first = first if first is not None else float(chain.observation['loss'])
    assert not torch.equal(p0, m.ps.params)
    assert float(chain.observation['loss']) < first        # it trains
    assert torch.isfinite(m.ps.params).all()


def test_reference_api_surface():
    m, chain = _build('positives')
    with pytest.raises(ValueError):
        MaskRCNN(n_fg_class=None)
    with pytest.raises(ValueError):
        MaskRCNN(n_fg_class=3, backbone='vgg')
    with pytest.raises(ValueError):
        MaskRCNN(n_fg_class=3, head_arch='nope')
    strict = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss, strict_batch1=True)
    b = _batch()
    with pytest.raises(ValueError):
        strict(b['imgs'], b['bboxes'], b['labels'], b['masks'], 1.0)
    assert m.n_class == 81 and m.head.mask_size == 28
    assert m.extractor.feat_strides == [4, 8, 16, 32, 64] and m.extractor.anchor_sizes == [32, 64, 128, 256, 512]
    m.use_preset('evaluate')
    assert (m.nms_thresh, m.score_thresh) == (0.3, 0.05)
    # reference forward signature (maskrcnn.py:135-155), train mode
    m.train = True
    roi_cls_locs, roi_scores, rois, roi_indices, mask = m(b['imgs'][:1], 1.0)
    R = rois.shape[0]
    assert roi_cls_locs.shape == (R, 4) and roi_scores.shape == (R, 81) and mask.shape == (R, 80, 28, 28)
    assert roi_indices.shape == (R,) and R <= 2000


@pytest.mark.parametrize('tile', [2])
def test_step_matches_oracle_with_winograd_everywhere(tile):
    """The same whole-step parity check with the Winograd thresholds lowered so that every 3x3 / stride-1 layer of the
    small test network (ResNet conv2's, FPN, RPN, box and mask heads) takes the F(2x2,3x3) / F(4x4,3x3) kernels in all
    three passes."""
    from chainer_maskrcnn import _hip
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(32, 64, tile))
    try:
        test_step_losses_and_gradients_match_oracle('all')
    finally:
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, 0))
