"""GPU parity tests of the legacy model variants (C4Backbone, Darknet, LightRoIMaskHead,
ResnetRoIMaskHead; forward only) against oracle/legacy.py (float64), and of the helper kernels they add."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import legacy as ol
from oracle import model as om
from oracle.proposal import ProposalCreator

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402
from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.nn import core  # noqa: E402

DEV = 'cuda:0'
D = torch.float64


def _rel(got, want):
    want = want.detach().double()
    return float((got.detach().double().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def _params(m):
    return {n: m.ps.p(n).detach().cpu().to(D) for n in m.ps.names()}


def _img(seed, H, W):
    x = np.random.RandomState(seed).rand(1, 3, H, W).astype(np.float32)
    img4 = torch.cat([torch.from_numpy(x).permute(0, 2, 3, 1), torch.zeros((1, H, W, 1))], -1).to(D)
    return torch.from_numpy(x).to(DEV), img4


def test_helper_kernels():
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 13, 18, 8), generator=g)
    got = ops.maxpool3x3s2_fwd(x.to(DEV)).cpu()
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, ceil_mode=True).permute(0, 2, 3, 1)
    assert got.shape == want.shape == (2, 6, 9, 8) and torch.equal(got, want)        # (13-3+1)//2+1 = 6, windows over the edge
    y = torch.randn((5, 7, 7, 12), generator=g)
    gap = ops.global_avg_pool(y.to(DEV)).cpu()
    assert (gap.double() - y.double().mean(dim=(1, 2))).abs().max() < 1e-6
    assert torch.equal(ops.relu(x.to(DEV)).cpu(), x.clamp_min(0))
    s = torch.randn((3, 11, 2), generator=g)
    assert (ops.softmax2(s.to(DEV)).cpu().double() - torch.softmax(s.double(), -1)).abs().max() < 1e-6


def test_rect_conv_matches_torch():
    from chainer_maskrcnn.model.head.light_roi_mask_head import RectConv
    from chainer_maskrcnn.nn.core import ParamStore
    ps = ParamStore()
    cv = [RectConv(ps, 'a', 64, 40, (15, 1), (7, 0)), RectConv(ps, 'b', 64, 40, (1, 15), (0, 7)), RectConv(ps, 'c', 32, 32, (5, 3), (2, 1))]
    ps.materialise(torch.device(DEV), 3)
    g = torch.Generator().manual_seed(1)
    for c in cv:
        x = torch.randn((2, 9, 20, c.cin_p), generator=g)
        ps.p(c.name + '/b').copy_(torch.randn((c.cout_p,), generator=g))
        got = c(x.to(DEV))
        want = ol.conv_rect(x.double(), ps.p(c.name + '/W').cpu().double(), ps.p(c.name + '/b').cpu().double(), (c.ph, c.pw))
        assert got.shape == want.shape == (2, 9, 20, c.cout_p)
        assert _rel(got, want) < 2e-5


@pytest.mark.parametrize('backbone,head', [('c4', 'res5'), ('c4', 'light'), ('darknet', 'light')])
def test_legacy_forward_matches_oracle(backbone, head):
    """MaskRCNN(backbone, head_arch) constructs like the reference (maskrcnn.py:58-98) and forward_legacy() matches the
    oracle: extractor output, RPN outputs + proposals (the oracle's ProposalCreator on the device's RPN outputs), head
    outputs on the device's RoIs.  Width-reduced ResNet stages keep the float64 oracle to seconds."""
    shrink = dict(stages=(1, 1, 1), width_div=2) if backbone == 'c4' else None      # channel counts stay multiples of 32
    m = MaskRCNN(n_fg_class=5, backbone=backbone, head_arch=head, device=DEV, seed=9, _test_shrink=shrink)
    with pytest.raises(TypeError):
        m(torch.zeros((1, 3, 64, 64), device=DEV))              # the reference's __call__ cannot drive these heads either
    H, W = 160, 192
    xd, img4 = _img(4, H, W)
    m.train, core.TRAIN = True, True
    roi_cls_locs, roi_scores, mask, rois, roi_indices = m.forward_legacy(xd, 1.0)
    R = rois.shape[0]
    n_class = 6
    assert R > 0 and roi_scores.shape == (R, n_class) and mask.shape == (R, n_class - 1, 14, 14)
    assert roi_cls_locs.shape == (R, n_class * 4 if head == 'res5' else 4)
    p = _params(m)
    with torch.no_grad():
        feat = ol.c4_backbone(p, img4, (1, 1, 1)) if backbone == 'c4' else ol.darknet(p, img4)
        dev_feat = m.extractor(m.to_nhwc4(xd))[0]
        assert tuple(dev_feat.shape[1:3]) == (H // 16, W // 16) and _rel(dev_feat, feat) < 1e-3
        # RPN: conv + fused heads on the oracle features; proposals from the DEVICE's outputs (index work)
        m.rpn.train = True
        out = m.rpn((dev_feat,), (H, W), 1.0)
        locs, scores = out[0], out[1]
        o = om.OracleStep(p, (), n_class, 0)
        wl, wsc = o.rpn([feat])
        assert _rel(locs, wl) < 1e-3 and _rel(scores, wsc) < 1e-3
        sc = scores[0].cpu().numpy()
        fg = sc[:, 1]
        if backbone == 'c4':            # ChainerCV's single-level RPN ranks by the softmax foreground probability
            e = np.exp(sc - sc.max(1, keepdims=True))
            fg = (e / e.sum(1, keepdims=True))[:, 1].astype(np.float32)
        anchors = out[4]
        kw = dict(n_test_pre_nms=50, n_test_post_nms=10) if backbone == 'darknet' else {}
        want_rois = ProposalCreator(**kw)(locs[0].cpu().numpy(), fg, anchors.cpu().numpy(), (H, W), 1.0, train=True)
        assert abs(len(want_rois) - R) <= 3
        k = min(R, len(want_rois), 30)
        np.testing.assert_allclose(rois.cpu().numpy()[:k], want_rois[:k], rtol=1e-5, atol=1e-3)
        # head on the device's RoIs
        fn = ol.res5_head if head == 'res5' else ol.light_head
        scale = 1. / 16
        wloc, wscore, wmask = fn(p, feat, rois.cpu().numpy(), roi_indices.cpu().numpy(), scale, n_class)
        assert _rel(roi_scores, wscore) < 1e-3 and _rel(roi_cls_locs, wloc) < 1e-3 and _rel(mask, wmask) < 1e-3
    if head == 'light':                 # inference mode: two passes, mask from the cached thin feature map
        m.train = False
        try:
            r = m.forward_legacy(xd, 1.0)
            assert len(r) == 4
            mk = m.head.predict_mask(r[2][:3], r[3][:3], 1. / 16)
            assert mk.shape == (3, n_class - 1, 14, 14)
        finally:
            m.train = True
