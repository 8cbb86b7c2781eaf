"""GPU parity tests: fused loss kernels (loss.hip, through the C ABI) against the NumPy oracle
(oracle/losses.py).  Floating point => tolerance 1e-5 relative on the loss, 1e-5 absolute-relative on
gradients (the north-star bar is 1e-3)."""
import numpy as np
import pytest
import torch

from oracle import losses as ol

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402

DEV = 'cuda:0'


def _close(got, want, tol=1e-5):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert np.abs(got - want).max() <= tol * max(np.abs(want).max(), 1e-6), np.abs(got - want).max()


@pytest.mark.parametrize('M,K,ld', [(1000, 2, 2), (37, 81, 96), (5, 3136, 3136), (0, 2, 2), (300000, 2, 2)])
def test_softmax_cross_entropy(M, K, ld):
    rs = np.random.RandomState(M + K)
    x = (rs.standard_normal((M, ld)) * 3).astype(np.float32)
    t = rs.randint(-1, K, M).astype(np.int32)
    loss, g = ol.softmax_cross_entropy(x[:, :K], t) if M else (np.float32(0), np.zeros((0, K), np.float32))
    xd = torch.from_numpy(x).to(DEV)
    out, gx = ops.softmax_ce(xd, torch.from_numpy(t).to(DEV), M, K, (1, ld, 0, 1), Kfill=ld if ld > K else 0)
    out = out.cpu().numpy()
    _close(out[0], loss)
    assert out[1] == max((t != -1).sum(), 1)
    if M:
        _close(gx.cpu().numpy()[:, :K], g)
        assert np.all(gx.cpu().numpy()[:, K:] == 0)


def test_softmax_cross_entropy_all_ignored():
    x = torch.randn((10, 2), device=DEV)
    t = torch.full((10,), -1, dtype=torch.int32, device=DEV)
    out, gx = ops.softmax_ce(x, t, 10, 2, (1, 2, 0, 1))
    assert out[0].item() == 0 and torch.all(gx == 0)


@pytest.mark.parametrize('R,HW,K,Cm', [(3, 49, 17, 32), (5, 3136, 17, 32), (1300, 64, 17, 32), (4, 200, 3, 4), (2, 777, 40, 64), (2, 100, 17, 20)])
def test_softmax_cross_entropy_keypoint_layout(R, HW, K, Cm):
    """rows = (roi, keypoint), elements strided over NHWC positions (train_keypoints.py:21-27): the one-wave-per-row kernel (HW < 64
    or a channel count the coalesced kernel does not take) and the channel-interleaved kernel (one workgroup per RoI, float4 loads
    over the (position, channel) plane, loss partials and gradient from one launch, more RoIs than partial slots); with the
    coalesced kernel the callee writes the WHOLE gradient tensor - zeros in the padded channels and the ignored rows - so a
    NaN-filled buffer must come back finite."""
    rs = np.random.RandomState(HW + K)
    x = (3.0 * rs.standard_normal((R, HW, Cm))).astype(np.float32)
    x[0, HW // 2, 0] = 60.0                                    # a dominant logit: the running maximum moves late
    t = rs.randint(-1, HW, (R, K)).astype(np.int32)
    t[-1] = -1                                                 # a RoI whose rows are all ignored
    logical = x[:, :, :K].transpose(0, 2, 1).reshape(R * K, HW)
    loss, g = ol.softmax_cross_entropy(logical, t.reshape(-1))
    xd = torch.from_numpy(x).to(DEV)
    xmap = (K, HW * Cm, 1, Cm)
    fills = ops.softmax_ce_fills_gradient(R * K, HW, xmap)
    assert fills == (HW >= 64 and Cm in (4, 32, 64))
    gx = torch.full_like(xd, float('nan')) if fills else torch.zeros_like(xd)
    out, gx = ops.softmax_ce(xd, torch.from_numpy(t.reshape(-1)).to(DEV), R * K, HW, xmap, gx=gx)
    _close(out[0].item(), loss)
    assert out[1].item() == max(1, int((t != -1).sum()))
    got = gx.cpu().numpy()
    assert np.isfinite(got).all()
    assert (got[:, :, K:] == 0).all() and (got[-1] == 0).all()
    _close(got[:, :, :K].transpose(0, 2, 1).reshape(R * K, HW), g)
    out2, _ = ops.softmax_ce(xd, torch.from_numpy(t.reshape(-1)).to(DEV), R * K, HW, xmap, want_grad=False)      # loss only
    assert out2[0].item() == out[0].item()


@pytest.mark.parametrize('sigma', [1.0, 3.0])
@pytest.mark.parametrize('M,ld', [(500, 4), (64, 32), (0, 4)])
def test_smooth_l1(sigma, M, ld):
    rs = np.random.RandomState(int(sigma) + M)
    x = rs.standard_normal((M, ld)).astype(np.float32)
    t = rs.standard_normal((M, 4)).astype(np.float32)
    label = rs.randint(-1, 3, M).astype(np.int32)
    if M:
        loss, g = ol.fast_rcnn_loc_loss(x[:, :4], t, label, sigma)
    out, gx = ops.smooth_l1(torch.from_numpy(x).to(DEV), ld, torch.from_numpy(t).to(DEV), torch.from_numpy(label).to(DEV),
                            M, sigma, gfill=ld)
    if M:
        _close(out[0].item(), loss)
        _close(gx.cpu().numpy()[:, :4], g)
        assert np.all(gx.cpu().numpy()[:, 4:] == 0)


def test_mask_bce_matches_calc_mask_loss():
    rs = np.random.RandomState(3)
    R, S, Cm, n_pos, n_cls = 12, 28, 96, 5, 80
    x = rs.standard_normal((R, S, S, Cm)).astype(np.float32) * 2
    label = np.zeros(R, np.int32)
    label[:n_pos] = rs.randint(1, n_cls + 1, n_pos)
    gt = rs.randint(0, 2, (R, S, S)).astype(np.int32)
    gt[n_pos:] = -1
    nchw = x[:, :, :, :n_cls].transpose(0, 3, 1, 2)
    loss, g = ol.calc_mask_loss(nchw, gt[:n_pos], label)
    out, gx = ops.mask_bce(torch.from_numpy(x).to(DEV), torch.from_numpy(gt).to(DEV), torch.from_numpy(label).to(DEV))
    _close(out[0].item(), loss)
    assert out[1].item() == n_pos * S * S
    got = gx.cpu().numpy()
    _close(got[:, :, :, :n_cls].transpose(0, 3, 1, 2), g)
    assert np.all(got[:, :, :, n_cls:] == 0)
