"""GPU tests: HBM-bound layer kernels (nn.hip) against plain PyTorch float64 CPU references
(floating-point kernels => torch reference; tolerance 1e-3 relative per BASELINE.json north_star,
observed ~1e-6) and the fused SGD update against the NumPy statement of train.py:107-109."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402

DEV = 'cuda:0'


def _rel(got, ref):
    return (got.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)


@pytest.mark.parametrize('shape', [(2, 9, 7, 64), (1, 5, 6, 256), (2, 3, 3, 2048), (1, 31, 17, 8), (3, 40, 40, 128)])
@pytest.mark.parametrize('relu,res', [(False, False), (True, False), (True, True)])
def test_bn_train_fwd_bwd(shape, relu, res):
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[-1]
    x = (torch.randn(shape, generator=g, dtype=torch.float64) * 2 + 3).requires_grad_(True)   # |mean| > std
    gamma = torch.rand((C,), generator=g, dtype=torch.float64).add_(0.5).requires_grad_(True)
    beta = torch.randn((C,), generator=g, dtype=torch.float64).requires_grad_(True)
    r = torch.randn(shape, generator=g, dtype=torch.float64).requires_grad_(True) if res else None
    xr = x.reshape(-1, C)
    mean, var = xr.mean(0), xr.var(0, unbiased=False)
    y = gamma * (x - mean) / torch.sqrt(var + 2e-5) + beta
    if res:
        y = y + r
    if relu:
        y = y.clamp_min(0)
    gy = torch.randn(shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    rm = torch.zeros((C,), device=DEV)
    rv = torch.ones((C,), device=DEV)
    xd = x.detach().float().to(DEV)
    yd, m, s = ops.bn_train_fwd(xd, gamma.detach().float().to(DEV), beta.detach().float().to(DEV),
                                r.detach().float().to(DEV) if res else None, relu, rm, rv)
    assert _rel(yd, y.detach()) < 1e-5
    assert _rel(m, mean.detach()) < 1e-6
    assert _rel(s, 1 / torch.sqrt(var.detach() + 2e-5)) < 1e-5
    P = xr.shape[0]
    assert _rel(rm, 0.1 * mean.detach()) < 1e-5
    assert _rel(rv, 0.9 + 0.1 * var.detach() * P / max(P - 1, 1)) < 1e-5
    gx, gres, gg, gb = ops.bn_train_bwd(gy.float().to(DEV), xd, yd, gamma.detach().float().to(DEV), m, s, relu, res)
    assert _rel(gx, x.grad) < 2e-5
    assert _rel(gg, gamma.grad) < 2e-5
    assert _rel(gb, beta.grad) < 2e-5
    if res:
        assert _rel(gres, r.grad) < 1e-6
    # bit-reproducible reductions
    gx2, _, gg2, _ = ops.bn_train_bwd(gy.float().to(DEV), xd, yd, gamma.detach().float().to(DEV), m, s, relu, res)
    assert torch.equal(gx, gx2) and torch.equal(gg, gg2)
    if relu and not res:
        # ReLU mask recomputed from x instead of read from y: bitwise the same gradients
        gx3, _, gg3, gb3 = ops.bn_train_bwd(gy.float().to(DEV), xd, None, gamma.detach().float().to(DEV), m, s, relu, False,
                                            beta=beta.detach().float().to(DEV))
        assert torch.equal(gx, gx3) and torch.equal(gg, gg3) and torch.equal(gb, gb3)


@pytest.mark.parametrize('shape', [(2, 8, 8, 16), (1, 9, 7, 64), (2, 5, 5, 8), (1, 1, 1, 4)])
def test_maxpool_cover_all(shape):
    g = torch.Generator().manual_seed(1 + sum(shape))
    x = torch.randn(shape, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2, ceil_mode=True).permute(0, 2, 3, 1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd = x.detach().float().to(DEV)
    yd = ops.maxpool2x2_fwd(xd)
    assert yd.shape == y.shape
    assert torch.equal(yd.cpu(), y.detach().float())
    gx = ops.maxpool2x2_bwd(xd, gy.float().contiguous().to(DEV))
    assert torch.equal(gx.cpu(), x.grad.float())


def test_maxpool_tie_goes_to_first_cell():
    x = torch.zeros((1, 2, 2, 4), device=DEV)
    gy = torch.ones((1, 1, 1, 4), device=DEV)
    gx = ops.maxpool2x2_bwd(x, gy).cpu()
    assert torch.equal(gx[0, 0, 0], torch.ones(4)) and gx.sum().item() == 4


@pytest.mark.parametrize('H,W', [(8, 8), (7, 9), (13, 25)])
def test_upsample_add_and_backward(H, W):
    g = torch.Generator().manual_seed(H * W)
    N, C = 2, 16
    Ht, Wt = (H + 1) // 2, (W + 1) // 2
    top = torch.randn((N, Ht, Wt, C), generator=g, dtype=torch.float64, requires_grad=True)
    lat = torch.randn((N, H, W, C), generator=g, dtype=torch.float64)
    up = top.repeat_interleave(2, 1).repeat_interleave(2, 2)[:, :H, :W]
    out = up + lat
    gout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    out.backward(gout)
    od = ops.upsample2x_add_fwd(top.detach().float().to(DEV), lat.float().to(DEV))
    assert torch.equal(od.cpu(), (up.detach().float() + lat.float()))
    gt = ops.upsample2x_bwd(gout.float().to(DEV), top_shape=tuple(top.shape))
    assert _rel(gt, top.grad) < 1e-6
    base = torch.randn(top.shape, generator=g).to(DEV)
    gt2 = ops.upsample2x_bwd(gout.float().to(DEV), gtop=base.clone())
    assert _rel(gt2, top.grad + base.cpu().double()) < 1e-6


@pytest.mark.parametrize('H,W,s', [(8, 8, 2), (7, 9, 2), (5, 5, 1)])
def test_subsample_bwd(H, W, s):
    g = torch.Generator().manual_seed(3)
    N, C = 2, 8
    x = torch.randn((N, H, W, C), generator=g, dtype=torch.float64, requires_grad=True)
    y = x[:, ::s, ::s]
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    gx = ops.subsample_bwd(gy.float().contiguous().to(DEV), tuple(x.shape), s)
    assert torch.equal(gx.cpu(), x.grad.float())
    base = torch.randn(x.shape, generator=g).to(DEV)
    gx2 = ops.subsample_bwd(gy.float().contiguous().to(DEV), tuple(x.shape), s, gx=base.clone())
    assert _rel(gx2, x.grad + base.cpu().double()) < 1e-6


def test_pixel_shuffle_roundtrip_matches_deconv_layout():
    g = torch.Generator().manual_seed(4)
    N, H, W, C = 2, 3, 5, 8
    t = torch.randn((N, H, W, 4 * C), generator=g)
    want = t.reshape(N, H, W, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(N, 2 * H, 2 * W, C)
    got = ops.pixel_shuffle2x(t.to(DEV))
    assert torch.equal(got.cpu(), want)
    back = ops.pixel_shuffle2x(got, inverse=True)
    assert torch.equal(back.cpu(), t)
    b = torch.randn((C,), generator=g)
    assert torch.equal(ops.pixel_shuffle2x(t.to(DEV), bias=b.to(DEV)).cpu(), want + b)


def test_image_layout_and_random_keys():
    g = torch.Generator().manual_seed(6)
    x = torch.rand((2, 3, 5, 7), generator=g)
    y = ops.image_nchw3_to_nhwc4(x.to(DEV)).cpu()
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1)) and torch.all(y[..., 3] == 0)
    k1 = ops.random_keys((2, 5000), 123, torch.device(DEV)).cpu().numpy().view(np.uint32)
    k2 = ops.random_keys((2, 5000), 123, torch.device(DEV)).cpu().numpy().view(np.uint32)
    k3 = ops.random_keys((2, 5000), 124, torch.device(DEV)).cpu().numpy().view(np.uint32)
    assert np.array_equal(k1, k2) and not np.array_equal(k1, k3)
    assert len(np.unique(k1)) > 9990 and abs(k1.mean() / 2 ** 32 - 0.5) < 0.02


@pytest.mark.parametrize('shape', [(2, 7, 5, 8), (1, 28, 28, 32), (3, 1, 4, 4)])
def test_bilinear2x_matches_resize_images_and_adjoint(shape):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g, dtype=torch.float64, requires_grad=True)
    N, H, W, C = shape
    if H > 1 and W > 1:
        y = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode='bilinear', align_corners=True).permute(0, 2, 3, 1)
    else:
        from oracle.model import resize_images_x2
        y = resize_images_x2(x)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    yd = ops.bilinear2x_fwd(x.detach().float().to(DEV))
    assert _rel(yd, y.detach()) < 1e-6
    gx = ops.bilinear2x_bwd(gy.float().contiguous().to(DEV))
    assert _rel(gx, x.grad) < 1e-6


def test_relu_bwd_and_add():
    g = torch.Generator().manual_seed(5)
    y = torch.randn((1000,), generator=g).clamp_min(0)
    gy = torch.randn((1000,), generator=g)
    got = ops.relu_bwd(gy.to(DEV), y.to(DEV))
    assert torch.equal(got.cpu(), gy * (y > 0))
    a, b = torch.randn((4096,), generator=g), torch.randn((4096,), generator=g)
    assert torch.equal(ops.add(a.to(DEV), b.to(DEV)).cpu(), a + b)


@pytest.mark.parametrize('n', [4096, 1000003])
def test_sgd_momentum_weight_decay(n):
    rs = np.random.RandomState(n)
    p, g, v = (rs.standard_normal(n).astype(np.float32) for _ in range(3))
    lr, mom, wd = np.float32(1e-3), np.float32(0.9), np.float32(5e-4)
    gg = g + wd * p                       # WeightDecay hook (train.py:109)
    v2 = mom * v - lr * gg                # MomentumSGD (train.py:107)
    p2 = p + v2
    pd, gd, vd = (torch.from_numpy(t).to(DEV) for t in (p, g, v))
    ops.sgd_momentum_wd(pd, gd, vd, float(lr), float(mom), float(wd))
    np.testing.assert_allclose(vd.cpu().numpy(), v2, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(pd.cpu().numpy(), p2, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('geom', [(2, 64, 64, 256, 1024, 1, 1, 0),       # res4 conv3
                                  (2, 128, 128, 256, 128, 1, 2, 0),      # strided 1x1 (first block of a stage)
                                  (2, 200, 272, 64, 64, 3, 1, 1),        # 3x3 below the Winograd channel threshold
                                  (1, 37, 53, 64, 96, 1, 1, 0),          # ragged M: the last row block is partial
                                  (2, 64, 64, 256, 256, 3, 1, 1),        # Winograd layer: partials from the output transform
                                  (2, 30, 45, 512, 512, 3, 1, 1)])       # Winograd, ragged tiles
def test_bn_statistics_from_the_convolution_epilogue(geom):
    """the forward GEMM epilogue leaves per-row-block sums / sums of squares (mrcnn_conv2d_fwd_bnstats_f32);
    BatchNorm finishes from them (mrcnn_bn_train_fwd_stats_f32).  Against float64: the convolution output is bitwise the
    plain kernel's, mean / inverse std / BN output within the bars of the two-pass path."""
    from chainer_maskrcnn._hip import nn as hnn
    N, H, W, Ci, Co, k, stride, pad = geom
    rs = np.random.RandomState(7)
    x = (rs.standard_normal((N, H, W, Ci)) + 0.3).astype(np.float32)
    w = (rs.standard_normal((Co, k, k, Ci)) * (1.0 / np.sqrt(Ci * k * k))).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, Co).astype(np.float32)
    beta = rs.standard_normal(Co).astype(np.float32)
    xt, wt = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    res = hnn.conv2d_fwd_bnstats_raw(xt, wt, stride, pad)
    assert res is not None
    y, _, part = res
    y_plain = hnn.conv2d_fwd_raw(xt, wt, None, stride, pad, False)
    assert torch.equal(y, y_plain)
    P = y.numel() // Co
    assert part.shape[1:] == (2, Co)
    yd = y.double().cpu().reshape(P, Co)
    np.testing.assert_allclose(part[:, 0].double().sum(0).cpu().numpy(), yd.sum(0).numpy(), rtol=1e-5, atol=1e-3)
    rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
    o, mean, invstd = ops.bn_train_fwd_stats(y, part, torch.from_numpy(gamma).to(DEV), torch.from_numpy(beta).to(DEV), None, True, rm, rv)
    m64, v64 = yd.mean(0), yd.var(0, unbiased=False)
    want = torch.relu((yd - m64) / torch.sqrt(v64 + 2e-5) * torch.from_numpy(gamma).double() + torch.from_numpy(beta).double())
    assert _rel(mean, m64) <= 2e-6 and _rel(invstd, 1.0 / torch.sqrt(v64 + 2e-5)) <= 2e-5
    assert _rel(o.reshape(P, Co), want) <= 2e-5
    o2, mean2, invstd2 = ops.bn_train_fwd(y, torch.from_numpy(gamma).to(DEV), torch.from_numpy(beta).to(DEV), None, True)
    assert _rel(o, o2.double().cpu()) <= 1e-5
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * m64.float().numpy(), rtol=1e-4, atol=1e-6)


def test_bn_statistics_fusion_declines_split_launches():
    from chainer_maskrcnn._hip import lib
    assert lib().mrcnn_conv2d_bnstats_rows(2, 64, 64, 256, 256, 3, 3, 1, 1) > 0         # Winograd layer: output transform
    assert lib().mrcnn_conv2d_bnstats_rows(2, 8, 8, 2048, 512, 1, 1, 1, 0) == 0         # few tiles, long K: split-K
    assert lib().mrcnn_conv2d_bnstats_rows(2, 64, 64, 256, 1024, 1, 1, 1, 0) > 0


@pytest.mark.parametrize('ratio', [20.0, 80.0, 400.0])
def test_bn_statistics_from_the_epilogue_with_a_large_mean(ratio):
    """the epilogue's partials are UN-shifted float32 sums, so var = E[x^2] - E[x]^2 cancels once |mean| >> std
    (a pretrained backbone's first BatchNorms).  A convolution output with |mean| / std of 20 / 80 / 400 per channel
    (an input with a large constant component): past |mean| / std = 32 k_bn_stats_final recomputes the channel's statistics
    exactly from x, so 1 / sigma stays within 2e-5 of float64 and the normalised output within 1e-4 of its scale - where the
    plain formula would be off by ~1e-7 * ratio^2 (6e-4 at 80, 1.6e-2 at 400)."""
    from chainer_maskrcnn._hip import nn as hnn
    N, H, W, Ci, Co = 2, 64, 64, 64, 128
    rs = np.random.RandomState(11)
    w = (rs.standard_normal((Co, 1, 1, Ci)) / np.sqrt(Ci)).astype(np.float32)
    x = rs.standard_normal((N, H, W, Ci)).astype(np.float32)
    # y = x w^T has unit std per channel; add a per-channel constant of `ratio` stds through a constant input component
    shift = np.linalg.lstsq(w.reshape(Co, Ci).astype(np.float64), np.full(Co, ratio), rcond=None)[0].astype(np.float32)
    x = x + shift
    xt, wt = torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)
    y, _, part = hnn.conv2d_fwd_bnstats_raw(xt, wt, 1, 0)
    P = y.numel() // Co
    yd = y.double().cpu().reshape(P, Co)
    m64, v64 = yd.mean(0), yd.var(0, unbiased=False)
    big = (m64.abs() / v64.sqrt()) > 0.5 * ratio
    assert int(big.sum()) >= Co // 2             # the construction worked: most channels have the intended |mean| / std
    gamma, beta = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
    o, mean, invstd = ops.bn_train_fwd_stats(y, part, gamma, beta, None, False, None, None)
    want_is = 1.0 / torch.sqrt(v64 + 2e-5)
    assert float(((invstd.double().cpu() - want_is).abs() / want_is).max()) <= 2e-5
    assert float((mean.double().cpu() - m64).abs().max()) <= 2e-6 * float(m64.abs().max())
    want = (yd - m64) * want_is
    assert float((o.double().cpu().reshape(P, Co) - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize('case', [(2, 40, 56, 64, 64, 256, 1, True), (2, 40, 56, 256, 64, 256, 1, False), (2, 41, 57, 256, 128, 512, 2, True),
                                  (2, 64, 64, 1024, 256, 1024, 1, False), (1, 64, 64, 512, 256, 1024, 2, True)],
                         ids=['res2a', 'res2b', 'res3a_odd', 'res4b_winograd', 'res4a'])
@pytest.mark.parametrize('flags', [(False, False, False), (True, True, False), (True, False, True), (False, True, True)],
                         ids=['plain', 'masked_in_out', 'masked_acc', 'mask_gx_acc'])
def test_composite_bottleneck_equals_the_per_layer_path(case, flags):
    """mrcnn_bottleneck_fwd_f32 / _bwd_f32 (csrc/blocks.hip) against the per-layer calls of nn/core.py Bottleneck on full-width blocks
    (identity and projection shortcuts, stride 1 and 2 on odd maps, a Winograd conv2 with its kept input transform): output, saved
    statistics, every gradient and the input gradient are the same bits for every combination of gy_masked / mask_gx / gx_acc."""
    from chainer_maskrcnn.nn import core
    N, H, W, cin, mid, cout, stride, project = case
    gy_masked, mask_gx, with_acc = flags
    ps = core.ParamStore()
    blk = core.Bottleneck(ps, 'b', cin, mid, cout, stride, project)
    ps.materialise(torch.device(DEV), seed=11)
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.relu(torch.randn((N, H, W, cin), generator=g)).to(DEV)            # a ReLU output, as in the network
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    gy0 = (torch.randn((N, Ho, Wo, cout), generator=g) * 1e-3).to(DEV)
    acc0 = (torch.randn((N, H, W, cin), generator=g) * 1e-3).to(DEV)
    out = []
    for on in (False, True):
        core.COMPOSITE_BLOCKS = on
        try:
            ps.grads.zero_()
            for k in ps.buffers:
                ps.buffers[k].copy_(torch.zeros_like(ps.buffers[k]) if k.endswith('avg_mean') else torch.ones_like(ps.buffers[k]))
            y, ctx = blk.fwd(x)
            assert (ctx[0] == 'composite') == on
            gy = gy0.clone()
            if gy_masked:
                gy = gy * (y > 0)
            acc = acc0.clone() if with_acc else None
            gx = blk.bwd(ctx, gy, gx_acc=acc, gy_masked=gy_masked, mask_gx=mask_gx)
            core.join_side_stream(torch.device(DEV))
            torch.cuda.synchronize()
            out.append((y.clone(), gx.clone(), ps.grads.clone(), torch.cat([v.flatten() for _, v in sorted(ps.buffers.items())])))
        finally:
            core.COMPOSITE_BLOCKS = True
    for a, b_ in zip(out[0], out[1]):
        assert torch.isfinite(a).all() and torch.equal(a, b_)
    assert float(out[0][2].abs().max()) > 0 and float(out[0][1].abs().max()) > 0


@pytest.mark.parametrize('P,C', [(4096, 64), (2 * 33 * 41, 256), (2048, 2048)])
@pytest.mark.parametrize('masked', [False, True], ids=['mask_from_y', 'gy_arrives_masked'])
def test_bn_pair_kernels_equal_the_layer_by_layer_sequence(P, C, masked):
    """(ABI v10) relu(BN_a(xa) + BN_b(xb)) in one apply kernel and both backward passes from one read of gy (nn.hip k_bn_apply2,
    k_bn_bwd_partial2 / _final2 / _apply2: the projection block of extractor/feature_pyramid_network.py:48-66) against the single-layer
    entry points run one after the other: output, saved statistics, running statistics and all four parameter gradients and both input
    gradients are the same BITS (statistics pass over the tensors here: part = NULL; the composite test covers the partial-row form)."""
    from chainer_maskrcnn import _hip
    lib, check, ptr, sp = _hip.lib(), _hip.check, _hip.ptr, _hip.stream_ptr
    from chainer_maskrcnn._hip.nn import workspace
    g = torch.Generator(device='cpu').manual_seed(P + C)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    xa, xb = mk(P, C) * 2 + 0.5, mk(P, C) * 0.7 - 1.0
    ga, ba, gb, bb = mk(C) * 0.5 + 1, mk(C) * 0.1, mk(C) * 0.5 + 1, mk(C) * 0.1
    rm = [torch.zeros(C, device=DEV) for _ in range(4)]
    rv = [torch.ones(C, device=DEV) for _ in range(4)]
    # reference: shortcut first (no ReLU), then the main branch with the residual
    r, mb_ref, sb_ref = ops.bn_train_fwd(xb, gb, bb, relu=False, running_mean=rm[0], running_var=rv[0])
    y_ref, ma_ref, sa_ref = ops.bn_train_fwd(xa, ga, ba, residual=r, relu=True, running_mean=rm[1], running_var=rv[1])
    y = torch.empty_like(xa)
    ma, sa, mb, sb = (torch.empty(C, device=DEV) for _ in range(4))
    ws = workspace(max(lib.mrcnn_bn_workspace_bytes(P, C), lib.mrcnn_bn_pair_workspace_bytes(P, C)), xa.device)
    check(lib.mrcnn_bn_train_fwd_pair_f32(ptr(xa), None, 0, ptr(ga), ptr(ba), ptr(ma), ptr(sa), ptr(rm[3]), ptr(rv[3]), ptr(xb), None, 0, ptr(gb), ptr(bb),
                                          ptr(mb), ptr(sb), ptr(rm[2]), ptr(rv[2]), ptr(y), P, C, 2e-5, 0.9, ptr(ws), ws.numel(), sp()))
    for got, want in ((y, y_ref), (ma, ma_ref), (sa, sa_ref), (mb, mb_ref), (sb, sb_ref), (rm[2], rm[0]), (rv[2], rv[0]), (rm[3], rm[1]), (rv[3], rv[1])):
        assert torch.equal(got, want)
    gy = mk(P, C) * 1e-3
    if masked:
        gy = torch.where(y_ref > 0, gy, torch.zeros_like(gy)).contiguous()
        gxa_ref, _, gga_ref, gba_ref = ops.bn_train_bwd(gy, xa, None, ga, ma_ref, sa_ref, relu=False)
        gr = gy
    else:
        gxa_ref, gr, gga_ref, gba_ref = ops.bn_train_bwd(gy, xa, y_ref, ga, ma_ref, sa_ref, relu=True, want_gres=True)
    gxb_ref, _, ggb_ref, gbb_ref = ops.bn_train_bwd(gr, xb, None, gb, mb_ref, sb_ref, relu=False)
    gxa, gxb = torch.empty_like(xa), torch.empty_like(xb)
    gga, gba, ggb, gbb = (torch.empty(C, device=DEV) for _ in range(4))
    check(lib.mrcnn_bn_train_bwd_pair_f32(ptr(gy), None if masked else ptr(y_ref), ptr(xa), ptr(xb), ptr(ga), ptr(ma_ref), ptr(sa_ref), ptr(gb), ptr(mb_ref),
                                          ptr(sb_ref), ptr(gxa), ptr(gxb), ptr(gga), ptr(gba), ptr(ggb), ptr(gbb), P, C, ptr(ws), ws.numel(), sp()))
    for got, want in ((gxa, gxa_ref), (gxb, gxb_ref), (gga, gga_ref), (gba, gba_ref), (ggb, ggb_ref), (gbb, gbb_ref)):
        assert torch.equal(got, want)
    assert torch.isfinite(gxa).all() and torch.isfinite(gxb).all()


@pytest.mark.parametrize('geom', [(2, 32, 32, 256, 256), (2, 64, 64, 64, 64), (1, 47, 61, 128, 128)], ids=lambda g_: 'x'.join(map(str, g_)))
def test_convolution_with_batchnorm_relu_on_load_equals_the_materialised_sequence(geom):
    """(ABI v10) conv -> BatchNorm -> ReLU -> conv 3x3 with the normalised activation never written: mrcnn_bn_train_stats_f32 +
    mrcnn_conv2d_fwd_inbn_f32 / mrcnn_conv2d_bwd_filter_inbn_f32 (the Winograd input transforms apply BatchNorm + ReLU to every tap they
    load; taps outside the image stay zero) against BatchNorm apply + the ordinary calls on the written activation: the same BITS in the
    output, the statistics partials of the output and the filter gradient - in the shipped arithmetic and in float32, where the geometry
    takes the Winograd path in both passes (mrcnn_conv2d_inbn_ok), refused elsewhere."""
    from chainer_maskrcnn import _hip
    from chainer_maskrcnn._hip import nn as hnn
    lib, check, ptr, sp = _hip.lib(), _hip.check, _hip.ptr, _hip.stream_ptr
    N, H, W, Ci, Co = geom
    g = torch.Generator(device='cpu').manual_seed(H * W + Ci)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    h1 = mk(N, H, W, Ci) * 1.5 + 0.3
    gam, bet = mk(Ci) * 0.3 + 1, mk(Ci) * 0.2
    w = mk(Co, 3, 3, Ci) * 0.03
    gy = mk(N, H, W, Co) * 1e-3
    keep = hnn.split_operands()
    try:
        for mode in ((0, 0, 0), (3, 3, 3)):
            check(lib.mrcnn_conv2d_set_split_operands(*mode))
            ok = lib.mrcnn_conv2d_inbn_ok(N, H, W, Ci, Co, 3, 3, 1, 1)
            a1, mean, invstd = ops.bn_train_fwd(h1, gam, bet, relu=True)
            P = N * H * W
            m2, s2 = torch.empty(Ci, device=DEV), torch.empty(Ci, device=DEV)
            ws_bn = hnn.workspace(lib.mrcnn_bn_workspace_bytes(P, Ci), h1.device)
            check(lib.mrcnn_bn_train_stats_f32(ptr(h1), None, 0, ptr(m2), ptr(s2), None, None, P, Ci, 2e-5, 0.9, ptr(ws_bn), ws_bn.numel(), sp()))
            assert torch.equal(m2, mean) and torch.equal(s2, invstd)
            nb = lib.mrcnn_conv2d_workspace_bytes(N, H, W, Ci, Co, 3, 3, 1, 1)
            ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=DEV)
            y = torch.empty((N, H, W, Co), device=DEV)
            rows = lib.mrcnn_conv2d_bnstats_rows(N, H, W, Ci, Co, 3, 3, 1, 1)
            if not ok:
                rc = lib.mrcnn_conv2d_fwd_inbn_f32(ptr(h1), ptr(gam), ptr(bet), ptr(mean), ptr(invstd), ptr(w), ptr(y), N, H, W, Ci, Co, 3, 3, 1, 1, None, None,
                                                   ptr(ws), ws.numel(), sp())
                assert rc != 0 and b'inbn_ok' in lib.mrcnn_last_error()
                continue
            part = torch.empty((rows, 2, Co), device=DEV) if rows else None
            check(lib.mrcnn_conv2d_fwd_inbn_f32(ptr(h1), ptr(gam), ptr(bet), ptr(mean), ptr(invstd), ptr(w), ptr(y), N, H, W, Ci, Co, 3, 3, 1, 1, ptr(part), None,
                                                ptr(ws), ws.numel(), sp()))
            if rows:
                y_ref, _, part_ref = hnn.conv2d_fwd_bnstats_raw(a1, w, 1, 1)
                assert torch.equal(part, part_ref)
            else:
                y_ref = hnn.conv2d_fwd_raw(a1, w, None, 1, 1, False)
            assert torch.equal(y, y_ref) and torch.isfinite(y).all()
            gw_ref, _ = hnn.conv2d_bwd_filter_raw(a1, gy, tuple(w.shape), 1, 1, False)
            gw = torch.empty_like(w)
            nbf = lib.mrcnn_conv2d_bwd_filter_workspace_bytes(N, H, W, Ci, Co, 3, 3, 1, 1)
            wsf = torch.empty((max(nbf, 1),), dtype=torch.uint8, device=DEV)
            check(lib.mrcnn_conv2d_bwd_filter_inbn_f32(ptr(h1), ptr(gam), ptr(bet), ptr(mean), ptr(invstd), ptr(gy), ptr(gw), N, H, W, Ci, Co, 3, 3, 1, 1, 0, None,
                                                       ptr(wsf), wsf.numel(), sp()))
            assert torch.equal(gw, gw_ref)
    finally:
        check(lib.mrcnn_conv2d_set_split_operands(*keep))
