"""GPU tests: MFMA implicit-GEMM convolution against a plain PyTorch fp32 CPU reference
(floating-point kernel => torch reference, tolerance 1e-3 relative per BASELINE.json north_star;
observed error is ~1e-6 because the f32 MFMA is an exact fmaf chain)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import nn as hnn  # noqa: E402

DEV = 'cuda:0'


def _ref_conv(x, w, b, stride, pad):
    # x NHWC, w OHWI -> torch NCHW / OIHW on CPU in float64 for a tight reference
    y = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(),
                 None if b is None else b.double(), stride=stride, padding=pad)
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [  # N, H, W, Cin, Cout, K, stride, pad
    (2, 9, 11, 32, 32, 1, 1, 0),
    (1, 16, 20, 64, 160, 3, 1, 1),
    (2, 13, 9, 96, 64, 3, 1, 1),
    (1, 12, 12, 64, 32, 1, 2, 0),
    (3, 1, 1, 256, 96, 1, 1, 0),      # linear layer shape
    (1, 40, 36, 32, 288, 3, 1, 1),    # several M and N tiles
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('relu', [False, True])
def test_conv_forward(case, relu):
    N, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, K, K, Cin), generator=g) / (K * K * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g)
    ref = _ref_conv(x, w, b, s, p)
    if relu:
        ref = ref.clamp_min(0)
    got = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), s, p, relu).cpu().double()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    assert err < 1e-5, err


@pytest.mark.parametrize('case', [c for c in CASES if c[6] == 1])
def test_conv_backward_data_and_filter(case):
    N, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(100 + sum(case))
    x = torch.randn((N, H, W, Cin), generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn((Cout, K, K, Cin), generator=g, dtype=torch.float64) / (K * K * Cin) ** 0.5).requires_grad_(True)
    b = torch.randn((Cout,), generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), b, stride=s, padding=p).permute(0, 2, 3, 1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    gyd = gy.float().contiguous().to(DEV)
    gx = hnn.conv2d_bwd_data_raw(gyd, w.detach().float().to(DEV), tuple(x.shape), s, p).cpu().double()
    gw, gb = hnn.conv2d_bwd_filter_raw(x.detach().float().to(DEV), gyd, tuple(w.shape), s, p, True)
    for name, got, ref in (('gx', gx, x.grad), ('gw', gw.cpu().double(), w.grad), ('gb', gb.cpu().double(), b.grad)):
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        assert err < 2e-5, (name, err)


def test_conv_backward_filter_stride2_and_determinism():
    N, H, W, Cin, Cout, K, s, p = 1, 24, 24, 64, 32, 1, 2, 0
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, H, W, Cin), generator=g, dtype=torch.float64)
    w = torch.randn((Cout, K, K, Cin), generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride=s, padding=p).permute(0, 2, 3, 1)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xd, gyd = x.float().to(DEV), gy.float().contiguous().to(DEV)
    gw1, _ = hnn.conv2d_bwd_filter_raw(xd, gyd, tuple(w.shape), s, p, False)
    gw2, _ = hnn.conv2d_bwd_filter_raw(xd, gyd, tuple(w.shape), s, p, False)
    assert torch.equal(gw1, gw2)                     # deterministic split-K
    err = (gw1.cpu().double() - w.grad).abs().max().item() / w.grad.abs().max().item()
    assert err < 2e-5, err


def test_conv_rejects_unpadded_channels():
    from chainer_maskrcnn._hip import MrcnnHipError
    with pytest.raises(MrcnnHipError):
        hnn.conv2d_fwd_raw(torch.zeros((1, 4, 4, 3), device=DEV), torch.zeros((32, 3, 3, 3), device=DEV), None, 1, 1, False)


# Grids a little larger than a whole number of rounds of workgroup slots take the tail-split path (the last tiles are
# computed by several workgroups along K and summed from slabs).  The slot count is 512 or 768 depending on the
# kernel's occupancy, so two pixel counts are used: 800 and 1050 M tiles of 128 rows.
@pytest.mark.parametrize('hw', [(320, 320), (420, 320)])
@pytest.mark.parametrize('k,cin,cout', [(1, 256, 128), (3, 32, 160)])
def test_conv_tail_split_forward_and_backward_data(hw, k, cin, cout):
    H, W = hw
    pad = k // 2
    g = torch.Generator().manual_seed(H + k)
    x = torch.randn((1, H, W, cin), generator=g)
    w = torch.randn((cout, k, k, cin), generator=g) / (k * k * cin) ** 0.5
    b = torch.randn((cout,), generator=g)
    ref = _ref_conv(x, w, b, 1, pad).clamp_min(0)
    got = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, pad, True)
    got2 = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, pad, True)
    assert torch.equal(got, got2)
    err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err
    # backward-data with accumulation into an existing gradient
    gy = torch.randn((1, H, W, cout), generator=g)
    base = torch.randn((1, H, W, cin), generator=g)
    refx = F.conv_transpose2d(gy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, stride=1,
                              padding=pad).permute(0, 2, 3, 1) + base.double()
    out = base.to(DEV).clone()
    hnn.conv2d_bwd_data_raw(gy.to(DEV), w.to(DEV), (1, H, W, cin), 1, pad, out=out)
    err = (out.cpu().double() - refx).abs().max().item() / refx.abs().max().item()
    assert err < 1e-5, err


@pytest.mark.parametrize('case', [(1, 16, 20, 64, 160, 3, 1, 1), (2, 64, 64, 256, 64, 1, 1, 0), (1, 320, 320, 32, 160, 3, 1, 1)])
def test_conv_backward_data_fused_relu_mask(case):
    """relu_x: gx is zeroed where the layer's input (a ReLU output) is <= 0 - on the plain, split-K and tail-split paths."""
    N, H, W, Cin, Cout, K, s, p = case
    g = torch.Generator().manual_seed(7 + sum(case))
    x = torch.randn((N, H, W, Cin), generator=g).clamp_min(0).to(DEV)
    w = (torch.randn((Cout, K, K, Cin), generator=g) / (K * K * Cin) ** 0.5).to(DEV)
    gy = torch.randn((N, H, W, Cout), generator=g).to(DEV)
    plain = hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), s, p)
    fused = hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), s, p, relu_x=x)
    assert torch.equal(fused, torch.where(x > 0, plain, torch.zeros_like(plain)))


# 3x3 / stride 1 / pad 1 layers with >= 2048 pixels and >= 256 channels take the Winograd F(2x2,3x3) path (filter,
# input and output transforms + one batched 1x1 GEMM launch).  Tolerance 3e-5 of the tensor scale: the transforms add a
# few ulp to the direct kernel's error; odd sizes exercise the clipped edge tiles and the padded GEMM rows.
# Per-pass tiles (mrcnn_conv2d_set_winograd_pass_tiles): (0,0,0) = every pass follows the global tile (both tiles are covered
# in all three passes); 'mixed' = the shipped default, forward F(2x2) + backward F(4x4): the forward's transformed input
# cannot be reused by the filter-gradient pass then (different tile) and the library must notice.
@pytest.mark.parametrize('tile,tol,pass_tiles', [(2, 3e-5, (0, 0, 0)), (4, 3e-4, (0, 0, 0)), (4, 3e-4, (2, 0, 0)), (2, 3e-4, (4, -1, 4))],
                         ids=['f2', 'f4', 'shipped', 'fwd4_bwd_direct_wgrad4'])
@pytest.mark.parametrize('case', [(2, 48, 48, 256, 256), (1, 67, 63, 256, 288), (3, 40, 36, 320, 256), (16, 14, 14, 256, 256)])
def test_conv_winograd_forward_and_backward_data(case, tile, tol, pass_tiles):
    from chainer_maskrcnn import _hip
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, tile))
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(*pass_tiles))
    try:
        _winograd_case(case, tol, shared_gy=(pass_tiles[1] == pass_tiles[2]))    # same tile in both backward passes
    finally:
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, 0))
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(2, 0, 0))


@pytest.mark.parametrize('case', [(1, 67, 63, 256, 288), (3, 40, 36, 64, 64), (2, 70, 132, 128, 128)])
def test_winograd_tile_order_is_speed_only(case):
    """The input transforms walk the tiles in column panels (default 16 tiles wide); every order - launch order, raster, panels that
    do and do not divide the tile row - writes the same rows of V, so all three passes give the same bits."""
    from chainer_maskrcnn import _hip
    lib = _hip.lib()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(5 + sum(case))
    x = torch.randn((N, H, W, Cin), generator=g).to(DEV)
    w = (torch.randn((Cout, 3, 3, Cin), generator=g) / (9 * Cin) ** 0.5).to(DEV)
    b = torch.randn((Cout,), generator=g).to(DEV)
    gy = torch.randn((N, H, W, Cout), generator=g).to(DEV)
    _hip.check(lib.mrcnn_conv2d_set_winograd_thresholds(64, 2048, 0))
    _hip.check(lib.mrcnn_conv2d_set_winograd_pass_tiles(0, 0, 0))
    try:
        outs = []
        for order in (16, 0, 1, 2, 5, 64):
            _hip.check(lib.mrcnn_debug_wino_banded(order))
            got = (hnn.conv2d_fwd_raw(x, w, b, 1, 1, True), hnn.conv2d_bwd_data_raw(gy, w, (N, H, W, Cin), 1, 1),
                   hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, 1, True))
            outs.append([o[0] if isinstance(o, (tuple, list)) else o for o in got])
        for o in outs[1:]:
            for a, r in zip(o, outs[0]):
                assert torch.equal(a, r)
    finally:
        _hip.check(lib.mrcnn_debug_wino_banded(16))
        _hip.check(lib.mrcnn_conv2d_set_winograd_thresholds(256, 2048, 0))
        _hip.check(lib.mrcnn_conv2d_set_winograd_pass_tiles(2, 0, 0))


def _winograd_case(case, tol, shared_gy=True):
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(31 + sum(case))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, 3, 3, Cin), generator=g) / (9 * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g)
    ref = _ref_conv(x, w, b, 1, 1)
    got = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, 1, False)
    assert torch.equal(got, hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, 1, False))       # reproducible
    err = (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    got = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, 1, True)
    err = (got.cpu().double() - ref.clamp_min(0)).abs().max().item() / ref.abs().max().item()
    assert err < tol, err
    # backward-data: plain, accumulating, and with the fused ReLU mask
    gy = torch.randn((N, H, W, Cout), generator=g)
    refx = F.conv_transpose2d(gy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, stride=1,
                              padding=1).permute(0, 2, 3, 1)
    gx = hnn.conv2d_bwd_data_raw(gy.to(DEV), w.to(DEV), (N, H, W, Cin), 1, 1)
    sc = refx.abs().max().item()
    assert (gx.cpu().double() - refx).abs().max().item() / sc < tol
    base = torch.randn((N, H, W, Cin), generator=g)
    out = base.to(DEV).clone()
    hnn.conv2d_bwd_data_raw(gy.to(DEV), w.to(DEV), (N, H, W, Cin), 1, 1, out=out)
    assert (out.cpu().double() - (refx + base.double())).abs().max().item() / sc < tol
    xr = torch.randn((N, H, W, Cin), generator=g).clamp_min(0).to(DEV)
    fused = hnn.conv2d_bwd_data_raw(gy.to(DEV), w.to(DEV), (N, H, W, Cin), 1, 1, relu_x=xr)
    assert torch.equal(fused, torch.where(xr > 0, gx, torch.zeros_like(gx)))
    # filter gradient (Winograd F(3x3, 2x2)): overwrite, accumulate, bias gradient, reproducibility
    xd = x.double().requires_grad_(False)
    wref = torch.autograd.functional.vjp(
        lambda ww: F.conv2d(xd.permute(0, 3, 1, 2), ww.permute(0, 3, 1, 2), None, stride=1, padding=1).permute(0, 2, 3, 1),
        w.double(), gy.double())[1]
    gbref = gy.double().sum((0, 1, 2))
    gw, gb = hnn.conv2d_bwd_filter_raw(x.to(DEV), gy.to(DEV), tuple(w.shape), 1, 1, True)
    gw2, _ = hnn.conv2d_bwd_filter_raw(x.to(DEV), gy.to(DEV), tuple(w.shape), 1, 1, True)
    assert torch.equal(gw, gw2)
    _, v = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), b.to(DEV), 1, 1, False, keep_v=True)      # transformed input kept by forward
    assert v is not None
    gw3, _ = hnn.conv2d_bwd_filter_raw(x.to(DEV), gy.to(DEV), tuple(w.shape), 1, 1, True, wino_v=v)
    assert torch.equal(gw, gw3)
    if shared_gy:   # one read of gy: the data-gradient call also emits the filter-gradient operand and the bias gradient
        gb4 = torch.zeros((Cout,), device=DEV)
        gx4, wt = hnn.conv2d_bwd_data_raw(gy.to(DEV), w.to(DEV), (N, H, W, Cin), 1, 1, emit_w=True, gb=gb4)
        assert torch.equal(gx4, gx)
        gw4, _ = hnn.conv2d_bwd_filter_raw(x.to(DEV), gy.to(DEV), tuple(w.shape), 1, 1, False, wino_v=v, wino_w=wt)
        assert torch.equal(gw, gw4)
        assert (gb4.cpu().double() - gbref).abs().max().item() / gbref.abs().max().item() < 2e-5
    else:           # the backward passes run other tiles than the forward: no shared transform is offered
        assert hnn.winograd_w_bytes((N, H, W, Cin), tuple(w.shape), 1, 1) == 0
    errw = (gw.cpu().double() - wref).abs().max().item() / wref.abs().max().item()
    print('winograd errors: fwd %.2e filter-grad %.2e' % (err, errw))
    assert errw < tol, errw
    assert (gb.cpu().double() - gbref).abs().max().item() / gbref.abs().max().item() < 2e-5
    acc_w, acc_b = (torch.ones_like(gw), torch.ones_like(gb))
    hnn.conv2d_bwd_filter_raw(x.to(DEV), gy.to(DEV), tuple(w.shape), 1, 1, True, gw=acc_w, gb=acc_b, accumulate=True)
    assert (acc_w.cpu().double() - (wref + 1)).abs().max().item() / wref.abs().max().item() < tol


# ---- EXPLORATORY opt-in: three-term split operands on the 16-bit MFMA (mrcnn_conv2d_set_split_operands) -----------------------
SPLIT_CASES = [  # N, H, W, Cin, Cout, K, pad
    (2, 9, 11, 32, 32, 1, 0), (1, 16, 20, 64, 160, 3, 1), (2, 13, 9, 96, 64, 3, 1), (3, 1, 1, 256, 96, 1, 0),
    (1, 40, 36, 32, 288, 3, 1),       # several M and N tiles, ragged M
    (2, 8, 8, 2048, 512, 1, 0),       # few tiles, long K: the forward / backward-data split-K launches
    (16, 14, 14, 256, 256, 3, 1),     # Winograd in all three passes
    (1, 67, 63, 256, 288, 3, 1),      # Winograd, ragged tiles, Cout not a tile multiple
]


@pytest.mark.parametrize('planes', [1, 2, 3, 11], ids=['bf16', 'half', 'bf16x6', 'bf16x6_plain_loop'])
@pytest.mark.parametrize('case', SPLIT_CASES)
def test_conv_split_operands_all_three_passes(case, planes):
    """Every GEMM of the three passes with hi + lo 16-bit planes per float32 operand and float32 accumulation of
    al*bh + ah*bl + ah*bh: against float64, bf16 planes within 3e-5 of the tensor scale (float32 MFMA: 5e-7), half planes within
    3e-6 - times 30 on the F(4x4) Winograd layers, whose transforms amplify every GEMM's error alike.  K- and row-contiguous
    operand layouts (the latter through ds_read_b64_tr_b16), split-K / tail-split launches, accumulate and ReLU-mask epilogues.
    planes = 3: hi + mid + lo bf16 planes (= the float32 operand exactly) and six products - held to the FLOAT32 kernels' own bar."""
    from chainer_maskrcnn import _hip
    piped, planes = planes == 11, (3 if planes == 11 else planes)      # 11: bf16x6 with the plain K loop (mrcnn_debug_conv_parts(8)) instead of the pipelined default
    N, H, W, Cin, Cout, K, p = case
    g = torch.Generator().manual_seed(900 + sum(case))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, K, K, Cin), generator=g) / (K * K * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g)
    gy = torch.randn((N, H, W, Cout), generator=g)
    base = torch.randn((N, H, W, Cin), generator=g)
    ref_y = _ref_conv(x, w, b, 1, p)
    ref_gx = F.conv_transpose2d(gy.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, stride=1, padding=p).permute(0, 2, 3, 1)
    ref_gw = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Cout, Cin, K, K), gy.double().permute(0, 3, 1, 2), 1, p).permute(0, 2, 3, 1)
    wino = K == 3 and Cin >= 256
    tol = (3e-5 if planes == 1 else 3e-6) * (30 if wino else 1)
    if planes == 3:
        tol = 3e-4 if wino else 2e-6              # the bar of the float32-MFMA default (last line of this test)
    rel = lambda got, ref: (got.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    xd, wd, bd, gyd = x.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(0, 0, 0))
    _hip.check(_hip.lib().mrcnn_conv2d_set_split_operands(planes, planes, planes))
    _hip.check(_hip.lib().mrcnn_debug_conv_parts(8 if piped else 0))
    try:
        y = hnn.conv2d_fwd_raw(xd, wd, bd, 1, p, False)
        assert torch.equal(y, hnn.conv2d_fwd_raw(xd, wd, bd, 1, p, False))                 # reproducible
        assert rel(y, ref_y) < tol, ('fwd', rel(y, ref_y))
        assert rel(hnn.conv2d_fwd_raw(xd, wd, bd, 1, p, True), ref_y.clamp_min(0)) < tol
        gx = hnn.conv2d_bwd_data_raw(gyd, wd, tuple(x.shape), 1, p)
        assert rel(gx, ref_gx) < tol, ('bwd_data', rel(gx, ref_gx))
        out = base.to(DEV).clone()
        hnn.conv2d_bwd_data_raw(gyd, wd, tuple(x.shape), 1, p, out=out)
        assert (out.cpu().double() - (ref_gx + base.double())).abs().max().item() / ref_gx.abs().max().item() < tol
        xr = torch.randn((N, H, W, Cin), generator=g).clamp_min(0).to(DEV)
        assert torch.equal(hnn.conv2d_bwd_data_raw(gyd, wd, tuple(x.shape), 1, p, relu_x=xr), torch.where(xr > 0, gx, torch.zeros_like(gx)))
        gw, gb = hnn.conv2d_bwd_filter_raw(xd, gyd, tuple(w.shape), 1, p, True)
        assert rel(gw, ref_gw) < tol, ('bwd_filter', rel(gw, ref_gw))
        assert rel(gb, gy.double().sum((0, 1, 2))) < 2e-5
    finally:
        _hip.check(_hip.lib().mrcnn_conv2d_set_split_operands(0, 0, 0))
        _hip.check(_hip.lib().mrcnn_debug_conv_parts(0))
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(2, 0, 0))
    # and the default is untouched: float32 MFMA again
    assert rel(hnn.conv2d_fwd_raw(xd, wd, bd, 1, p, False), ref_y) < (3e-4 if wino else 2e-6)


def test_shared_gy_transform_with_the_emulated_arithmetic():
    """nn/core.py WINOGRAD_SHARED_GY_TRANSFORM (opt-in: one read of gy for the data gradient, the filter-gradient operand and the
    bias gradient) together with split mode 3 on a layer large enough for the plane-GEMM plan: the filter-gradient call then arrives with a
    cached float32 W and takes k_conv_igemm with ITS OWN split-K plan (WinoFLayout ks32 / kc32), not the plane GEMM's.  Gradients against
    the default path (separate transforms) of the same arithmetic."""
    from chainer_maskrcnn.nn import core
    from chainer_maskrcnn import _hip
    lib = _hip.lib()
    ps = core.ParamStore()
    conv = core.Conv(ps, 'c', 256, 256, 3, 1, 1, bias=True)
    ps.materialise(torch.device(DEV), seed=3)
    g = torch.Generator(device='cpu').manual_seed(8)
    x = torch.randn((2, 96, 96, 256), generator=g).to(DEV)
    gy = (torch.randn((2, 96, 96, 256), generator=g) * 1e-2).to(DEV)
    out = []
    keep = hnn.winograd_pass_tiles()
    try:
        hnn.set_winograd_pass_tiles(2, 0, 0)
        _hip.check(lib.mrcnn_conv2d_set_split_operands(3, 3, 3))
        for shared in (False, True):
            core.WINOGRAD_SHARED_GY_TRANSFORM = shared
            ps.grads.zero_()
            y, ctx = conv.fwd(x)
            gx = conv.bwd(ctx, gy.clone())
            core.join_side_stream(torch.device(DEV))
            torch.cuda.synchronize()
            out.append((gx.clone(), ps.g('c/W').clone(), ps.g('c/b').clone()))
    finally:
        core.WINOGRAD_SHARED_GY_TRANSFORM = False
        _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
        hnn.set_winograd_pass_tiles(*keep)
    for a, b in zip(*out):
        assert torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max())
    assert float(out[0][1].abs().max()) > 0
