"""GPU parity tests: HIP ROIAlign (through the C ABI) against the NumPy oracle.

Tolerances: sample indices AND weights bit-exact (integer / same-float-op contract);
forward values bit-exact on the NHWC and generic kernels (same operation order, no FMA);
backward values within rtol 1e-5 / atol 1e-5*max|gy| (summation order differs: the tile
kernel pre-sums weights per bin).
"""
import numpy as np
import pytest
import torch

from oracle import roi_align as ora
from tests.util import config2_inputs, rand_rois_xy

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.functions.roi_align.roi_align_2d import roi_align_2d, roi_align_sample_tables  # noqa: E402
from chainer_maskrcnn.functions.roi_align_2d_yx import _roi_align_2d_yx  # noqa: E402
from chainer_maskrcnn import _hip  # noqa: E402

DEV = 'cuda:0'


def _edge_rois(N, H, W, scale):
    s = 1.0 / scale
    return np.array([
        [0, -30 * s, -30 * s, -20 * s, -20 * s],            # fully outside (void samples)
        [N - 1, 4 * s, 4 * s, 4 * s, 4 * s],                 # zero area
        [0, 0, 0, W * s, H * s],                             # whole map
        [0, (W - 1) * s, (H - 1) * s, (W + 3) * s, (H + 3) * s],   # last row/col and beyond
        [N - 1, -0.9 * s, -0.9 * s, 1.1 * s, 1.3 * s],       # straddles the -1 validity edge
        [0, 3.5 * s, 2.25 * s, 3.5 * s + 0.5, 2.25 * s + 0.5],     # sub-cell RoI (clamped to 1 cell)
    ], np.float32)


@pytest.mark.parametrize('sr', [1, 2, 0])
@pytest.mark.parametrize('P', [7, 14])
def test_sample_indices_and_weights_bit_exact(sr, P):
    rs = np.random.RandomState(10 + sr)
    H, W, scale = 50, 68, 0.0625
    rois = np.concatenate([_edge_rois(2, H, W, scale), rand_rois_xy(rs, 300, 2, H, W, scale)], 0)
    smax = 64 if sr else 256
    cnt, idx, wgt = ora.roi_align_sample_tables(rois, H, W, P, P, scale, sr, smax)
    dcnt, didx, dwgt = roi_align_sample_tables(torch.from_numpy(rois).to(DEV), H, W, P, P, scale, sr, smax)
    np.testing.assert_array_equal(dcnt.cpu().numpy(), cnt)
    np.testing.assert_array_equal(didx.cpu().numpy(), idx)
    np.testing.assert_array_equal(dwgt.cpu().numpy().view(np.uint32), wgt.view(np.uint32))


@pytest.mark.parametrize('layout', ['nhwc', 'nchw'])
@pytest.mark.parametrize('C,P,sr', [(8, 7, 2), (256, 7, 2), (12, 14, 2), (8, 3, 1), (8, 5, 0), (6, 7, 2)])
def test_forward_matches_oracle(layout, C, P, sr):
    rs = np.random.RandomState(C + P)
    N, H, W, scale = 2, 21, 30, 0.125
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.concatenate([_edge_rois(N, H, W, scale), rand_rois_xy(rs, 40, N, H, W, scale)], 0)
    want = ora.roi_align_fwd(x, rois, P, P, scale, sr)
    xt = torch.from_numpy(x).to(DEV)
    if layout == 'nhwc':
        xt = xt.contiguous(memory_format=torch.channels_last)
    got = roi_align_2d(xt, torch.from_numpy(rois).to(DEV), P, P, scale, sr)
    assert got.shape == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)      # bit-exact


@pytest.fixture(params=[2, 1], ids=['waves', 'tiles'])
def bwd_variant(request):
    """The two fast backward kernels: 2 = independent waves that derive the geometry themselves (default), 1 = barrier-synchronised
    8x8 tiles (the fallback for tensors beyond the 32-bit buffer offsets of variant 2)."""
    _hip.check(_hip.lib().mrcnn_roi_align_set_bwd_variant(request.param))
    yield request.param
    _hip.check(_hip.lib().mrcnn_roi_align_set_bwd_variant(2))


def _bwd_ws(gyt, N, C, H, W, rois_t, R, P, scale, sr, gx):
    """mrcnn_roi_align_bwd_ws_f32 with the workspace its query asks for."""
    nb = _hip.lib().mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, P, P, sr)
    ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=DEV)
    _hip.check(_hip.lib().mrcnn_roi_align_bwd_ws_f32(_hip.ptr(gyt), 1, N, C, H, W, _hip.ptr(rois_t), R, P, P, scale, sr, _hip.ptr(gx),
                                                     _hip.ptr(ws), nb, _hip.stream_ptr()))


@pytest.mark.parametrize('layout', ['nhwc', 'nchw'])
@pytest.mark.parametrize('C,P,sr', [(8, 7, 2), (256, 7, 2), (12, 14, 2), (8, 3, 1), (8, 5, 0), (6, 7, 2),
                                    (260, 7, 2)])
def test_backward_matches_oracle(layout, C, P, sr, bwd_variant):
    rs = np.random.RandomState(100 + C + P)
    N, H, W, scale = 2, 21, 30, 0.125
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.concatenate([_edge_rois(N, H, W, scale), rand_rois_xy(rs, 40, N, H, W, scale)], 0)
    gy = rs.standard_normal((rois.shape[0], C, P, P)).astype(np.float32)
    want = ora.roi_align_bwd(gy, rois, x.shape, scale, sr)
    xt = torch.from_numpy(x).to(DEV)
    if layout == 'nhwc':
        xt = xt.contiguous(memory_format=torch.channels_last)
    xt.requires_grad_(True)
    y = roi_align_2d(xt, torch.from_numpy(rois).to(DEV), P, P, scale, sr)
    y.backward(torch.from_numpy(gy).to(DEV))
    got = xt.grad.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * np.abs(gy).max() * 4)


def test_backward_many_rois_on_one_tile_and_segments(bwd_variant):
    """Hundreds of RoIs on one tile => queue overflow drains / multi-round read-modify-write; R = 1500 => three 512-RoI segment
    tables (variant 2) / two 1024-RoI segments (variant 1)."""
    rs = np.random.RandomState(7)
    N, C, H, W, scale, P = 1, 8, 16, 16, 0.25, 7
    R = 1500
    c = rs.uniform(10, 30, (R, 2))
    hw = rs.uniform(2, 20, (R, 2))
    rois = np.stack([np.zeros(R), c[:, 1] - hw[:, 1], c[:, 0] - hw[:, 0], c[:, 1] + hw[:, 1],
                     c[:, 0] + hw[:, 0]], 1).astype(np.float32)
    gy = rs.standard_normal((R, C, P, P)).astype(np.float32)
    want = ora.roi_align_bwd(gy, rois, (N, C, H, W), scale, 2)
    gx = torch.empty((N, C, H, W), device=DEV).contiguous(memory_format=torch.channels_last)
    gyt = torch.from_numpy(gy).to(DEV).contiguous(memory_format=torch.channels_last)
    _bwd_ws(gyt, N, C, H, W, torch.from_numpy(rois).to(DEV), R, P, scale, 2, gx)
    np.testing.assert_allclose(gx.cpu().numpy(), want, rtol=2e-5, atol=1e-4 * np.abs(want).max())


def test_empty_and_error_paths():
    x = torch.zeros((1, 8, 5, 5), device=DEV).contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y = roi_align_2d(x, torch.zeros((0, 5), device=DEV), 7, 7, 0.25)
    assert y.shape == (0, 8, 7, 7)
    y.sum().backward()
    assert torch.all(x.grad == 0)
    with pytest.raises(_hip.MrcnnHipError):
        roi_align_2d(torch.zeros((1, 8, 5, 5)), torch.zeros((1, 5)), 7, 7, 0.25)   # CPU tensor: no fallback
    with pytest.raises(_hip.MrcnnHipError):
        _hip.check(_hip.lib().mrcnn_roi_align_fwd_f32(None, 1, 1, 8, 5, 5, None, 1, 7, 7, 0.25, 2, None, None))
    with pytest.raises(ValueError):
        roi_align_2d(x, torch.zeros((3, 4), device=DEV), 7, 7, 0.25)


def test_yx_shim_column_order():
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1, 4, 12, 12)).astype(np.float32)
    yx = np.array([[0, 3, 5, 30, 41], [0, 0, 0, 47, 20]], np.float32)     # (idx,y1,x1,y2,x2)
    want = ora.roi_align_2d_yx(x, yx, 7, 7, 0.25)
    got = _roi_align_2d_yx(torch.from_numpy(x).to(DEV), torch.from_numpy(yx).to(DEV), 7, 7, 0.25)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_config2_full_size_properties_and_sampled_parity():
    """BASELINE configs[1] at full size: adjoint identity <y,gy> == <gx,x>, linearity, run-to-run
    bit reproducibility of the tile backward, and oracle parity on a 48-RoI subset."""
    x, yx, gy = config2_inputs()
    xy = yx[:, [0, 2, 1, 4, 3]]
    xt = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rt = torch.from_numpy(xy).to(DEV)
    gyt = torch.from_numpy(gy).to(DEV)
    y = roi_align_2d(xt, rt, 7, 7, 0.25)
    y.backward(gyt)
    gx = xt.grad.clone()
    lhs = torch.sum(y.double() * gyt.double()).item()
    rhs = torch.sum(gx.double() * xt.detach().double()).item()
    assert abs(lhs - rhs) <= 1e-6 * abs(lhs) + 1e-3
    xt.grad = None
    y2 = roi_align_2d(xt, rt, 7, 7, 0.25)
    y2.backward(2.0 * gyt)
    assert torch.equal(xt.grad, 2.0 * gx)                      # linear, and bit-reproducible
    sub = np.arange(0, 512, 11)[:48]
    want_y = ora.roi_align_fwd(x, xy[sub], 7, 7, 0.25, 2)
    np.testing.assert_array_equal(y.detach()[torch.from_numpy(sub).to(DEV)].cpu().numpy(), want_y)
    gsub = torch.zeros_like(gyt)
    gsub[torch.from_numpy(sub).to(DEV)] = gyt[torch.from_numpy(sub).to(DEV)]
    xt.grad = None
    roi_align_2d(xt, rt, 7, 7, 0.25).backward(gsub)
    want_gx = ora.roi_align_bwd(gy[sub], xy[sub], x.shape, 0.25, 2)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), want_gx, rtol=1e-5, atol=2e-5 * np.abs(gy).max())


_config2_oracle = {}


def _config2_want(sr):
    """The NumPy oracle on configs[1] (3 s / 9 s of host time per sampling ratio): evaluated once, shared by the three backward variants."""
    if sr not in _config2_oracle:
        x, yx, gy = config2_inputs()
        xy = yx[:, [0, 2, 1, 4, 3]]
        _config2_oracle[sr] = (ora.roi_align_fwd(x, xy, 7, 7, 0.25, sr), ora.roi_align_bwd(gy, xy, x.shape, 0.25, sr))
    return _config2_oracle[sr]


@pytest.mark.parametrize('sr', [2, 0])
def test_config2_all_512_rois_against_the_oracle(sr, bwd_variant):
    """BASELINE configs[1] exactly as benchmarked - ALL 512 RoIs on the 256 x 200 x 272 map, 7 x 7: forward
    bit-exact, backward within 1e-5, for the reference's sampling grid (sampling_ratio 2) and the adaptive one (0), both
    backward kernels."""
    x, yx, gy = config2_inputs()
    xy = yx[:, [0, 2, 1, 4, 3]]
    xt = torch.from_numpy(x).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = roi_align_2d(xt, torch.from_numpy(xy).to(DEV), 7, 7, 0.25, sampling_ratio=sr)
    y.backward(torch.from_numpy(gy).to(DEV))
    want_y, want_gx = _config2_want(sr)
    np.testing.assert_array_equal(y.detach().cpu().numpy(), want_y)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), want_gx, rtol=1e-5, atol=2e-5 * np.abs(gy).max())



@pytest.mark.parametrize('P', [7, 14])
def test_fpn_backward_coarse_levels_split_matches_oracle(P, bwd_variant):
    """Multi-level backward with most RoIs on the coarse levels (what map_rois_to_fpn_levels produces): the
    RoI-split path (several workgroups per tile + ordered slab sum) against the per-level oracle, with and
    without accumulation, bit-reproducible run to run, and equal (to rounding) to the unsplit path."""
    from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
    rs = np.random.RandomState(40 + P)
    N, C = 2, 8
    shapes = [(40, 48), (20, 24), (10, 12), (5, 6)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    R = 150
    xy = rand_rois_xy(rs, R, N, 40, 48, 0.25)
    lev = rs.choice(4, size=R, p=[0.05, 0.1, 0.35, 0.5]).astype(np.int32)
    gy = rs.standard_normal((R, P, P, C)).astype(np.float32)
    base = [rs.standard_normal((N, h, w, C)).astype(np.float32) for h, w in shapes]
    want = []
    for l, (h, w) in enumerate(shapes):
        sel = np.nonzero(lev == l)[0]
        g = ora.roi_align_bwd(np.ascontiguousarray(gy[sel].transpose(0, 3, 1, 2)), xy[sel], (N, C, h, w), scales[l], 2)
        want.append(g.transpose(0, 2, 3, 1))
    rt, lt, gyt = torch.from_numpy(xy).to(DEV), torch.from_numpy(lev).to(DEV), torch.from_numpy(gy).to(DEV)

    def run(accumulate, use_ws):
        gxs = [torch.from_numpy(b).to(DEV).clone() for b in base]
        if use_ws:
            hd.roi_align_fpn_bwd(gyt, gxs, rt, lt, P, scales, accumulate=accumulate)
        else:
            L, arr_p, Hs, Ws, sc = hd._level_args(gxs, scales)
            _hip.check(_hip.lib().mrcnn_roi_align_fpn_bwd_f32(_hip.ptr(gyt), arr_p, Hs, Ws, sc, L, N, C, _hip.ptr(rt),
                                                               _hip.ptr(lt), R, P, P, 2, int(accumulate), None, 0,
                                                               _hip.stream_ptr()))
        return [g.cpu().numpy() for g in gxs]

    L, arr_p, Hs, Ws, sc = hd._level_args([torch.from_numpy(b) for b in base], scales)
    assert _hip.lib().mrcnn_roi_align_fpn_bwd_workspace_bytes(Hs, Ws, L, N, C, R, P, P, 2) > 0     # these levels do split
    tol = dict(rtol=1e-5, atol=2e-5 * np.abs(gy).max())
    for acc in (False, True):
        a, a2, b = run(acc, True), run(acc, True), run(acc, False)
        for l in range(4):
            ref = want[l] + (base[l] if acc else 0)
            np.testing.assert_allclose(a[l], ref, **tol)
            np.testing.assert_allclose(b[l], ref, **tol)
            np.testing.assert_array_equal(a[l], a2[l])


@pytest.mark.parametrize('R', [127, 128, 513, 3000])
def test_forward_map_order_walk_gives_the_same_rows(R):
    """ABI v6: with scratch the forward ranks the RoIs by (level, image, band of the box centre, centre column) and walks them in that
    order, writing every RoI's OWN output rows: identical bits to the caller-order walk (single level and FPN; RoIs with equal keys,
    bad image indices, edge RoIs; R below the threshold, not a multiple of 64, several ranking workgroups)."""
    from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
    rs = np.random.RandomState(R)
    N, C, P = 2, 64, 7
    shapes = [(100, 136), (50, 68), (25, 34), (13, 17)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    xy = rand_rois_xy(rs, R, N, 100, 136, 0.25)
    xy[:6] = _edge_rois(N, 100, 136, 0.25)
    xy[10:20] = xy[10]                                         # ten RoIs with one key: ties are broken by index
    xy[20, 0], xy[21, 0] = 7, -3                               # image index out of range: rows of zeros either way
    lev = rs.choice(4, size=R, p=[0.1, 0.2, 0.3, 0.4]).astype(np.int32)
    xs = [torch.from_numpy(rs.standard_normal((N, h, w, C)).astype(np.float32)).to(DEV) for h, w in shapes]
    rt, lt = torch.from_numpy(xy).to(DEV), torch.from_numpy(lev).to(DEV)
    lib = _hip.lib()
    assert lib.mrcnn_roi_align_fwd_workspace_bytes(R) >= 4 * R
    out = {}
    try:
        for on in (1, 0):
            _hip.check(lib.mrcnn_roi_align_set_fwd_map_order(on))
            out[on] = (hd.roi_align_fpn_fwd(xs, rt, lt, P, scales),
                       roi_align_2d(xs[0].permute(0, 3, 1, 2), rt, P, P, 0.25))
    finally:
        _hip.check(lib.mrcnn_roi_align_set_fwd_map_order(1))
    for a, b in zip(out[1], out[0]):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    # and the ABI's scratch-less entry point is the caller-order walk
    y = torch.full((R, P, P, C), float('nan'), device=DEV)
    _hip.check(lib.mrcnn_roi_align_fwd_f32(_hip.ptr(xs[0]), 1, N, C, 100, 136, _hip.ptr(rt), R, P, P, 0.25, 2, _hip.ptr(y), _hip.stream_ptr()))
    assert torch.equal(y.permute(0, 3, 1, 2), out[0][1])


def _planned_vs_fused(xs_shapes, scales, N, C, xy, lev, P, sr, accumulate, plan_bytes=None, corrupt=None, split=True, verified=False):
    """gxs of mrcnn_roi_align_fpn_bwd_f32 (fused) and of plan + mrcnn_roi_align_fpn_bwd_planned_f32 on the same operands; the plan header."""
    import ctypes
    from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
    lib = _hip.lib()
    rs = np.random.RandomState(P * 100 + len(xy))
    R = xy.shape[0]
    gyt = torch.from_numpy(rs.standard_normal((R, P, P, C)).astype(np.float32)).to(DEV)
    base = [torch.from_numpy(rs.standard_normal((N, h, w, C)).astype(np.float32)).to(DEV) for h, w in xs_shapes]
    rt = torch.from_numpy(xy).to(DEV)
    lt = torch.from_numpy(lev).to(DEV) if lev is not None else None
    L, _, Hs, Ws, sc = hd._level_args(base, scales)
    nb = lib.mrcnn_roi_align_fpn_bwd_workspace_bytes(Hs, Ws, L, N, C, R, P, P, sr) if split else 0
    ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=DEV)
    out = []
    hdr = None
    for planned in (False, True):
        gxs = [b.clone() if accumulate else torch.full_like(b, float('nan')) for b in base]
        _, arr_p, _, _, _ = hd._level_args(gxs, scales)
        if planned:
            pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, L, N, R, P, P, int(split)) if plan_bytes is None else plan_bytes
            assert pb > 0
            plan = torch.full((pb,), 0x5A, dtype=torch.uint8, device=DEV)
            _hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, L, N, C, _hip.ptr(rt), _hip.ptr(lt), R, P, P, sr, int(split), _hip.ptr(plan),
                                                            pb, _hip.stream_ptr()))
            if corrupt is not None:
                corrupt(plan)
            hdr = plan[:256].view(torch.int32).cpu().numpy().copy()
            _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), arr_p, Hs, Ws, sc, L, N, C, _hip.ptr(rt), _hip.ptr(lt), R, P, P, sr,
                                                               int(accumulate), _hip.ptr(ws) if nb else None, nb, _hip.ptr(plan), pb, int(verified), _hip.stream_ptr()))
        else:           # (the fused multi-level entry wants a level per RoI; the plan pair takes NULL for a single level)
            lf = lt if lt is not None else torch.zeros((R,), dtype=torch.int32, device=DEV)
            _hip.check(lib.mrcnn_roi_align_fpn_bwd_f32(_hip.ptr(gyt), arr_p, Hs, Ws, sc, L, N, C, _hip.ptr(rt), _hip.ptr(lf), R, P, P, sr,
                                                       int(accumulate), _hip.ptr(ws) if nb else None, nb, _hip.stream_ptr()))
        torch.cuda.synchronize()
        out.append(gxs)
    for a, b in zip(*out):
        assert torch.isfinite(a).all()
        assert torch.equal(a, b)
    return hdr


NP_MAGIC = 0x4E504C4E


@pytest.mark.parametrize('P,sr', [(7, 2), (14, 2), (7, 1), (5, 3), (16, 4)])
def test_planned_backward_equals_the_fused_backward_config2(P, sr):
    """(r5) mrcnn_roi_align_fpn_bwd_plan_f32 + mrcnn_roi_align_fpn_bwd_planned_f32 (entry lists built ahead, lean streaming backward) on
    configs[1]'s map with its 512 RoIs plus edge RoIs, a bad image index and a three-segment RoI count (R > 1024): the same bits as the
    fused backward for 7x7 / 14x14 and odd pooled sizes / sampling ratios; the plan's header says it was used (magic, no overflow)."""
    x, yx, _ = config2_inputs()
    H, W = x.shape[2], x.shape[3]
    rs = np.random.RandomState(P * 10 + sr)
    rois = np.concatenate([yx[:, [0, 2, 1, 4, 3]], _edge_rois(1, H, W, 0.25), rand_rois_xy(rs, 600, 1, H, W, 0.25),
                           np.array([[3, 10, 10, 200, 200], [-1, 5, 5, 100, 80]], np.float32)], 0)
    assert rois.shape[0] > 1024
    hdr = _planned_vs_fused([(H, W)], [0.25], 1, 256, rois, None, P, sr, accumulate=False, split=False)
    assert hdr[0] == NP_MAGIC and hdr[1] == 0 and hdr[5] == rois.shape[0] and hdr[6] == P
    # the verified form (the lean kernel alone, no launch behind it): the header says the plan is complete, so the caller may ask for it
    _planned_vs_fused([(H, W)], [0.25], 1, 256, rois, None, P, sr, accumulate=False, split=False, verified=True)


def test_plan_status_query():
    """mrcnn_roi_align_bwd_plan_status: (header valid, tiles flagged, pool nodes used) read back after the builder - what a caller checks
    before it passes plan_verified."""
    from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
    rs = np.random.RandomState(5)
    xy = rand_rois_xy(rs, 300, 1, 40, 48, 0.25)
    rt = torch.from_numpy(xy).to(DEV)
    maps = [torch.empty((1, 40, 48, 64), device=DEV)]
    plan = hd.roi_align_fpn_bwd_plan(maps, rt, None, 7, [0.25])
    ok, flagged, used = hd.roi_align_bwd_plan_status(plan)
    assert ok and flagged == 0 and used >= 0
    plan[:4].zero_()
    assert hd.roi_align_bwd_plan_status(plan)[0] is False


@pytest.mark.parametrize('P', [7, 14])
@pytest.mark.parametrize('accumulate', [False, True])
def test_planned_backward_fpn_levels_split_and_accumulate(P, accumulate):
    """The multi-level form as the training step calls it: most RoIs on the coarse levels (RoI-split slabs + ordered sum), two images,
    ragged maps, channel tail (C = 260: two channel passes), overwrite and accumulate modes - planned == fused bit for bit."""
    rs = np.random.RandomState(7 + P)
    N, C = 2, 260
    shapes = [(41, 50), (21, 25), (11, 13), (6, 7)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    R = 700
    xy = rand_rois_xy(rs, R, N, 41, 50, 0.25)
    lev = rs.choice(4, size=R, p=[0.05, 0.1, 0.35, 0.5]).astype(np.int32)
    hdr = _planned_vs_fused(shapes, scales, N, C, xy, lev, P, 2, accumulate)
    assert hdr[0] == NP_MAGIC and hdr[1] == 0
    assert hdr[2] > 0           # the crowded coarse patches flushed more than once: pool nodes were chained


def test_fused_backward_second_channel_pass_on_ragged_maps_matches_oracle():
    """C > 256 (two channel passes) on maps whose edge tiles have waves without a patch, more than one 512-RoI segment, split levels: the
    waves without a patch used to skip the workgroup barrier between the channel passes (wrong values in channels >= 256; found by the
    planned-vs-fused comparison above).  Fused and planned backward against the per-level oracle."""
    from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
    rs = np.random.RandomState(14)
    N, C, P, R = 2, 260, 7, 700
    shapes = [(41, 50), (21, 25), (11, 13), (6, 7)]
    scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    xy = rand_rois_xy(rs, R, N, 41, 50, 0.25)
    lev = rs.choice(4, size=R, p=[0.05, 0.1, 0.35, 0.5]).astype(np.int32)
    gy = rs.standard_normal((R, P, P, C)).astype(np.float32)
    rt, lt, gyt = torch.from_numpy(xy).to(DEV), torch.from_numpy(lev).to(DEV), torch.from_numpy(gy).to(DEV)
    for planned in (False, True):
        gxs = [torch.full((N, h, w, C), float('nan'), device=DEV) for h, w in shapes]
        plan = hd.roi_align_fpn_bwd_plan(gxs, rt, lt, P, scales) if planned else None
        hd.roi_align_fpn_bwd(gyt, gxs, rt, lt, P, scales, accumulate=False, plan=plan)
        for l, (h, w) in enumerate(shapes):
            sel = np.nonzero(lev == l)[0]
            want = ora.roi_align_bwd(np.ascontiguousarray(gy[sel].transpose(0, 3, 1, 2)), xy[sel], (N, C, h, w), scales[l], 2).transpose(0, 2, 3, 1)
            np.testing.assert_allclose(gxs[l].cpu().numpy(), want, rtol=1e-5, atol=2e-5 * np.abs(gy).max(), err_msg='planned %s level %d' % (planned, l))


def test_planned_backward_falls_back_when_the_plan_does_not_hold():
    """A plan buffer too small for the pool (overflow flag set by the builder), a foreign buffer (no magic), a plan built for other RoIs
    counts: the planned entry point detects it on the device and computes the fused path - the same bits, never garbage."""
    rs = np.random.RandomState(3)
    N, C, P = 1, 64, 14
    shapes, scales = [(8, 8)], [1 / 32.]
    R = 600
    xy = rand_rois_xy(rs, R, N, 8, 8, 1 / 32.)
    import ctypes
    lib = _hip.lib()
    Hs, Ws = (ctypes.c_int * 1)(8), (ctypes.c_int * 1)(8)
    full = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, P, P, 0)
    hdr = _planned_vs_fused(shapes, scales, N, C, xy, None, P, 2, False, split=False)
    assert hdr[0] == NP_MAGIC and hdr[1] == 0 and hdr[2] > 8           # 600 RoIs on 4 patches: long chains
    stride = 3584           # one 14x14 node (96 entries) rounded to 256 bytes
    small = 256 + (4 * 4 + 3) * stride                                  # room for the patch slots and three pool nodes only
    assert small < full
    hdr = _planned_vs_fused(shapes, scales, N, C, xy, None, P, 2, False, plan_bytes=small, split=False)
    assert hdr[0] == NP_MAGIC and hdr[1] >= 1                           # failed allocations counted, the tile flagged; result still the fused one
    hdr = _planned_vs_fused(shapes, scales, N, C, xy, None, P, 2, False, split=False, corrupt=lambda p: p[:4].zero_())
    assert hdr[0] == 0
    def other_count(p):
        h = p[:256].view(torch.int32)
        h[5] = R - 1
    _planned_vs_fused(shapes, scales, N, C, xy, None, P, 2, False, split=False, corrupt=other_count)
