"""GEMM-level accuracy of the float32-accurate emulation: every distinct convolution geometry of the benchmarked
training step (BASELINE.json configs[2]: two 1024x1024 images, 512 sampled RoIs) x the passes it occurs in, at FULL size - so the launch
plans, tiles, split-K factors and kernel families (k_conv_igemm<..., 3>, the plane GEMMs k_pgemm_pp / k_pgemm_gpp of planes_gemm.h) are
the ones the step runs - with split mode 3 (three bf16 planes = the float32 operand exactly, six products on v_mfma_f32_32x32x16_bf16,
float32 accumulate) against the float32-MFMA kernels ON THE SAME OPERANDS:

    rms |emulated - float64| <= 1.25 x rms |float32-MFMA kernel - float64|   (+ 2e-8 of the tensor scale)
    max |emulated - float64| <= 2    x max |float32-MFMA kernel - float64|   (+ 2e-7 of the tensor scale: one float32 rounding of the output)

(the maximum over ~10^6 sampled outputs is itself a noisy statistic - two float32 realisations of one F(4x4) Winograd layer differ by
+-30 % in it - so the 1.25 bar is put on the root mean square and the maximum gets a factor of two)

The float64 reference is plain PyTorch on the device (gathered patches x weights in float64 for 4096 sampled output positions of the
forward / backward-data passes, every tap of the filter gradient over ALL pixels), independent of both kernels.  Then adversarial
operands on three geometries: 2^+-40 exponent spread inside a K row, cancelling sums, values below 2^-100, and +-Inf / NaN, which must
come out non-finite (the planes of an Inf are Inf, Inf - Inf = NaN: the emulated result is NaN where the float32 kernel gives Inf or NaN)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from chainer_maskrcnn import _hip  # noqa: E402
from chainer_maskrcnn._hip import nn as hnn  # noqa: E402

DEV = 'cuda:0'
# (N, H, W, Cin, Cout, KH, KW, stride, pad), per-pass Winograd tiles of the call (forward, backward-data, backward-filter), passes it
# occurs in - the distinct convolution calls of one bs-2 1024^2 step (bench_step._replay_split's `geoms`; padded channel counts)
STEP_GEOMS = [
    ((512, 14, 14, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),      # mask head (x4)
    ((512, 14, 14, 256, 384, 1, 1, 1, 0), (0, 0, 0), 'fdw'),      # merged deconvolution + 1x1
    ((512, 7, 7, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),        # box head conv
    ((512, 1, 1, 12544, 1024, 1, 1, 1, 0), (2, 0, 0), 'fdw'),     # fc1
    ((512, 1, 1, 1024, 1024, 1, 1, 1, 0), (2, 0, 0), 'fdw'),      # fc2
    ((512, 1, 1, 1024, 96, 1, 1, 1, 0), (2, 0, 0), 'fdw'),        # score + loc
    ((2, 256, 256, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),      # FPN conv_p2 / RPN conv on p2
    ((2, 128, 128, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),
    ((2, 64, 64, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),
    ((2, 32, 32, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),
    ((2, 16, 16, 256, 256, 3, 3, 1, 1), (0, 0, 0), 'fdw'),
    ((2, 256, 256, 256, 32, 1, 1, 1, 0), (2, 0, 0), 'fdw'),       # RPN loc + score heads
    ((2, 128, 128, 256, 32, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 16, 16, 256, 32, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 256, 256, 256, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),      # FPN lateral p2
    ((2, 128, 128, 512, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 64, 64, 1024, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 32, 32, 2048, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),       # toplayer
    ((2, 1024, 1024, 4, 64, 7, 7, 2, 3), (2, 0, 0), 'fw'),        # conv1 (image layer: no data gradient)
    ((2, 256, 256, 64, 64, 1, 1, 1, 0), (2, 0, 0), 'fdw'),        # res2
    ((2, 256, 256, 64, 64, 3, 3, 1, 1), (2, 0, 0), 'fdw'),
    ((2, 256, 256, 64, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 256, 256, 256, 64, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 256, 256, 256, 128, 1, 1, 2, 0), (2, 0, 0), 'fw'),       # res3 (strided 1x1: the data gradient runs on the lattice)
    ((2, 256, 256, 256, 512, 1, 1, 2, 0), (2, 0, 0), 'fw'),
    ((2, 128, 128, 256, 128, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 128, 128, 256, 512, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 128, 128, 128, 128, 3, 3, 1, 1), (2, 0, 0), 'fdw'),
    ((2, 128, 128, 128, 512, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 128, 128, 512, 128, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 128, 128, 512, 256, 1, 1, 2, 0), (2, 0, 0), 'fw'),       # res4
    ((2, 128, 128, 512, 1024, 1, 1, 2, 0), (2, 0, 0), 'fw'),
    ((2, 64, 64, 512, 256, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 64, 64, 512, 1024, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 64, 64, 256, 1024, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 64, 64, 1024, 256, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 64, 64, 1024, 512, 1, 1, 2, 0), (2, 0, 0), 'fw'),        # res5
    ((2, 64, 64, 1024, 2048, 1, 1, 2, 0), (2, 0, 0), 'fw'),
    ((2, 32, 32, 1024, 512, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 32, 32, 1024, 2048, 1, 1, 1, 0), (2, 0, 0), 'd'),
    ((2, 32, 32, 512, 512, 3, 3, 1, 1), (2, 0, 0), 'fdw'),
    ((2, 32, 32, 512, 2048, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 32, 32, 2048, 512, 1, 1, 1, 0), (2, 0, 0), 'fdw'),
    ((2, 32, 32, 256, 256, 1, 1, 2, 0), (2, 0, 0), 'fw'),         # p6 subsample path
]
NS = 4096


def _ref_fwd(x, w, geom, sample):
    """float64 y at the sampled output positions (flat indices into (N, Ho, Wo)): (S, Cout)."""
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    Ho, Wo = hnn.conv_out(H, KH, s, p), hnn.conv_out(W, KW, s, p)
    n, ho, wo = sample // (Ho * Wo), (sample // Wo) % Ho, sample % Wo
    xp = F.pad(x, (0, 0, p, p, p, p))
    acc = torch.zeros((sample.numel(), Cout), dtype=torch.float64, device=x.device)
    for kh in range(KH):
        for kw in range(KW):
            acc += xp[n, ho * s + kh, wo * s + kw, :].double() @ w[:, kh, kw, :].double().t()
    return acc


def _ref_bwd_data(gy, w, geom, sample):
    """float64 gx at sampled input positions (stride 1): (S, Cin)."""
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    assert s == 1
    n, hi, wi = sample // (H * W), (sample // W) % H, sample % W
    gp = F.pad(gy, (0, 0, KW - 1, KW - 1, KH - 1, KH - 1))
    acc = torch.zeros((sample.numel(), Cin), dtype=torch.float64, device=gy.device)
    for kh in range(KH):
        for kw in range(KW):
            acc += gp[n, hi + p - kh + (KH - 1), wi + p - kw + (KW - 1), :].double() @ w[:, kh, kw, :].double()
    return acc


def _ref_bwd_filter(x, gy, geom):
    """float64 gw (Cout, KH, KW, Cin) over all pixels."""
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    Ho, Wo = gy.shape[1], gy.shape[2]
    xp = F.pad(x, (0, 0, p, p, p, p))
    g2 = gy.reshape(-1, Cout).double()
    out = torch.empty((Cout, KH, KW, Cin), dtype=torch.float64, device=x.device)
    for kh in range(KH):
        for kw in range(KW):
            xs = xp[:, kh:kh + (Ho - 1) * s + 1:s, kw:kw + (Wo - 1) * s + 1:s, :].reshape(-1, Cin).double()
            out[:, kh, kw, :] = g2.t() @ xs
    return out


def _run_pass(kind, geom, x, w, gy):
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    if kind == 'f':
        return hnn.conv2d_fwd_raw(x, w, None, s, p, False)
    if kind == 'd':
        return hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), s, p)
    return hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), s, p, False)[0]


def _operands(geom, seed, dev=DEV):
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    g = torch.Generator(device='cpu').manual_seed(seed)
    Ho, Wo = hnn.conv_out(H, KH, s, p), hnn.conv_out(W, KW, s, p)
    x = torch.randn((N, H, W, Cin), generator=g).to(dev)
    w = (torch.randn((Cout, KH, KW, Cin), generator=g) / (KH * KW * Cin) ** 0.5).to(dev)
    gy = (torch.randn((N, Ho, Wo, Cout), generator=g) * 1e-3).to(dev)          # gradients of this network are 1e-6 .. 1e-3
    return x, w, gy


def _errors(kind, geom, x, w, gy, seed):
    """((max, rms) |f32 kernel - ref|, (max, rms) |emulation - ref|, max |ref|) on the sampled outputs of one pass."""
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    g = torch.Generator(device='cpu').manual_seed(seed + 1)
    lib = _hip.lib()
    outs = {}
    for mode in (0, 3):
        _hip.check(lib.mrcnn_conv2d_set_split_operands(mode, mode, mode))
        outs[mode] = _run_pass(kind, geom, x, w, gy)
        assert torch.equal(outs[mode], _run_pass(kind, geom, x, w, gy)), 'not reproducible'
    _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
    if kind == 'f':
        Ho, Wo = outs[0].shape[1], outs[0].shape[2]
        sample = torch.randint(0, N * Ho * Wo, (NS,), generator=g).to(x.device)
        ref = _ref_fwd(x, w, geom, sample)
        got = {m: o.reshape(-1, Cout)[sample].double() for m, o in outs.items()}
    elif kind == 'd':
        sample = torch.randint(0, N * H * W, (NS,), generator=g).to(x.device)
        ref = _ref_bwd_data(gy, w, geom, sample)
        got = {m: o.reshape(-1, Cin)[sample].double() for m, o in outs.items()}
    else:
        ref = _ref_bwd_filter(x, gy, geom)
        got = {m: o.double() for m, o in outs.items()}
    st = lambda d: (float(d.abs().max()), float(d.pow(2).mean().sqrt()))
    return st(got[0] - ref), st(got[3] - ref), float(ref.abs().max())


# `-m gpu` runs one geometry per kernel family / plan kind of the step (the rows below); the other 27 carry `gpu_long` (tests/conftest.py)
SHORT = {(512, 14, 14, 256, 256, 3, 3, 1, 1), (512, 14, 14, 256, 384, 1, 1, 1, 0), (512, 7, 7, 256, 256, 3, 3, 1, 1), (512, 1, 1, 12544, 1024, 1, 1, 1, 0),
         (512, 1, 1, 1024, 96, 1, 1, 1, 0), (2, 256, 256, 256, 256, 3, 3, 1, 1), (2, 16, 16, 256, 256, 3, 3, 1, 1), (2, 256, 256, 256, 32, 1, 1, 1, 0),
         (2, 256, 256, 256, 256, 1, 1, 1, 0), (2, 32, 32, 2048, 256, 1, 1, 1, 0), (2, 1024, 1024, 4, 64, 7, 7, 2, 3), (2, 256, 256, 64, 64, 3, 3, 1, 1),
         (2, 256, 256, 256, 512, 1, 1, 2, 0), (2, 64, 64, 512, 1024, 1, 1, 1, 0), (2, 32, 32, 512, 512, 3, 3, 1, 1), (2, 32, 32, 2048, 512, 1, 1, 1, 0),
         (2, 32, 32, 256, 256, 1, 1, 2, 0)}
assert SHORT <= {c[0] for c in STEP_GEOMS}


@pytest.mark.parametrize('case', [c if c[0] in SHORT else pytest.param(c, marks=pytest.mark.gpu_long) for c in STEP_GEOMS],
                         ids=lambda c: 'x'.join(str(v) for v in c[0]) + '_' + c[2])
def test_emulation_is_as_accurate_as_the_float32_mfma_on_every_step_geometry(case):
    geom, tiles, passes = case
    lib = _hip.lib()
    keep = hnn.winograd_pass_tiles()
    hnn.set_winograd_pass_tiles(*tiles)
    try:
        x, w, gy = _operands(geom, 4000 + sum(geom))
        for kind in passes:
            (m32, r32), (m6, r6), scale = _errors(kind, geom, x, w, gy, 17)
            assert m32 <= 3e-4 * scale, (kind, 'float32 kernel off', m32 / scale)        # (F(4x4) layers: ~1e-5 of the scale)
            assert r6 <= 1.25 * r32 + 2e-8 * scale, (kind, 'rms: emulation %.3e, float32 MFMA %.3e (of scale)' % (r6 / scale, r32 / scale))
            assert m6 <= 2.0 * m32 + 2e-7 * scale, (kind, 'max: emulation %.3e, float32 MFMA %.3e (of scale)' % (m6 / scale, m32 / scale))
    finally:
        _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
        hnn.set_winograd_pass_tiles(*keep)


ADV_GEOMS = [((16, 14, 14, 256, 256, 3, 3, 1, 1), (2, 2, 2)),        # Winograd F(2x2): the transforms see the adversarial values too
             ((2, 32, 32, 256, 512, 1, 1, 1, 0), (2, 0, 0)),          # direct 1x1, all three passes on k_conv_igemm<..., 3>
             ((64, 14, 14, 64, 96, 3, 3, 1, 1), (2, 2, 2))]           # direct 3x3 in all passes (F(2x2) needs 256 channels)


def _sum_abs(kind, geom, x, w, gy):
    """float64 sum |a||b| per output (the scale a floating-point dot product's error is proportional to), whole tensors (small cases)."""
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    xa, wa, ga = x.abs().double().cpu(), w.abs().double().cpu(), gy.abs().double().cpu()
    if kind == 'f':
        return F.conv2d(xa.permute(0, 3, 1, 2), wa.permute(0, 3, 1, 2), None, s, p).permute(0, 2, 3, 1)
    if kind == 'd':
        return F.conv_transpose2d(ga.permute(0, 3, 1, 2), wa.permute(0, 3, 1, 2), None, s, p).permute(0, 2, 3, 1)
    return torch.nn.grad.conv2d_weight(xa.permute(0, 3, 1, 2), (Cout, Cin, KH, KW), ga.permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)


def _exact(kind, geom, x, w, gy):
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    xd, wd, gd = x.double().cpu(), w.double().cpu(), gy.double().cpu()
    if kind == 'f':
        return F.conv2d(xd.permute(0, 3, 1, 2), wd.permute(0, 3, 1, 2), None, s, p).permute(0, 2, 3, 1)
    if kind == 'd':
        return F.conv_transpose2d(gd.permute(0, 3, 1, 2), wd.permute(0, 3, 1, 2), None, s, p).permute(0, 2, 3, 1)
    return torch.nn.grad.conv2d_weight(xd.permute(0, 3, 1, 2), (Cout, Cin, KH, KW), gd.permute(0, 3, 1, 2), s, p).permute(0, 2, 3, 1)


@pytest.mark.parametrize('case', ADV_GEOMS, ids=lambda c: 'x'.join(str(v) for v in c[0]))
@pytest.mark.parametrize('what', ['exponent_spread', 'cancellation', 'tiny'])
def test_emulation_on_adversarial_operands(case, what):
    """Error measured against sum |a||b| per output element (the quantity float32 accumulation error is proportional to): the
    emulation within 1.25 x the float32 MFMA kernel's root mean square and 2 x its worst element (+ one output rounding), for
      exponent_spread  every operand element scaled by an independent power of two in 2^-40 .. 2^+40
      cancellation     operands arranged so that every dot product cancels to ~1e-6 of its sum of magnitudes
      tiny             activations / gradients of magnitude 2^-110 .. 2^-100 (the low planes of such values are still normal
                       numbers; below ~2^-118 the lo, then the mid plane would turn subnormal and the emulation would degrade towards
                       the bf16 hi plane alone - products of such operands are themselves below float32's normal range)."""
    geom, tiles = case
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    lib = _hip.lib()
    keep = hnn.winograd_pass_tiles()
    hnn.set_winograd_pass_tiles(*tiles)
    g = torch.Generator(device='cpu').manual_seed(77 + sum(geom))
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, KH, KW, Cin), generator=g) / (KH * KW * Cin) ** 0.5
    gy = torch.randn((N, H, W, Cout), generator=g)
    wino = KH == 3 and Cin >= 256
    if what == 'exponent_spread':
        # (behind a Winograd transform the spread is 2^+-4: the transforms add and subtract neighbouring values and their own rounding,
        # relative to the direct algorithm's sum |a||b|, grows with the spread in BOTH arithmetics - the ratio below is the test)
        sp, sw = (4, 2) if wino else (40, 20)
        x = x * torch.exp2(torch.randint(-sp, sp + 1, x.shape, generator=g).float())
        gy = gy * torch.exp2(torch.randint(-sp, sp + 1, gy.shape, generator=g).float())
        w = w * torch.exp2(torch.randint(-sw, sw + 1, w.shape, generator=g).float())
    elif what == 'cancellation':
        # pairs of channels carry +v and -v(1 + 1e-6): with equal weights on the pair every dot product nearly cancels
        x[..., 1::2] = -x[..., 0::2] * (1 + 1e-6)
        w[..., 1::2] = w[..., 0::2]
        gy[..., 1::2] = -gy[..., 0::2] * (1 + 1e-6)
        w[1::2] = w[0::2]
    xn = x
    if what == 'tiny':
        # one operand of every pass is tiny, its partner normal (two tiny operands multiply to zero in float32 and in the emulation alike):
        # forward x, backward-data gy, filter gradient gy with a normal x
        x, gy = x * 2.0 ** -105, gy * 2.0 ** -105
    x, xn, w, gy = x.to(DEV), xn.to(DEV), w.to(DEV), gy.to(DEV)
    try:
        for kind in 'fdw':
            if kind == 'w':
                x = xn
            sab = _sum_abs(kind, geom, x, w, gy)
            ref = _exact(kind, geom, x, w, gy)
            rel, rms = {}, {}
            for mode in (0, 3):
                _hip.check(lib.mrcnn_conv2d_set_split_operands(mode, mode, mode))
                out = _run_pass(kind, geom, x, w, gy).double().cpu()
                assert torch.isfinite(out).all(), (kind, mode)
                q = (out - ref).abs() / sab.clamp_min(1e-300)
                rel[mode], rms[mode] = float(q.max()), float(q.pow(2).mean().sqrt())
            # float32 accumulation: a few ulp of sum |a||b| (1.2e-7 each); Winograd adds its transforms' rounding on top
            assert rel[0] <= (2e-3 if wino else 2e-5), (kind, what, 'float32 kernel', rel[0])
            assert rms[3] <= 1.25 * rms[0] + 1e-8, (kind, what, 'rms: emulation %.3e float32 MFMA %.3e' % (rms[3], rms[0]))
            assert rel[3] <= 2.0 * rel[0] + 1.2e-7, (kind, what, 'max: emulation %.3e float32 MFMA %.3e' % (rel[3], rel[0]))
    finally:
        _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
        hnn.set_winograd_pass_tiles(*keep)


@pytest.mark.parametrize('case', ADV_GEOMS[1:], ids=lambda c: 'x'.join(str(v) for v in c[0]))
def test_emulation_keeps_non_finite_values_non_finite(case):
    """+-Inf and NaN in an operand: every output element whose dot product touches one is non-finite in both arithmetics (the
    float32 MFMA gives +-Inf or NaN; the planes of an Inf are (Inf, NaN, NaN) - Inf - Inf - so the emulation gives NaN), every other
    element is finite and within the usual bar.  (Direct kernels; behind a Winograd transform a non-finite input reaches every
    output of its tile in both arithmetics.)"""
    geom, tiles = case
    N, H, W, Cin, Cout, KH, KW, s, p = geom
    lib = _hip.lib()
    keep = hnn.winograd_pass_tiles()
    hnn.set_winograd_pass_tiles(*tiles)
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((Cout, KH, KW, Cin), generator=g) / (KH * KW * Cin) ** 0.5
    x[0, 3, 4, 5] = float('inf'); x[1, 7, 2, 9] = float('-inf'); x[1, 0, 0, 1] = float('nan')
    clean = x.clone()
    bad = ~torch.isfinite(x)
    clean[bad] = 0.0
    touched = F.conv2d(bad.double().permute(0, 3, 1, 2), torch.ones((Cout, Cin, KH, KW), dtype=torch.float64), None, s, p).permute(0, 2, 3, 1) > 0
    ref = F.conv2d(clean.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, s, p).permute(0, 2, 3, 1)
    try:
        for mode in (0, 3):
            _hip.check(lib.mrcnn_conv2d_set_split_operands(mode, mode, mode))
            y = hnn.conv2d_fwd_raw(x.to(DEV), w.to(DEV), None, s, p, False).cpu()
            assert not torch.isfinite(y[touched]).any(), mode
            assert torch.isfinite(y[~touched]).all(), mode
            assert float((y[~touched].double() - ref[~touched]).abs().max()) <= 2e-6 * float(ref.abs().max()), mode
    finally:
        _hip.check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
        hnn.set_winograd_pass_tiles(*keep)


def _planes(x, layout):
    """float32 (R, C) -> three bf16 planes through the library's own split kernel (layout 0 = P16, 1 = P16R4, 2 = PR)."""
    R_, C_ = x.shape
    out = torch.empty((R_ * C_ * 3,), dtype=torch.int16, device=x.device)
    _hip.check(_hip.lib().mrcnn_debug_split_planes_f32(_hip.ptr(x), _hip.ptr(out), R_, C_, layout, _hip.stream_ptr()))
    return out


@pytest.mark.parametrize('nb,rows,K,N', [(1, 256, 16, 256), (3, 256, 48, 256), (2, 512, 80, 512), (1, 768, 272, 256), (5, 256, 32, 256)])
def test_plane_gemm_forward_kind_edge_shapes(nb, rows, K, N):
    """k_pgemm_pp (the Winograd forward / backward-data GEMM of the emulated arithmetic) called directly through the measurement entry
    on shapes the step does not have: one K step per tile, odd numbers of K steps (the three-stage ring wraps inside a tile and across
    the tiles of the persistent walk), more tiles than one round, several batches.  C[b] = A[b] B[b]^T against float64."""
    g = torch.Generator(device='cpu').manual_seed(1000 * nb + K)
    A = (torch.randn((nb * rows, K), generator=g) * torch.exp2(torch.randint(-6, 6, (nb * rows, 1), generator=g).float())).to(DEV)
    B = (torch.randn((nb * N, K), generator=g) / K ** 0.5).to(DEV)
    C = torch.full((nb * rows, N), float('nan'), device=DEV)
    _hip.check(_hip.lib().mrcnn_debug_planes_gemm(4, _hip.ptr(A), _hip.ptr(_planes(B, 1)), _hip.ptr(C), nb * rows, N, K, rows, nb, 1, 256, 256,
                                                  _hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.isfinite(C).all(), 'unwritten output'
    for b in range(nb):
        ref = A[b * rows:(b + 1) * rows].double() @ B[b * N:(b + 1) * N].double().t()
        err = float((C[b * rows:(b + 1) * rows].double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (b, err)


@pytest.mark.parametrize('nb,rows,M,N,ks', [(1, 32, 256, 256, 1), (2, 96, 256, 512, 3), (3, 544, 512, 256, 2), (1, 2080, 256, 256, 5)])
def test_plane_gemm_filter_gradient_kind_edge_shapes(nb, rows, M, N, ks):
    """k_pgemm_gpp (the Winograd filter-gradient GEMM): dU[b] = W[b]^T V[b] summed over `rows` tiles in `ks` splits - a single pair
    of K steps, splits of unequal length (the last one shorter), several batches - against float64."""
    g = torch.Generator(device='cpu').manual_seed(17 * nb + rows)
    A = (torch.randn((nb * rows, M), generator=g) * 1e-3).to(DEV)
    B = torch.randn((nb * rows, N), generator=g).to(DEV)
    C = torch.full((ks, M, nb, N), float('nan'), device=DEV)
    _hip.check(_hip.lib().mrcnn_debug_planes_gemm(5, _hip.ptr(_planes(A, 2)), _hip.ptr(B), _hip.ptr(C), M, N, rows, rows, nb, ks, 256, 256,
                                                  _hip.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.isfinite(C).all(), 'unwritten output'
    Cs = C.double().sum(dim=0)
    for b in range(nb):
        ref = A[b * rows:(b + 1) * rows].double().t() @ B[b * rows:(b + 1) * rows].double()
        err = float((Cs[:, b, :] - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (b, err)
