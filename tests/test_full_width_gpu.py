"""Full-size parity: the REAL ResNet-50-FPN Mask R-CNN (80 classes, full width and depth) - not the
reduced test network - against the float64 oracle (oracle/model.py) on whole images:

  (i)   activations p2..p6, RPN locs / scores, box-head and mask-head outputs <= 1e-3 of their tensor scale
        (BASELINE.json north_star: "conv activations and losses within 1e-3 relative fp32");
  (ii)  every parameter gradient < max(1e-3, 6 x floor) (see _check), floor = the SAME oracle evaluated in float32 on the CPU
        against its float64 self - the larger of two float32 realisations (oneDNN convolutions / plain im2col ones, i.e. two
        summation orders): the float32 noise of this network (training-mode BatchNorm over a few hundred pixels, ReLU /
        max-pool decisions on values within rounding of a tie), which no float32 implementation can beat;
  (iii) the layer with the largest recorded error (res5/b2/conv2/W, profiles/r01_full_width_parity.txt) in isolation:
        the oracle's own x and gy of that layer through mrcnn_conv2d_bwd_filter_f32 <= 2e-5;
for the direct kernels, Winograd F(2x2,3x3), the shipped configuration (backward passes F(4x4) where cheaper; forward
F(2x2) in the ResNet, F(4x4) behind it) and the opt-in 'fast' configuration (F(4x4) in the ResNet's forward pass too:
activation / loss bars only, see _check) - and for the Keypoint R-CNN of train_keypoints.py (configs[4]) as shipped.
The per-tensor table is written to gpurun_out/ (committed copy: profiles/r02_full_width_parity_*.txt)."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import model as om

pytestmark = pytest.mark.gpu

from chainer_maskrcnn import _hip  # noqa: E402
from chainer_maskrcnn._hip import nn as hnn  # noqa: E402
from chainer_maskrcnn.nn import core  # noqa: E402
from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss, calc_keypoint_loss  # noqa: E402
from chainer_maskrcnn.utils.synthetic import make_batch  # noqa: E402

DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# name -> (Winograd thresholds (min channels, min pixels, tile), per-pass tiles (forward, backward-data, backward-filter))
MODES = {'direct': ((100000, 1 << 30, 0), (0, 0, 0)),
         'winograd_f2': ((256, 2048, 2), (0, 0, 0)),
         # the shipped configuration: backward F(4x4) where cheaper; forward F(2x2) in the ResNet, F(4x4) where cheaper in the
         # FPN / RPN / head convolutions (nn/core.py LAYER_TILE_HINTS, profiles/r02_winograd_layer_probe.txt)
         'shipped': ((256, 2048, 0), (2, 0, 0)),
         'uniform_f2_forward': ((256, 2048, 0), (2, 0, 0), False),   # the same without the per-layer hints
         'fast': ((256, 2048, 0), (0, 0, 0)),           # F(4x4) in the forward pass too (opt-in)
         # EXPLORATORY opt-in: the shipped selection with the forward-kind GEMMs (forward convolutions and both Winograd batched
         # GEMMs) on three-term split-bf16 operands (mrcnn_conv2d_set_split_bf16): the same bars as 'shipped'
         'split_bf16': ((256, 2048, 0), (2, 0, 0)), 'split_f16_fwd': ((256, 2048, 0), (2, 0, 0)), 'split_f16': ((256, 2048, 0), (2, 0, 0)),
         'split_f16_fwd_only': ((256, 2048, 0), (2, 0, 0)), 'split_bf16_bwd_only': ((256, 2048, 0), (2, 0, 0)),
         'bf16x6': ((256, 2048, 0), (2, 0, 0)), 'bf16x6_fwd': ((256, 2048, 0), (2, 0, 0)),
         'bf16x6_fwd_only': ((256, 2048, 0), (2, 0, 0)), 'bf16x6_bwd_only': ((256, 2048, 0), (2, 0, 0)),
         # measurement: bf16x6 in every pass except the FORWARD pass of the layers behind the backbone (no BatchNorm behind them)
         'bf16x6_backbone_fwd': ((256, 2048, 0), (2, 0, 0)),
         # ... and the other way round: float32 forward in the backbone (c2 .. c5 bit-identical to the float32 step), emulated forward behind it
         'bf16x6_behind_backbone_fwd': ((256, 2048, 0), (2, 0, 0))}
# split operands per pass (forward, backward-data, backward-filter) of the exploratory modes: 1 = bf16 hi / lo planes, 2 = half planes
SPLIT = {'split_bf16': (1, 1, 1), 'split_f16_fwd': (2, 1, 1), 'split_f16': (2, 2, 2), 'split_f16_fwd_only': (2, 0, 0), 'split_bf16_bwd_only': (0, 1, 1),
         'bf16x6': (3, 3, 3), 'bf16x6_fwd': (3, 1, 1), 'bf16x6_fwd_only': (3, 0, 0), 'bf16x6_bwd_only': (0, 3, 3), 'bf16x6_backbone_fwd': (3, 3, 3), 'bf16x6_behind_backbone_fwd': (3, 3, 3)}
DEFAULT = MODES['shipped']
NAMES = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
TAP = 'extractor/resnet/res5/b2'
_cache = {}


def _model(keypoints=False):
    key = 'mk' if keypoints else 'm'
    if key not in _cache:
        if keypoints:       # train_keypoints.py's model: 1 class, 17 keypoints, 8 mask convs, 56x56 heat maps
            m = MaskRCNN(n_fg_class=1, n_keypoints=17, head_arch='fpn_keypoint', device=DEV, seed=5)
            chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_keypoint_loss, binary_mask=False, mask_rows='all')
        else:
            m = MaskRCNN(n_fg_class=80, device=DEV, seed=5)
            chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss, mask_rows='all')
        chain.keep_outputs = True
        _cache[key] = (m, chain)
    return _cache[key]


def _targets(chain):
    t = {k: v.cpu().numpy() for k, v in chain.targets.items() if torch.is_tensor(v)}
    t['gt_rpn_loc'], t['gt_rpn_label'] = (x.cpu().numpy() for x in chain.rpn_targets)
    t['mask_rois_xy5'], t['mask_levels'], t['mask_label'] = (x.cpu().numpy() for x in chain.mask_inputs)
    return t


def _oracle(m, t, img4, dtype, tap=None, keypoints=False):
    """One oracle step (forward + backward) in `dtype` on the device's sampled targets."""
    om.set_dtype(dtype)
    try:
        ps = m.ps
        params = {n: ps.p(n).detach().cpu().to(dtype).requires_grad_(True) for n in ps.names()}
        kw = dict(mask_conv_names=['mask_convs/%d' % i for i in range(len(m.head.mask_convs))], n_keypoints=17) if keypoints else {}
        o = om.OracleStep(params, tuple(len(s) for s in m.extractor.stages), m.head.n_class, m.head.LOC0, tap=tap, **kw)
        out = o.losses(img4.to(dtype), t)
        sum(out[k] for k in NAMES).backward()
        grads = {n: (params[n].grad if params[n].grad is not None else torch.zeros_like(params[n])).double() for n in ps.names()}
        return o, out, grads
    finally:
        om.set_dtype(torch.float64)


def _same_targets(a, b):
    return a is not None and all(np.array_equal(a[k], b[k]) for k in a)


def _rel(got, want):
    want = want.detach().double()
    return float((got.detach().double().cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-30)


def _run(S, mode, keypoints=False, N=1, seed=11, G=6):
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    m, chain = _model(keypoints)
    b = make_batch(seed, N, S, S, G=G, n_fg_class=1 if keypoints else 80, n_keypoints=17 if keypoints else None)
    b['bboxes'][:, :, 2:] = np.minimum(b['bboxes'][:, :, 2:], [S, S])
    bt = {k: torch.from_numpy(v).to(DEV) for k, v in b.items()}
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(*MODES[mode][0]))
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(*MODES[mode][1]))
    core.LAYER_TILE_HINTS = MODES[mode][2] if len(MODES[mode]) > 2 else True
    core.FWD_EMULATION_BEHIND_BACKBONE = mode != 'bf16x6_backbone_fwd'
    core.FWD_EMULATION_IN_BACKBONE = mode != 'bf16x6_behind_backbone_fwd'
    _hip.check(_hip.lib().mrcnn_conv2d_set_split_operands(*SPLIT.get(mode, (0, 0, 0))))
    try:
        chain.proposal_target_creator.set_seed(21)
        chain.anchor_target_creator.set_seed(22)
        # the step as MomentumSGD.update runs it: the chain knows that a backward pass with d loss = 1 follows and starts the RPN's
        # backward pass early, beside the proposal chain - the summation order of the shipped training step
        chain.backward_follows = chain.unit_upstream = True
        try:
            loss = chain(bt['imgs'], bt['bboxes'], bt['labels'], bt['keypoints' if keypoints else 'masks'], 1.0)
            outs = {k: ([f.clone() for f in v] if k == 'features' else v.clone()) for k, v in chain.outputs.items()}
            loss.backward()
        finally:
            chain.backward_follows = chain.unit_upstream = False
        obs = {k: float(v) for k, v in chain.observation.items()}
        t = _targets(chain)
        img4 = torch.cat([bt['imgs'].cpu().permute(0, 2, 3, 1), torch.zeros((N, S, S, 1))], -1)
        key = ('oracle', S, keypoints, N, seed, G)
        if not (key in _cache and _same_targets(_cache[key]['t'], t)):      # proposals can differ between conv paths
            for k in [k for k in _cache if isinstance(k, tuple) and k[0] == 'oracle']:      # one oracle at a time: ~10 GB each at bs 2, 1024^2
                del _cache[k]
            t0 = time.time()
            o64, out64, g64 = _oracle(m, t, img4, torch.float64, tap=TAP, keypoints=keypoints)
            _, out32, g32 = _oracle(m, t, img4, torch.float32, keypoints=keypoints)
            # a SECOND float32 realisation of the same step (other convolution kernels => other summation order): a single
            # realisation under-estimates the rounding noise of individual tensors by up to 40x (head/fc1/W of the
            # keypoint model: floor 3.7e-4, 3.4e-3 or 1.5e-2 depending on the sampled RoIs of the run)
            with torch.backends.mkldnn.flags(enabled=False):
                _, _, g32b = _oracle(m, t, img4, torch.float32, keypoints=keypoints)
            h1, y2 = o64.taps[TAP + '/conv2']
            _cache[key] = dict(t=t, out=out64, g64=g64, g32=g32, g32b=g32b, tap=(h1.detach(), y2.grad.detach()), secs=time.time() - t0)
        c = _cache[key]
        # ---- (i) activations
        want, acts = c['out'], {}
        for l, (f, w) in enumerate(zip(outs['features'], want['feats'])):
            acts['p%d' % (l + 2)] = _rel(f, w)
        acts['rpn_locs'] = _rel(outs['locs'], want['locs'])
        acts['rpn_scores'] = _rel(outs['scores'], want['scores'])
        nc, l0 = m.head.n_class, m.head.LOC0
        acts['roi_scores'] = _rel(outs['box'][:, :nc], want['box'][:, :nc])
        acts['roi_cls_locs'] = _rel(outs['box'][:, l0:l0 + 4], want['box'][:, l0:l0 + 4])
        mc = m.head.mask_out_channels
        acts['mask'] = _rel(outs['mask'][..., :mc], want['mask'][..., :mc])
        losses = {k: abs(obs[k] - float(want[k].detach())) / max(abs(float(want[k].detach())), 1e-3) for k in NAMES}
        # ---- (ii) gradients against the float64 oracle, with the float32 oracle's own error as the noise floor
        ps = m.ps
        gmax = max(float(g.abs().max()) for g in c['g64'].values())
        rows = []
        for n in ps.names():
            w64 = c['g64'][n]
            scale = max(float(w64.abs().max()), 1e-3 * gmax)
            err = float((ps.g(n).cpu().double() - w64).abs().max()) / scale
            floor = max(float((c['g32'][n] - w64).abs().max()), float((c['g32b'][n] - w64).abs().max())) / scale
            rows.append((n, err, floor))
        # ---- (iii) the worst layer of round 1 in isolation: oracle x, gy -> device filter gradient
        h1, gy2 = c['tap']
        wname = TAP + '/conv2/W'
        gw, _ = hnn.conv2d_bwd_filter_raw(h1.float().contiguous().to(DEV), gy2.float().contiguous().to(DEV),
                                          tuple(ps.p(wname).shape), 1, 1, False)
        iso = _rel(gw, c['g64'][wname])
    finally:
        _hip.check(_hip.lib().mrcnn_conv2d_set_split_operands(0, 0, 0))
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(*DEFAULT[0]))
        _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(*DEFAULT[1]))
        core.LAYER_TILE_HINTS = True
        core.FWD_EMULATION_BEHIND_BACKBONE = True
        core.FWD_EMULATION_IN_BACKBONE = True
    # ---- report
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    errs = sorted(r[1] for r in rows)
    flo = sorted(r[2] for r in rows)
    q = lambda v, f: v[int(f * (len(v) - 1))]
    with open(os.path.join(out_dir, 'full_width_parity_%d_%s%s%s.txt' % (S, mode, '_keypoint' if keypoints else '', '_n%d' % N if N > 1 else '')), 'w') as f:
        f.write('# full ResNet-50-FPN %s R-CNN, %d' % ('Keypoint' if keypoints else 'Mask', N))
        f.write(' %dx%d image(s), conv path %s; oracle float64 + float32 took %.0f s\n' % (S, S, mode, c['secs']))
        f.write('# activations, max |device - fp64 oracle| / max |oracle|: %s\n' % ', '.join('%s %.2e' % kv for kv in acts.items()))
        f.write('# losses, relative: %s\n' % ', '.join('%s %.2e' % kv for kv in losses.items()))
        f.write('# isolated %s filter gradient (oracle x, gy -> mrcnn_conv2d_bwd_filter_f32): %.2e\n' % (wname, iso))
        f.write('# gradient error quantiles  device: median %.2e 90%% %.2e max %.2e | float32 oracle (floor): median %.2e 90%% %.2e max %.2e\n'
                % (q(errs, .5), q(errs, .9), errs[-1], q(flo, .5), q(flo, .9), flo[-1]))
        f.write('# %-48s %10s %10s %8s\n' % ('parameter', 'device', 'fp32floor', 'ratio'))
        for n, e, fl in rows:
            f.write('%-50s %10.3e %10.3e %8.2f\n' % (n, e, fl, e / max(fl, 1e-12)))
    return acts, losses, rows, iso


def _check(S, mode, keypoints=False, iso_tol=2e-5, above3_frac=0.03, **kw):
    acts, losses, rows, iso = _run(S, mode, keypoints, **kw)
    for k, v in acts.items():
        assert v <= 1e-3, ('activation', k, v)          # BASELINE.json north_star: conv activations within 1e-3 relative
    for k, v in losses.items():
        assert v <= 1e-4, ('loss', k, v)
    assert iso <= iso_tol, ('isolated res5/b2/conv2 filter gradient', iso)
    ratios = sorted(e / max(fl, 1e-12) for n, e, fl in rows if e >= 1e-3)
    if mode == 'fast':
        # F(4x4) in the FORWARD pass: activations stay within 1e-3 (above), but their ~2e-4 errors are amplified by the
        # curvature of the losses into gradient errors far above the float32 floor in the layers without BatchNorm
        # (measured: rpn/conv/W 2.1e-2 at 512^2, 1.2e-2 at 1024^2; profiles/r02_winograd_pass_probe.txt).  That is why this
        # mode is opt-in; the bound here only keeps the deviation where it was measured.
        assert max(e for _, e, _ in rows) < 0.5 and all(e < max(1e-3, 12 * fl, 4e-2) for _, e, fl in rows), \
            sorted(rows, key=lambda r: -r[1])[:5]
        return
    # (ii): the float32 oracle's own error is ONE realisation of the rounding noise of this network, and so is the
    # device's: per-tensor ratios of two realisations scatter (ReLU / max-pool decisions flip on values within rounding of
    # a tie) - the direct kernels themselves show up to 5.2x on single tensors.  Bars: every tensor < max(1e-3, 6 x floor),
    # at most 3 % of the tensors above 3 x floor, and the typical tensor AT the floor (median ratio <= 1.3).
    bad = [(n, e, fl) for n, e, fl in rows if not e < max(1e-3, 6 * fl)]
    assert not bad, bad[:10]
    above3 = [r for r in rows if not r[1] < max(1e-3, 3 * r[2])]
    assert len(above3) <= above3_frac * len(rows), above3[:10]
    assert ratios[len(ratios) // 2] <= 1.3, ratios[len(ratios) // 2]


@pytest.mark.parametrize('mode', [pytest.param('direct', marks=pytest.mark.gpu_long), pytest.param('winograd_f2', marks=pytest.mark.gpu_long),
                                  pytest.param('uniform_f2_forward', marks=pytest.mark.gpu_long), 'shipped', pytest.param('fast', marks=pytest.mark.gpu_long)])
def test_full_width_512(mode):
    _check(512, mode)


@pytest.mark.gpu_long
def test_full_width_1024_shipped():
    """BASELINE.json configs[2]'s image size with the shipped (benchmarked) kernel selection."""
    _check(1024, 'shipped')


@pytest.mark.gpu_long
def test_full_width_1024_batch2_shipped():
    """The benchmarked configuration itself: BASELINE.json configs[2] = TWO 1024x1024 images (bench.py's
    batch: make_batch(100, 2, 1024, 1024, G=8)), full width, shipped kernel selection, mask branch on all 256 rows - per-image
    target blocks, roi_indices, BatchNorm statistics over two images and the concatenated-batch loss normalisers at full
    size, against the float64 oracle with the same bars as the one-image tests."""
    _check(1024, 'shipped', N=2, seed=100, G=8)        # (the oracle stays cached for the split-bf16 variant of the same batch below)


@pytest.mark.gpu_long
def test_full_width_512_split_bf16_backward_opt_in():
    """EXPLORATORY opt-in, validated by the SAME bars as the shipped float32 configuration: float32 MFMA in the
    forward pass, three-term split-bf16 operands on the bf16 MFMA in BOTH backward passes (every backward-data and
    backward-filter GEMM, Winograd or not).  The forward pass is the shipped one bit for bit (same activations, losses,
    sampled targets); every parameter gradient is held to the float32 noise floor like the float32 kernels."""
    _check(512, 'split_bf16_bwd_only')      # (this layer takes the direct kernel at 512^2: 9.5e-6)


@pytest.mark.gpu_long
def test_full_width_1024_batch2_split_bf16_backward_opt_in():
    """The same on the benchmarked configuration (two 1024x1024 images, bench.py's batch)."""
    # the isolated res5 filter gradient takes the F(4x4) Winograd path here: the transforms amplify the bf16 planes' 4e-6 to 2.7e-4
    # of the tensor scale (float32 MFMA: 8e-6) - two orders below the float32 noise floor of the gradients it is part of
    _check(1024, 'split_bf16_bwd_only', N=2, seed=100, G=8, iso_tol=1e-3)


# The SHIPPED training arithmetic (model/fpn_maskrcnn_train_chain.py DEFAULT_GEMM_ARITHMETIC = 'bf16x6_behind_backbone', what train.py runs
# and bench.py reports as `value`): the float32-accurate three-plane emulation on the bf16 MFMA in both backward passes of every layer and
# in the forward pass of every layer behind the backbone; float32 MFMA in the forward pass of the ResNet's convolutions.
HEADLINE = 'bf16x6_behind_backbone_fwd'


def test_full_width_1024_batch2_shipped_arithmetic():
    """The benchmarked batch (bench.py: make_batch(100, 2, 1024, 1024, G=8)) in the shipped arithmetic with EVERY bar of the float32
    configuration unrelaxed: activations <= 1e-3, losses <= 1e-4, every gradient tensor < max(1e-3, 6 x floor),
    at most 3 % of the tensors above 3 x floor, median ratio <= 1.3, isolated filter gradient <= 2e-5."""
    _check(1024, HEADLINE, N=2, seed=100, G=8)


@pytest.mark.gpu_long
def test_full_width_1024_batch2_emulation_in_the_backward_passes_only():
    """The more conservative arithmetic 'bf16x6_backward' (float32 MFMA in the whole forward pass: activations, losses and sampled
    targets bit-identical to the float32 step), same batch, same unrelaxed bars."""
    _check(1024, 'bf16x6_bwd_only', N=2, seed=100, G=8)


@pytest.mark.gpu_long
def test_full_width_1024_batch2_float32_accurate_emulation_in_every_pass_opt_in():
    """OPT-IN, not the shipped arithmetic: bf16x6 in the forward pass of the BACKBONE too.  On THIS batch the emulated ResNet forward is
    another realisation of the rounding noise (50 layers of training-mode BatchNorm) in which one near-tie decision downstream falls the
    other way: the tensors behind it (toplayer, lat_p2..p4, conv_p3, rpn/conv/b) sit at 3.4 - 4.7 x the floor, 4 - 5 % of the tensors
    against the 3 % bar (emulating ONLY the backbone's forward pass reproduces it, emulating only the forward pass behind the backbone
    does not: profiles/r04_full_width_parity_*backbone*.txt) - which is why the backbone's forward pass stays on the float32 MFMA.  On four other batches (seeds
    101 .. 104) this mode passes the 3 % bar too (profiles/r04_full_width_parity_five_seeds.txt); here the bar is 6 % and everything
    else (activations, losses, every tensor < 6 x floor, median, isolated filter gradient) is held as for the float32 configuration."""
    _check(1024, 'bf16x6', N=2, seed=100, G=8, above3_frac=0.06)
    _cache.pop(('oracle', 1024, False, 2, 100, 8), None)       # ~10 GB of float64 gradients and activations


@pytest.mark.gpu_long
@pytest.mark.parametrize('seed', [101, 102, 103, 104])
def test_full_width_1024_batch2_shipped_arithmetic_other_batches(seed):
    """Four more batches of the benchmarked shape, shipped arithmetic, the same unrelaxed bars (one float64 + two float32 oracle
    evaluations per batch, ~2 minutes each on the GPU box's host cores)."""
    try:
        _check(1024, HEADLINE, N=2, seed=seed, G=8)
    finally:
        _cache.pop(('oracle', 1024, False, 2, seed, 8), None)


@pytest.mark.gpu_long
@pytest.mark.parametrize('mode', ['bf16x6', 'bf16x6_fwd'])
def test_full_width_512_float32_accurate_emulation_opt_in(mode):
    """EXPLORATORY opt-in: three bf16 planes per operand (hi + mid + lo = the float32 value exactly) and the six products of weight
    >= 2^-16 - a float32-ACCURATE GEMM on the bf16 MFMA (the dropped terms are of the size of the float32 MFMA's own accumulation
    rounding) - in every pass ('bf16x6'), or in the forward pass with the two-plane bf16 split in the backward passes
    ('bf16x6_fwd').  Held to ALL the bars of the float32 configuration."""
    _check(512, mode)


@pytest.mark.gpu_long
def test_full_width_512_split_half_forward_opt_in():
    """EXPLORATORY opt-in, the fastest combination: half hi / lo planes in the forward pass (22 significant bits, weights scaled by
    2^12 into half's normal range), bf16 planes in the backward passes.  Activations <= 1e-3 and losses <= 1e-4 like the shipped
    configuration (measured: activations at the float32 kernels' own 6e-5 .. 1.3e-4); gradients: the typical tensor AT the
    float32 noise floor (median ratio <= 1.3) and no tensor beyond 12 x floor - the float32 bar is 6 x: the forward products are
    accurate to 5e-7 instead of 6e-8, so a few more ReLU / max-pool decisions on near-ties flip (measured: 2 of 180 tensors
    between 6 x and 7.6 x, profiles/r03_full_width_parity_512_split_f16_fwd.txt)."""
    acts, losses, rows, iso = _run(512, 'split_f16_fwd')
    for k, v in acts.items():
        assert v <= 1e-3, ('activation', k, v)
    for k, v in losses.items():
        assert v <= 1e-4, ('loss', k, v)
    assert iso <= 2e-5, iso
    assert not [(n, e, fl) for n, e, fl in rows if not e < max(1e-3, 12 * fl)]
    ratios = sorted(e / max(fl, 1e-12) for n, e, fl in rows if e >= 1e-3)
    assert ratios[len(ratios) // 2] <= 1.3, ratios[len(ratios) // 2]


def test_full_width_keypoint_1024_batch2_shipped():
    """BASELINE.json configs[4]'s per-GPU shape: the Keypoint R-CNN of train_keypoints.py, TWO 1024x1024 images,
    full width, bench.py's batch (make_batch(100, ...)), the SHIPPED arithmetic: the bars of the mask test."""
    _check(1024, HEADLINE, keypoints=True, N=2, seed=100, G=8)


@pytest.mark.gpu_long
def test_full_width_keypoint_1024_batch2_float32_kernels():
    """The same batch on the float32 MFMA kernels (its own oracle evaluation: the sampled RoIs differ from the emulated step's)."""
    _check(1024, 'shipped', keypoints=True, N=2, seed=100, G=8)


@pytest.mark.gpu_long
def test_full_width_keypoint_512_shipped():
    """BASELINE.json configs[4]'s model at full width (train_keypoints.py: 1 class, 17 keypoints, 8 keypoint convolutions,
    56x56 heat maps, softmax cross-entropy over positions), one 512x512 image, shipped kernel selection: the same bars."""
    _check(512, 'shipped', keypoints=True)
