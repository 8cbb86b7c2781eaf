"""bench.py workload 'step': BASELINE.json configs[2] - ResNet50-FPN Mask R-CNN full forward + backward +
optimizer update, batch 2 per GPU, 1024x1024, fp32, synthetic COCO-shaped data resident in HBM."""
import json
import os
import time

import numpy as np
import torch

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak (no sparsity)


def bench_step(args, rank, world):
    from chainer_maskrcnn.model.maskrcnn import MaskRCNN
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import (FPNMaskRCNNTrainChain, calc_mask_loss, calc_keypoint_loss, GEMM_ARITHMETIC,
                                                                 DEFAULT_GEMM_ARITHMETIC)
    from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
    from chainer_maskrcnn.utils.synthetic import make_batch
    from chainer_maskrcnn._hip import nn as hnn
    dev = torch.device('cuda', 0 if os.environ.get('MRCNN_BENCH_SINGLE_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    N, H, W = 2, 1024, 1024
    mask_rows = getattr(args, 'mask_rows', 'all')
    tiles = os.environ.get('MRCNN_WINO_PASS_TILES')       # e.g. "2,0,0": forward F(2x2) (strict gradient parity), measurement knob
    if tiles:
        from chainer_maskrcnn._hip import lib, check
        check(lib().mrcnn_conv2d_set_winograd_pass_tiles(*[int(v) for v in tiles.split(',')]))
    keypoints = bool(getattr(args, 'keypoints', False))
    arith = getattr(args, 'gemm_arithmetic', None) or DEFAULT_GEMM_ARITHMETIC        # the shipped training default (train.py)
    if keypoints:       # BASELINE.json configs[4] per-GPU shape: train_keypoints.py's model (1 class, 17 keypoints, 8 mask convs, 56x56 maps)
        model = MaskRCNN(n_fg_class=1, n_keypoints=17, head_arch='fpn_keypoint', device=dev, seed=1234)
        chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_keypoint_loss, binary_mask=False, mask_rows=mask_rows, gemm_arithmetic=arith)
    else:
        model = MaskRCNN(n_fg_class=80, device=dev, seed=1234)
        chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows=mask_rows, gemm_arithmetic=arith)
    opt = MomentumSGD(lr=1e-3, momentum=0.9).setup(chain)
    opt.add_hook(WeightDecay(0.0005))
    rccl = None
    if world > 1:
        from chainer_maskrcnn.optimizers import rccl_evidence
        rccl = rccl_evidence(dev, world)      # raises on every rank when two ranks share a device (and the functional-check switch is off)
        opt.enable_data_parallel()
    b = make_batch(100 + rank, N, H, W, G=8, n_fg_class=1 if keypoints else 80, n_keypoints=17 if keypoints else None)
    imgs, bb, lab, masks = (torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'keypoints' if keypoints else 'masks'))

    graphed = None
    mode = 'eager launches'
    if world == 1 and getattr(args, 'graph', 0):
        try:
            from chainer_maskrcnn.optimizers import GraphedStep
            graphed = GraphedStep(opt, chain, [imgs, bb, lab, masks], 1.0)
            mode = 'one HIP graph per step (forward + backward + SGD captured once, replayed)'
        except Exception as e:      # capture is an optimisation: report and measure the eager path instead
            graphed = None
            mode = 'eager launches (graph capture failed: %s)' % str(e).split('\n')[0][:120]
            torch.cuda.synchronize()

    def step():
        if graphed is not None:
            return graphed(imgs, bb, lab, masks)
        return opt.update(chain, imgs, bb, lab, masks, 1.0)

    for _ in range(args.warmup):
        step()
    _sync(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _sync(world)
    dt = time.perf_counter() - t0
    per_rank_ms = _gather_over_ranks(dt / args.steps * 1e3, world, dev)
    dt = _max_over_ranks(dt, world, dev)
    loss = float(chain.observation['loss'])
    ips = N * world * args.steps / dt

    dp_report = None
    if world > 1:       # two more steps with HIP events around every bucket's all-reduce: is the exchange hidden under backward?
        opt.sync.timing = True
        for _ in range(2):
            step()
        _sync(world)
        dp_report = opt.sync.timing_report()
        opt.sync.timing = False

    def timed(n_warm=2, n=5):
        for _ in range(n_warm):
            opt.update(chain, imgs, bb, lab, masks, 1.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            opt.update(chain, imgs, bb, lab, masks, 1.0)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        if world > 1:       # every rank runs the same sequence (the gradient exchange is inside update): whole-job rate over the slowest rank
            el = _max_over_ranks(el, world, dev)
        return N * world * n / el

    # the same step with the mask branch on the positive rows only (identical loss and gradients)
    alt = None
    if mask_rows == 'all' and world == 1:
        chain.mask_rows = 'positives'
        alt = timed()
        chain.mask_rows = 'all'

    # opt-in mode: F(4x4) Winograd in the forward pass of the ResNet conv2 layers too (activations still <= 2.2e-4 of
    # scale, but the gradients of the layers fed by c4 / c5 move by up to 2e-2: profiles/r02_winograd_layer_probe.txt) -
    # reported, never `value`
    fast = None
    if world == 1 and not tiles:
        from chainer_maskrcnn._hip import lib, check
        check(lib().mrcnn_conv2d_set_winograd_pass_tiles(0, 0, 0))
        try:
            fast = timed()
        finally:
            check(lib().mrcnn_conv2d_set_winograd_pass_tiles(2, 0, 0))

    # The same step under the other GEMM arithmetics, same process (reported beside `value`, never as `value`): ALWAYS the all-float32
    # MFMA step (what a reader who does not accept the float32-accurate emulation falls back on), the emulation in every pass, and two
    # NARROWER schemes (two-plane splits: 16 / 22 operand bits) that are opt-in lines forever.  A failure of an opt-in mode is
    # reported in the line, it does not lose the line.
    other = {}
    if not tiles:
        from chainer_maskrcnn._hip import lib, check
        from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import select_gemm_arithmetic
        keep = chain.gemm_arithmetic
        chain.gemm_arithmetic = None
        # N > 1: only the float32-MFMA step (the line's fallback number); it is a collective sequence, so a failure is not caught there
        names = ('f32', 'bf16x6_behind_backbone', 'bf16x6_backward', 'bf16x6', 'split_bf16_backward', 'split_half_forward_bf16_backward') if world == 1 else ('f32',)
        for name in names:
            if name == arith:
                continue
            try:
                if name in GEMM_ARITHMETIC:
                    select_gemm_arithmetic(name)
                else:
                    select_gemm_arithmetic('f32')
                    check(lib().mrcnn_conv2d_set_split_operands(*{'split_bf16_backward': (0, 1, 1), 'split_half_forward_bf16_backward': (2, 1, 1)}[name]))
                other[name] = round(timed(), 3)
            except Exception as e:
                if world > 1:
                    raise
                other[name] = 'failed: %s' % str(e).split('\n')[0][:160]
            finally:
                select_gemm_arithmetic(arith)
        chain.gemm_arithmetic = keep
        opt.update(chain, imgs, bb, lab, masks, 1.0)

    # roofline of the dominant kernel family (k_conv_igemm): instrumented steps, HIP events around every launch
    hnn.PROFILE = []
    chain.use_aux_stream = False      # instrumented steps: one stream, every conv launch bracketed by events
    n_prof = 2
    for _ in range(n_prof):
        opt.update(chain, imgs, bb, lab, masks, 1.0)      # eager, single stream: every launch bracketed by events
    torch.cuda.synchronize()
    recs, hnn.PROFILE = hnn.PROFILE, None
    chain.use_aux_stream = True
    agg = {}
    for rec in recs:
        kind, macs, e0, e1, _shape, executed = rec[:6]
        a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += 2.0 * macs
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += 2.0 * executed
    flops = sum(a[1] for a in agg.values()) / n_prof
    exe_flops = sum(a[3] for a in agg.values()) / n_prof
    secs = sum(a[2] for a in agg.values()) / n_prof
    launches = sum(a[0] for a in agg.values()) // n_prof
    split = _replay_split(recs, n_prof, dev)
    pmc = _pmc_step_counters()
    smode = GEMM_ARITHMETIC[arith][0]
    gk = split['gemm_by_kind']          # keys: (pass, 'f32' | 'bf16x6' | ...) -> ms and executed flops of its GEMM launches

    def pipe(which, mult, peak):
        keys = [k for k in gk if k.endswith('/' + which)]
        fl = sum(gk[k]['executed_flops'] for k in keys) * mult
        ms = sum(gk[k]['ms_per_step'] for k in keys)
        return {'passes': sorted(k.split('/')[0] for k in keys), 'executed_TFLOPs': round(fl / (ms * 1e-3) / 1e12, 2) if ms else None, 'peak': peak,
                'frac': round(fl / (ms * 1e-3) / 1e12 / peak, 4) if ms else None, 'gemm_ms_per_step': round(ms, 3), 'executed_flops_per_step': fl}
    emu_kinds = [k for k in gk if k.endswith('/bf16x6')]
    f32_kinds = [k for k in gk if k.endswith('/f32')]
    # The dominant kernel = the GEMM launches of the convolution calls.  Each pass is priced on the pipe it runs on: a float32-MFMA
    # pass executes 2 flops per MAC on v_mfma_f32_32x32x2_f32 (peak 157.3 TF/s), an emulated pass SIX bf16 MFMA products per MAC on
    # v_mfma_f32_32x32x16_bf16 (peak 2500 TF/s dense).  `roofline` is the larger of the two groups by time.
    emu = pipe('bf16x6', 6.0, MFMA_BF16_PEAK_TFLOPS) if emu_kinds else None
    f32 = pipe('f32', 1.0, MFMA_F32_PEAK_TFLOPS) if f32_kinds else None
    head = emu if (emu and (not f32 or emu['gemm_ms_per_step'] >= f32['gemm_ms_per_step'])) else f32
    head_is_emu = head is emu
    gemm_ms = sum(v['ms_per_step'] for v in gk.values())
    out = {
        'metric': 'images/sec (1024^2 COCO, bs=2/GPU) at 1/2/4/8 MI355X; ROIAlign bwd HBM GB/s',
        'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': ('configs[4] per-GPU shape: Keypoint R-CNN (train_keypoints.py: 1 class, 17 keypoints, 8 mask convs, 56x56 heat '
                                'maps) full fwd+bwd+SGD step, batch 2/GPU, 1024x1024, fp32, 8 gt boxes/image, keypoint branch on %s rows'
                                if keypoints else
                                'configs[2] ResNet50-FPN Mask R-CNN full fwd+bwd+SGD step, batch 2/GPU, 1024x1024, fp32, '
                                '8 gt boxes/image, 2000 proposals -> 256 sampled RoIs/image, mask branch on %s rows')
                               % ('the <=64 positive' if mask_rows == 'positives' else 'all 256 sampled'),
                   'global_batch': N * world, 'launch_mode': mode, 'parallelism': 'dp%d: RCCL all-reduce of the flat gradient buffer in 25 MB '
                   'buckets on a side stream' % world if world > 1 else 'single GPU', 'final_loss': round(loss, 4),
                   'box_to_box_note': 'identical code measures 20.5 - 21.9 ms per step and a float32-GEMM fraction of 0.60 - 0.68 on different MI355X boxes '
                                      '(device / DVFS variance, MI355X_MICROARCH.md "DVFS give-back" item 5): compare rounds by same-box A/B (profiles/r06_ab_*), not by this line alone',
                   'gemm_arithmetic': {
                       'name': arith, 'split_operands_fwd_bwddata_bwdfilter': list(smode),
                       'scheme': {'bf16x6_behind_backbone': 'float32-ACCURATE emulation on v_mfma_f32_32x32x16_bf16 - every float32 operand is carried EXACTLY by three bf16 '
                                                            'planes hi + mid + lo, the six products of weight >= 2^-16 (lh + hl + mm + mh + hm + hh) are accumulated in float32; the '
                                                            'dropped terms are <= 3 x 2^-24 |ab|, the size of the float32 MFMA\'s own accumulation rounding - in the backward-data and '
                                                            'backward-filter passes of every layer and in the forward pass of every layer BEHIND the backbone (FPN, RPN, heads); the '
                                                            'forward pass of the ResNet-50\'s convolutions runs on v_mfma_f32_32x32x2_f32 (c2 .. c5 are those of the all-float32 step '
                                                            'bit for bit)',
                                  'f32': 'v_mfma_f32_32x32x2_f32 in every pass: float32 operands, float32 accumulate (bit for bit an fmaf chain)',
                                  'bf16x6_backward': 'forward pass: v_mfma_f32_32x32x2_f32 on float32 operands (activations, losses and sampled targets are '
                                                     'those of the all-float32 step bit for bit); backward-data and backward-filter passes: float32-ACCURATE '
                                                     'emulation on v_mfma_f32_32x32x16_bf16 - every float32 operand is carried EXACTLY by three bf16 planes '
                                                     'hi + mid + lo, the six products of weight >= 2^-16 (lh + hl + mm + mh + hm + hh) are accumulated in float32; '
                                                     'the dropped terms are <= 3 x 2^-24 |ab|, the size of the float32 MFMA\'s own accumulation rounding',
                                  'bf16x6': 'the three-plane / six-product float32-accurate emulation on v_mfma_f32_32x32x16_bf16 in every pass'}[arith],
                       'tensors': 'float32 in HBM everywhere; float32 accumulators; no tensor is stored in 16 bits',
                       'why_this_is_the_value': 'VERDICT r3 ruling: the three-plane emulation is not narrower than the float32 MFMA (per-GEMM error against '
                                                'float64 <= the float32 kernel\'s: tests/test_split_gemm_gpu.py); it carries the headline in the passes where the '
                                                'full-width parity bars of the float32 configuration hold UNRELAXED on five batches '
                                                '(tests/test_full_width_gpu.py, profiles/r04_full_width_parity_*.txt): both backward passes, and the forward pass '
                                                'behind the backbone.  With the backbone\'s forward pass emulated too, one of the five batches has 4 % of the gradient '
                                                'tensors above 3 x the float32 floor (bar 3 %): that configuration stays an opt-in line'},
                   'images_per_sec_f32_mfma': other.get('f32') if arith != 'f32' else round(ips, 3)},
        # The GEMM launches of the headline pass group on the pipe they run on (see above); the whole convolution bracket (GEMMs +
        # Winograd transforms + sums) and the float32 group follow as extra keys.
        'roofline': {'bound': 'mfma',
                     'kernel': ('k_conv_igemm<..., split 3> + k_pgemm_pp / k_pgemm_gpp (plane GEMMs): the GEMM launches of every emulated convolution call (%s)'
                                % ', '.join(head['passes'])) if head_is_emu else
                               'k_conv_igemm<..., float32>: the GEMM launches of the float32-MFMA calls (%s)' % ', '.join(head['passes']),
                     'achieved': head['executed_TFLOPs'], 'peak': head['peak'], 'unit': 'TFLOP/s', 'frac': head['frac'],
                     'pipe': 'v_mfma_f32_32x32x16_bf16, dense bf16 peak; executed flops = 6 products x 2 x MACs the pipes execute' if head_is_emu else
                             'v_mfma_f32_32x32x2_f32, float32 MFMA peak; executed flops = 2 x MACs the pipes execute',
                     'gemm_ms_per_step': head['gemm_ms_per_step'], 'executed_flops_per_step': head['executed_flops_per_step'],
                     'effective_fp32_TFLOPs': round(head['executed_flops_per_step'] / (6.0 if head_is_emu else 1.0) / (head['gemm_ms_per_step'] * 1e-3) / 1e12, 2)
                     if head['gemm_ms_per_step'] else None,
                     # the same rate against the float32 MFMA peak (157.3 TFLOP/s), whatever pipe the headline group runs on: comparable
                     # across rounds and arithmetics
                     'frac_f32_equivalent': round(head['executed_flops_per_step'] / (6.0 if head_is_emu else 1.0) / (head['gemm_ms_per_step'] * 1e-3) / 157.3e12, 4)
                     if head['gemm_ms_per_step'] else None,
                     'in_step': _in_step_gemm(f32, emu) if not keypoints else None,      # (the committed trace is of the configs[2] step)
                     'traffic': pmc['traffic'], 'traffic_source': pmc['traffic_source'],
                     'hbm_bytes_per_step_by_family': pmc.get('hbm_bytes_per_step_by_family'),
                     'hbm_bytes_per_step_whole_step': pmc.get('hbm_bytes_per_step_whole_step'),
                     'pmc_mfma_busy_fraction_by_kernel': pmc.get('pmc_mfma_busy_fraction_by_kernel'),
                     'pmc_mfma_source': pmc.get('pmc_mfma_source'),
                     'gemm_kernels_only': {'ms_per_step_all_passes': round(gemm_ms, 3), 'by_pass_and_arithmetic': gk,
                                           'float32_mfma_calls': f32, 'bf16_emulated_calls': emu},
                     'conv_bracket': {'ms_per_step': round(secs * 1e3, 3), 'launches': launches,
                                      'executed_macs_x2_per_step': exe_flops, 'algorithmic_flops_per_step': flops,
                                      'effective_fp32_TFLOPs': round(flops / secs / 1e12, 3),
                                      'by_kind': {k: {'launches': a[0] // n_prof, 'effective_fp32_TFLOPs': round(a[1] / a[2] / 1e12, 3),
                                                      'ms': round(a[2] / n_prof * 1e3, 3)} for k, a in agg.items()}},
                     'note': 'gemm_kernels_only: every distinct convolution call of the step replayed standalone with '
                             'mrcnn_conv2d_set_debug_skip(2) (GEMM launches only), HIP events, times weighted by the call counts of %d instrumented '
                             'single-stream steps that ran right after the timed region; conv_bracket: HIP events around every conv call of those '
                             'steps (GEMMs + Winograd transforms + slab / column sums); effective_fp32_TFLOPs = 2 x MACs of the direct algorithm on '
                             'un-padded channels (conv_bracket) or 2 x executed MACs (roofline) per second - a rate in float32-equivalent work, '
                             'not a fraction of any peak' % n_prof},
        'roofline_winograd_transforms': dict(split['aux'], traffic=(lambda fam: (fam['winograd_transforms'] + fam['slab_tail_column_sums'])
                                                                     if fam and 'winograd_transforms' in fam and 'slab_tail_column_sums' in fam else None)(pmc.get('hbm_bytes_per_step_by_family'))),
    }
    if alt is not None:
        out['config']['images_per_sec_mask_branch_on_positive_rows_only'] = round(alt, 3)
    if fast is not None:
        out['config']['images_per_sec_opt_in_winograd_f4_forward'] = round(fast, 3)
    if other:
        out['config']['other_gemm_arithmetics_same_process'] = {
            'images_per_sec': other,
            'note': 'mrcnn_conv2d_set_split_operands(forward, backward-data, backward-filter): f32 = (0,0,0); bf16x6 = (3,3,3) the float32-accurate '
                    'emulation in every pass (opt-in, see gemm_arithmetic.why_this_is_the_value); bf16x6_backward = (0,3,3); split_bf16_backward = (0,1,1) '
                    'and split_half_forward_bf16_backward = (2,1,1) are NARROWER two-plane splits (16 / 22 operand bits): opt-in lines, never `value`'}
    out['config']['winograd_tiles_fwd_bwddata_bwdfilter'] = tiles or ('2,0,0 (0 = F(4x4) where the layer is large enough, else F(2x2)); the FPN / '
                                                                      'RPN / head convolutions run their forward pass with 0')
    if dp_report is not None:
        out['config']['allreduce_rank0'] = dp_report
    if rccl is not None:
        out['config']['rccl'] = rccl
    if world > 1:       # a straggler shows here: `ms_per_step` is the MAX over ranks (the contract), this is every rank's own clock
        out['config']['ms_per_step_per_rank'] = {'min': round(min(per_rank_ms), 3), 'max': round(max(per_rank_ms), 3),
                                                 'all': [round(v, 3) for v in per_rank_ms]}
    if getattr(args, 'cpu_affinity', None) is not None:
        out['config']['cpu_affinity_rank0'] = args.cpu_affinity
    return out, model, dev


def _sync(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()


def _max_over_ranks(v, world, dev):
    if world == 1:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def latest_profile(suffix):
    """(path relative to the repo root, parsed JSON) of the newest profiles/rNN_<suffix> (highest round), or (None, None)."""
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import glob
    files = sorted(glob.glob(os.path.join(root, 'profiles', 'r[0-9][0-9]_' + suffix)))
    for f in reversed(files):
        try:
            return os.path.relpath(f, root), json.load(open(f))
        except Exception:
            continue
    return None, None


def _in_step_gemm(f32, emu):
    """roofline.in_step: the GEMM launches' durations INSIDE the multi-stream step - from the committed kernel trace
    summary (tools/trace_in_step_gemm.py; a trace needs rocprofv3, so this is a constant of the repository like `traffic`) - and the
    blended fraction of the two pipes' peaks: (float32 flops / 157.3 TF + bf16 product flops / 2500 TF) / that time."""
    f, d = latest_profile('step_in_step_gemm.json')
    if d is None:
        return None
    try:
        t = (d['f32_gemm_ms'] + d['emulated_gemm_ms']) * 1e-3
        ideal = (f32 or {}).get('executed_flops_per_step', 0.0) / 157.3e12 + (emu or {}).get('executed_flops_per_step', 0.0) / 2500e12
        return {'gemm_ms_per_step_in_step': round(t * 1e3, 3), 'f32_gemm_ms': d['f32_gemm_ms'], 'emulated_gemm_ms': d['emulated_gemm_ms'],
                'blended_frac_of_pipe_peaks': round(ideal / t, 4) if t > 0 else None, 'gemm_launches_per_step': d.get('gemm_launches'),
                'kernels_per_step': d.get('kernels'), 'profiled_step_wall_ms': d.get('wall_ms'),
                'source': {'file': f, 'collected_at_commit': d.get('_commit'),
                           'kind': 'constant read from the committed rocprofv3 --kernel-trace summary, not collected by this run'}}
    except Exception:
        return None


def _gather_over_ranks(v, world, dev):
    if world == 1:
        return [v]
    t = torch.tensor([v], dtype=torch.float64, device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu')
    out = [torch.zeros_like(t) for _ in range(world)]
    torch.distributed.all_gather(out, t)
    return [float(o.item()) for o in out]


def _pmc_step_counters():
    """What the hardware counters said about this step - NOT measured by this run: counters need rocprofv3 (separate --pmc
    passes, tools/round_end.sh), so the numbers are read from the summaries committed under profiles/ and the line says
    which file and which commit of the kernels they were collected at (`traffic_source`).  A summary that lacks a key gives
    None fields, never an exception."""
    out = {'traffic': None, 'traffic_source': None}
    try:
        f, d = latest_profile('step_pmc_traffic.json')
        if d is not None:
            out['traffic'] = d['conv_bracket']['hbm_bytes_per_step']
            fam = {k: v['hbm_bytes_per_step'] for k, v in d.items() if isinstance(v, dict) and 'hbm_bytes_per_step' in v and k != 'conv_bracket'}
            out['traffic_source'] = {'file': f, 'collected_at_commit': d.get('_commit'), 'kind': 'constant read from the committed rocprofv3 --pmc summary, '
                                     'not collected by this run', 'method': d.get('_method')}
            out['hbm_bytes_per_step_by_family'] = fam
            out['hbm_bytes_per_step_whole_step'] = sum(fam.values())
    except Exception as e:
        out['traffic_source'] = {'error': 'unreadable step_pmc_traffic summary: %s' % str(e)[:120]}
    try:
        g, m = latest_profile('conv_pmc_mfma.json')
        if m is not None:
            ks = {k: v for k, v in m.get('kernels', {}).items() if k.startswith('k_conv_igemm') or k.startswith('k_pgemm')}
            out['pmc_mfma_busy_fraction_by_kernel'] = {k: v.get('mfma_busy_fraction') for k, v in ks.items()}
            out['pmc_mfma_source'] = {'file': g, 'collected_at_commit': m.get('_commit')}
    except Exception as e:
        out['pmc_mfma_source'] = {'error': 'unreadable conv_pmc_mfma summary: %s' % str(e)[:120]}
    return out


def _replay_split(recs, n_prof, dev):
    """Every distinct convolution call of the step replayed standalone (random operands, 3 repetitions, HIP events) with the
    library's measurement knob, under the GEMM arithmetic in force: GEMM launches only (per pass), and everything but the GEMM
    launches (Winograd transforms, slab / tail / column sums).  Returns {'gemm_by_kind', 'aux'}."""
    from chainer_maskrcnn._hip import nn as hnn, lib, check
    HBM_PEAK = 8000.0
    geoms = {}
    for rec in recs:
        key = (rec[0], rec[6], rec[7], rec[8] if len(rec) > 8 else (0, 0, 0))
        geoms[key] = geoms.get(key, 0) + 1
    gemm_s, exe_k = {}, {}
    aux_s, aux_bytes = 0.0, 0.0
    keep = hnn.PROFILE
    hnn.PROFILE = None
    base = hnn.winograd_pass_tiles()
    base_split = hnn.split_operands()
    names = {0: 'f32', 1: 'bf16x3', 2: 'f16x3', 3: 'bf16x6'}
    try:
        for (kind, g, tiles, sm), cnt in geoms.items():
            cnt = cnt / n_prof
            hnn.set_winograd_pass_tiles(*tiles)         # the call's own tiles (FPN / RPN / head layers bracket theirs)
            check(lib().mrcnn_conv2d_set_split_operands(*sm))        # ... and its own arithmetic (the backbone's forward pass may differ)
            kk = '%s/%s' % (kind, names[sm[{'fwd': 0, 'bwd_data': 1, 'bwd_filter': 2}[kind]]])
            N, H, W, Cin, Cout, KH, KW, stride, pad = g
            Ho, Wo = hnn.conv_out(H, KH, stride, pad), hnn.conv_out(W, KW, stride, pad)
            x = torch.empty((N, H, W, Cin), device=dev).normal_()
            w = torch.empty((Cout, KH, KW, Cin), device=dev).normal_()
            gy = torch.empty((N, Ho, Wo, Cout), device=dev).normal_()
            v = hnn.conv2d_fwd_raw(x, w, None, stride, pad, False, keep_v=True)[1] if kind == 'bwd_filter' else None
            fn = {'fwd': lambda: hnn.conv2d_fwd_raw(x, w, None, stride, pad, False, keep_v=True),
                  'bwd_data': lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), stride, pad),
                  'bwd_filter': lambda: hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), stride, pad, False, wino_v=v)}[kind]
            for mask in (2, 1):           # 2: GEMMs only, 1: the rest
                check(lib().mrcnn_conv2d_set_debug_skip(mask))
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / 3 * 1e-3 * cnt
                if mask == 2:
                    gemm_s[kk] = gemm_s.get(kk, 0.0) + t
                else:
                    aux_s += t
            exe_k[kk] = exe_k.get(kk, 0.0) + 2.0 * lib().mrcnn_conv2d_executed_macs(*g, {'fwd': 0, 'bwd_data': 1, 'bwd_filter': 2}[kind]) * cnt
            vb = lib().mrcnn_conv2d_winograd_v_bytes(*g)
            if vb:          # transforms stream: activation in + transformed operand out, GEMM result in + activation out
                wb = lib().mrcnn_conv2d_winograd_w_bytes(*g)
                ain, aout = 4.0 * N * H * W * Cin, 4.0 * N * H * W * Cout
                aux_bytes += cnt * ({'fwd': ain + vb + wb + aout, 'bwd_data': aout + wb + vb + ain, 'bwd_filter': aout + wb}[kind])
    finally:
        check(lib().mrcnn_conv2d_set_debug_skip(0))
        check(lib().mrcnn_conv2d_set_split_operands(*base_split))
        hnn.set_winograd_pass_tiles(*base)
        hnn.PROFILE = keep
    by_kind = {k: {'ms_per_step': round(gemm_s[k] * 1e3, 3), 'executed_flops': exe_k[k]} for k in gemm_s}
    aux = {'bound': 'hbm', 'kernel': 'k_wino_input / k_wino_output / k_wino_gy / k_wino_filter* + slab, tail and column sums',
           'achieved': round(aux_bytes / aux_s / 1e9, 1), 'peak': HBM_PEAK, 'unit': 'GB/s',
           'frac': round(aux_bytes / aux_s / 1e9 / HBM_PEAK, 4), 'traffic': None, 'ms_per_step': round(aux_s * 1e3, 3),
           'algorithmic_bytes_per_step': aux_bytes,
           'note': 'bytes = activations + transformed operands (V, M / W) of the Winograd calls, each crossing HBM once (float32 sizes; the W operand '
                   'of the large filter-gradient GEMMs is written as three bf16 planes, 6 B per element, which this count prices at 4); the time '
                   'also contains the split-K slab sums and bias column sums of the direct layers; traffic = HBM bytes per step of the transform + '
                   'sum kernel families from the committed rocprofv3 --pmc summary (roofline.traffic_source)'}
    return {'gemm_by_kind': by_kind, 'aux': aux}
