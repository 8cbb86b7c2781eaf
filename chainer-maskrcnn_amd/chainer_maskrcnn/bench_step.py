"""bench.py workload 'step': BASELINE.json configs[2] - ResNet50-FPN Mask R-CNN full forward + backward +
optimizer update, batch 2 per GPU, 1024x1024, fp32, synthetic COCO-shaped data resident in HBM."""
import json
import os
import time

import numpy as np
import torch

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA dense peak


def bench_step(args, rank, world):
    from chainer_maskrcnn.model.maskrcnn import MaskRCNN
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
    from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
    from chainer_maskrcnn.utils.synthetic import make_batch
    from chainer_maskrcnn._hip import nn as hnn
    dev = torch.device('cuda', 0 if os.environ.get('MRCNN_BENCH_SINGLE_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    N, H, W = 2, 1024, 1024
    mask_rows = getattr(args, 'mask_rows', 'all')
    model = MaskRCNN(n_fg_class=80, device=dev, seed=1234)
    chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows=mask_rows)
    opt = MomentumSGD(lr=1e-3, momentum=0.9).setup(chain)
    opt.add_hook(WeightDecay(0.0005))
    if world > 1:
        opt.enable_data_parallel()
    b = make_batch(100 + rank, N, H, W, G=8)
    imgs, bb, lab, masks = (torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks'))

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
    dt = _max_over_ranks(dt, world, dev)
    loss = float(chain.observation['loss'])
    ips = N * world * args.steps / dt

    # the same step with the mask branch on the positive rows only (identical loss and gradients)
    alt = None
    if mask_rows == 'all' and world == 1:
        chain.mask_rows = 'positives'
        for _ in range(2):
            opt.update(chain, imgs, bb, lab, masks, 1.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            opt.update(chain, imgs, bb, lab, masks, 1.0)
        torch.cuda.synchronize()
        alt = N * 5 / (time.perf_counter() - t1)
        chain.mask_rows = 'all'

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
    for kind, macs, e0, e1, _shape, executed in recs:
        a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += 2.0 * macs
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += 2.0 * executed
    flops = sum(a[1] for a in agg.values()) / n_prof
    exe_flops = sum(a[3] for a in agg.values()) / n_prof
    secs = sum(a[2] for a in agg.values()) / n_prof
    launches = sum(a[0] for a in agg.values()) // n_prof
    ach = flops / secs / 1e12
    out = {
        'metric': 'images/sec (1024^2 COCO, bs=2/GPU) at 1/2/4/8 MI355X; ROIAlign bwd HBM GB/s',
        'value': round(ips, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[2] ResNet50-FPN Mask R-CNN full fwd+bwd+SGD step, batch 2/GPU, 1024x1024, fp32, '
                               '8 gt boxes/image, 2000 proposals -> 256 sampled RoIs/image, mask branch on %s rows'
                               % ('the <=64 positive' if mask_rows == 'positives' else 'all 256 sampled'),
                   'global_batch': N * world, 'launch_mode': mode, 'parallelism': 'dp%d: RCCL all-reduce of the flat gradient buffer in 25 MB '
                   'buckets on a side stream' % world if world > 1 else 'single GPU', 'final_loss': round(loss, 4)},
        'roofline': {'bound': 'mfma', 'kernel': 'k_conv_igemm<fwd|bwd_data|bwd_filter> (all %d launches of a step)' % launches,
                     'achieved': round(ach, 3), 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(ach / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': None,
                     'algorithmic_flops_per_step': flops, 'conv_ms_per_step': round(secs * 1e3, 3),
                     'mfma_executed': {'flops_per_step': exe_flops, 'TFLOPs': round(exe_flops / secs / 1e12, 3),
                                       'frac_of_peak': round(exe_flops / secs / 1e12 / MFMA_F32_PEAK_TFLOPS, 4)},
                     'by_kind': {k: {'launches': a[0] // n_prof, 'TFLOPs': round(a[1] / a[2] / 1e12, 3),
                                     'executed_TFLOPs': round(a[3] / a[2] / 1e12, 3),
                                     'ms': round(a[2] / n_prof * 1e3, 3)} for k, a in agg.items()},
                     'note': 'HIP events around every conv call on %d instrumented steps right after the timed region. '
                             'achieved = ALGORITHMIC flops (2 x the direct convolution\'s MACs on un-padded channels) / time; the '
                             '3x3 layers with >= 256 channels run as Winograd F(2x2,3x3) (transforms + batched GEMM inside the '
                             'bracket), which executes 2.25x fewer MFMA flops than it is credited with: mfma_executed is what the '
                             'pipes really do (padded channels included)' % n_prof},
    }
    if alt is not None:
        out['config']['images_per_sec_mask_branch_on_positive_rows_only'] = round(alt, 3)
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
