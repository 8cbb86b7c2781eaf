"""GEMM launches only (mrcnn_conv2d_set_debug_skip(2)) of every distinct convolution call of the step, replayed standalone:
where does the GEMM time go, and how far is each call from the MFMA peak?"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn._hip import nn as hnn, lib, check
dev = torch.device('cuda:0')
SPLIT = tuple(int(v) for v in sys.argv[1].split(',')) if len(sys.argv) > 1 else (0, 0, 0)      # mrcnn_conv2d_set_split_operands
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for _ in range(2):
    opt.update(chain, *args, 1.0)
hnn.PROFILE = []
chain.use_aux_stream = False
opt.update(chain, *args, 1.0)
torch.cuda.synchronize()
recs, hnn.PROFILE = hnn.PROFILE, None
geoms = {}
for rec in recs:
    geoms[(rec[0], rec[6], rec[7])] = geoms.get((rec[0], rec[6], rec[7]), 0) + 1
rows = []
check(lib().mrcnn_conv2d_set_split_operands(*SPLIT))
PLANS = [[int(v) for v in pl.split(',')] for pl in sys.argv[2].split(';')] if len(sys.argv) > 2 else []      # mrcnn_debug_conv_plan(fill, filter_rounds, force_tile)
if len(PLANS) == 1:
    check(lib().mrcnn_debug_conv_plan(*PLANS[0]))
PARTS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]
base = hnn.winograd_pass_tiles()
for (kind, g, tiles), cnt in geoms.items():
    hnn.set_winograd_pass_tiles(*tiles)
    N, H, W, Cin, Cout, KH, KW, stride, pad = g
    Ho, Wo = hnn.conv_out(H, KH, stride, pad), hnn.conv_out(W, KW, stride, pad)
    x = torch.empty((N, H, W, Cin), device=dev).normal_(); w = torch.empty((Cout, KH, KW, Cin), device=dev).normal_()
    gy = torch.empty((N, Ho, Wo, Cout), device=dev).normal_()
    v = hnn.conv2d_fwd_raw(x, w, None, stride, pad, False, keep_v=True)[1] if kind == 'bwd_filter' else None
    fn = {'fwd': lambda: hnn.conv2d_fwd_raw(x, w, None, stride, pad, False, keep_v=True),
          'bwd_data': lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), stride, pad),
          'bwd_filter': lambda: hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), stride, pad, False, wino_v=v)}[kind]
    check(lib().mrcnn_conv2d_set_debug_skip(2))
    fn()
    ts = []
    for parts in PARTS:
        check(lib().mrcnn_debug_conv_parts(parts))
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5 * 1e-3)
    if len(PLANS) > 1:          # several plans: the first is the reference (ts[0]), the others are A/B columns
        ts = []
        for pl in PLANS:
            check(lib().mrcnn_debug_conv_plan(*pl))
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e-3)
        check(lib().mrcnn_debug_conv_plan(*PLANS[0]))
    check(lib().mrcnn_debug_conv_parts(0))
    check(lib().mrcnn_conv2d_set_debug_skip(0))
    t = ts[0]
    exe = 2.0 * lib().mrcnn_conv2d_executed_macs(*g, {'fwd': 0, 'bwd_data': 1, 'bwd_filter': 2}[kind])
    rows.append((t * cnt, kind, g, tiles, cnt, t, exe, ts))
hnn.set_winograd_pass_tiles(*base)
tot = sum(r[0] for r in rows); totf = sum(r[6] * r[4] for r in rows)
print('split operands %s: GEMM-only total %.2f ms, %.1f TF/s executed (float32-equivalent flops)' % (SPLIT, tot * 1e3, totf / tot / 1e12))
for kd in ('fwd', 'bwd_data', 'bwd_filter'):
    print('  %-10s %.2f ms' % (kd, 1e3 * sum(r[0] for r in rows if r[1] == kd)))
print('%-10s %-38s %5s %3s %8s %7s %9s' % ('kind', 'N,H,W,Cin,Cout,KH,KW,s,p', 'tiles', 'n', 'ms', 'TF/s', 'lost@135'))
if len(PLANS) > 1:
    print('totals per plan %s: %s ms; best plan per call: %.2f ms' % (PLANS, ' '.join('%.2f' % (1e3 * sum(r[7][i] * r[4] for r in rows)) for i in range(len(PLANS))),
                                                                      1e3 * sum(min(r[7]) * r[4] for r in rows)))
    for kd in ('fwd', 'bwd_data', 'bwd_filter'):
        print('  %-10s %s | best %.2f' % (kd, ' '.join('%.2f' % (1e3 * sum(r[7][i] * r[4] for r in rows if r[1] == kd)) for i in range(len(PLANS))),
                                          1e3 * sum(min(r[7]) * r[4] for r in rows if r[1] == kd)))
if len(PARTS) > 1:
    print('totals per debug_conv_parts mask %s: %s ms' % (PARTS, ' '.join('%.2f' % (1e3 * sum(r[7][i] * r[4] for r in rows)) for i in range(len(PARTS)))))
for r in sorted(rows, key=lambda r: -(r[0] - r[6] * r[4] / 135e12)):
    print('%-10s %-38s %5s %3d %8.3f %7.1f %9.3f  %s' % (r[1], ','.join(map(str, r[2])), ''.join(str(t) if t >= 0 else 'd' for t in r[3]), r[4], r[0] * 1e3,
                                                  r[6] / r[5] / 1e12, (r[0] - r[6] * r[4] / 135e12) * 1e3, ' '.join('%.0f' % (v * 1e6) for v in r[7])))
