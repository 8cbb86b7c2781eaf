"""Per-layer convolution timing of one full training step (HIP events around every launch)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows=sys.argv[1] if len(sys.argv) > 1 else 'all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for _ in range(3):
    opt.update(chain, *args, 1.0)
hnn.PROFILE = []
chain.use_aux_stream = False
opt.update(chain, *args, 1.0)
torch.cuda.synchronize()
agg = {}
for rec in hnn.PROFILE:
    kind, macs, e0, e1, shape, ex = rec[:6]
    a = agg.setdefault((kind,) + shape, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += 2.0 * macs; a[2] += e0.elapsed_time(e1); a[3] += 2.0 * ex
tot = sum(a[2] for a in agg.values())
print('total conv ms %.2f (whole call brackets: GEMM + transforms + sums)' % tot)
cat = {}
for key, a in agg.items():
    c = cat.setdefault((key[0], '1x1' if key[2] == 1 else ('3x3 winograd' if a[3] < 0.9 * a[1] else 'kxk direct')), [0, 0.0, 0.0])
    c[0] += a[0]; c[1] += a[2]; c[2] += a[3]
for k, c in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print('%-11s %-13s n=%3d %7.3f ms  executed %6.1f TF/s' % (k + (c[0], c[1], c[2] / c[1] / 1e9)))
print('%-10s %9s %2s %5s %5s %4s %8s %7s %7s %6s' % ('kind', 'pixels', 'k', 'cin', 'cout', 'n', 'ms', 'effTF/s', 'exeTF/s', '%'))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print('%-10s %9d %2d %5d %5d %4d %8.3f %7.1f %7.1f %6.2f' % (key + (a[0], a[2], a[1] / a[2] / 1e9, a[3] / a[2] / 1e9, 100 * a[2] / tot)))
print('--- by time lost vs 135 TF/s executed')
lost = sorted(((a[2] - a[3] / 135e9, key, a) for key, a in agg.items()), reverse=True)
for l_, key, a in lost[:28]:
    print('%-10s %9d %2d %5d %5d %4d %8.3f %7.1f  lost %.3f ms' % (key + (a[0], a[2], a[3] / a[2] / 1e9, l_)))
print('total lost', sum(l_ for l_, _, _ in lost if l_ > 0))
