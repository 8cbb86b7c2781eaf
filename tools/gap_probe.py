"""Is the GPU idle between the end of backward and the SGD kernel?  Events around the hand-over, no profiler."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn._hip import ops
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
ev = []
orig_bwd = chain.backward
def bwd():
    orig_bwd()
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(['bwd_end', e, time.perf_counter()])
chain.backward = bwd
orig_sgd = ops.sgd_momentum_wd
def sgd(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(['sgd_begin', e, time.perf_counter()])
    r = orig_sgd(*a, **k)
    e2 = torch.cuda.Event(enable_timing=True); e2.record(); ev.append(['sgd_end', e2, time.perf_counter()])
    return r
ops.sgd_momentum_wd = sgd
import chainer_maskrcnn.optimizers as O
O.ops.sgd_momentum_wd = sgd
for _ in range(8):
    opt.update(chain, *args, 1.0)
torch.cuda.synchronize()
for i in range(len(ev) - 1):
    a, b2 = ev[i], ev[i + 1]
    print('%-10s -> %-10s  gpu %.3f ms   host %.3f ms' % (a[0], b2[0], a[1].elapsed_time(b2[1]), (b2[2] - a[2]) * 1e3))
