"""A/B in one process: BatchNorm statistics from the convolution epilogue on / off."""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn.nn import core
from chainer_maskrcnn.model import fpn_maskrcnn_train_chain as tc
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
chain = tc.FPNMaskRCNNTrainChain(model, mask_loss_fun=tc.calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
def run(on):
    core.FUSE_BN_STATS = on
    for _ in range(3): opt.update(chain, *args, 1.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): opt.update(chain, *args, 1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
for rnd in range(4):
    print('round %d: separate statistics pass %.2f ms | from the GEMM epilogue %.2f ms' % (rnd, run(False), run(True)))
core.FUSE_BN_STATS = True
n = [0, 0]
orig = hnn.conv2d_fwd_bnstats_raw
def counting(*a, **k):
    r = orig(*a, **k); n[0] += 1; n[1] += r is not None; return r
hnn.conv2d_fwd_bnstats_raw = counting
opt.update(chain, *args, 1.0); torch.cuda.synchronize()
print('BatchNorm convolutions asked: %d, fused: %d' % tuple(n))
