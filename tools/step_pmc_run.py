"""Exactly K configs[2] training steps (batch 2, 1024x1024, the bench.py workload) and nothing else, for rocprofv3 --pmc
passes: every launch in the counter file belongs to one of the K steps.  usage: step_pmc_run.py [K]"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
if os.environ.get('MRCNN_WINO_ORDER'):      # measurement: tile order of the Winograd input transforms (mrcnn_debug_wino_banded)
    from chainer_maskrcnn._hip import lib as _lib
    _lib().mrcnn_debug_wino_banded(int(os.environ['MRCNN_WINO_ORDER']))
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev, seed=1234)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all', gemm_arithmetic=os.environ.get('MRCNN_GEMM_ARITHMETIC', 'bf16x6_behind_backbone'))
opt = MomentumSGD(lr=1e-3, momentum=0.9).setup(chain)
opt.add_hook(WeightDecay(0.0005))
b = make_batch(100, 2, 1024, 1024, G=8)
imgs, bb, lab, masks = (torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks'))
for _ in range(K):
    opt.update(chain, imgs, bb, lab, masks, 1.0)
torch.cuda.synchronize()
print('loss', float(chain.observation['loss']))
