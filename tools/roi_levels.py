"""Histogram of the FPN levels / sizes of the sampled RoIs in the benchmark step (random-init weights)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for it in range(6):
    opt.update(chain, *args, 1.0)
    rois, levels, label = chain.mask_inputs
    r = rois.cpu()
    w = (r[:, 3] - r[:, 1]); h = (r[:, 4] - r[:, 2])
    print(it, 'levels', torch.bincount(levels.cpu().long(), minlength=5).tolist(), 'median w/h %.0f/%.0f' % (w.median(), h.median()),
          'n_pos', int((label > 0).sum()))
