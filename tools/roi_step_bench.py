"""ROIAlign FPN forward / backward timed in isolation on the RoIs the benchmark step actually samples."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for it in range(4):
    opt.update(chain, *args, 1.0)
rois, levels, label = chain.mask_inputs
rois = rois.clone(); levels = levels.clone()
print('levels', torch.bincount(levels.cpu().long(), minlength=5).tolist())
scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32., 1 / 64.]
feats = [torch.randn((2, 1024 // s, 1024 // s, 256), device=dev) for s in (4, 8, 16, 32, 64)]


def timed(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for P in (7, 14):
    gy = torch.randn((rois.shape[0], P, P, 256), device=dev)
    gxs = [torch.zeros_like(f) for f in feats]
    t_f = timed(lambda: hd.roi_align_fpn_fwd(feats, rois, levels, P, scales))
    t_b = timed(lambda: hd.roi_align_fpn_bwd(gy, gxs, rois, levels, P, scales, accumulate=False))
    L, arr_p, Hs, Ws, sc = hd._level_args(gxs, scales)
    def nows():
        _hip.check(_hip.lib().mrcnn_roi_align_fpn_bwd_f32(_hip.ptr(gy), arr_p, Hs, Ws, sc, L, 2, 256, _hip.ptr(rois), _hip.ptr(levels),
                                                           rois.shape[0], P, P, 2, 0, None, 0, _hip.stream_ptr()))
    t_n = timed(nows)
    print('P=%2d  fwd %7.1f us   bwd (RoI split) %7.1f us   bwd (one workgroup per tile) %7.1f us' % (P, t_f, t_b, t_n))

print('--- backward P=14 on subsets')
P = 14
for name, sel in (('all', torch.ones_like(levels, dtype=torch.bool)), ('one RoI', torch.arange(levels.numel(), device=dev) == 0),
                  ('level 0-1', levels <= 1), ('level 2', levels == 2), ('level 3', levels == 3), ('level 4', levels == 4)):
    r2, l2 = rois[sel].contiguous(), levels[sel].contiguous()
    gy = torch.randn((r2.shape[0], P, P, 256), device=dev)
    gxs = [torch.zeros_like(f) for f in feats]
    t_b = timed(lambda: hd.roi_align_fpn_bwd(gy, gxs, r2, l2, P, scales, accumulate=False))
    t_a = timed(lambda: hd.roi_align_fpn_bwd(gy, gxs, r2, l2, P, scales, accumulate=True))
    print('%-10s R=%4d  overwrite %7.1f us   accumulate %7.1f us' % (name, r2.shape[0], t_b, t_a))
