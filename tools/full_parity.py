"""Whole-step parity of the FULL-WIDTH network (ResNet-50-FPN, 80 classes) against the float64 oracle on one image,
for the direct convolution kernels and the Winograd tile choices.  Checker code (oracle/) - test infrastructure only."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.utils.synthetic import make_batch
from oracle import model as om
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device('cuda:0')
D = torch.float64
torch.set_num_threads(min(32, os.cpu_count() or 1))
m = MaskRCNN(n_fg_class=80, device=dev, seed=5)
chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss, mask_rows='all')
b = make_batch(11, 1, S, S, G=6)
b['bboxes'][:, :, 2:] = np.minimum(b['bboxes'][:, :, 2:], [S, S])
bt = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
ps = m.ps
results = {}
oracle_grads = None
for name, (minc, minpix, tile) in (('direct', (100000, 1 << 30, 0)), ('winograd F(2x2)', (256, 2048, 2)), ('winograd auto (F(4x4) where cheaper)', (256, 2048, 0))):
    _hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(minc, minpix, tile))
    loss = chain(bt['imgs'], bt['bboxes'], bt['labels'], bt['masks'], 1.0)
    loss.backward()
    obs = {k: float(v) for k, v in chain.observation.items()}
    if True:       # the proposals (hence the sampled targets) depend on the convolution path: one oracle run per path
        t0 = time.time()
        params = {n: ps.p(n).detach().cpu().to(D).requires_grad_(True) for n in ps.names()}
        t = {k: v.cpu().numpy() for k, v in chain.targets.items() if torch.is_tensor(v)}
        t['gt_rpn_loc'], t['gt_rpn_label'] = (x.cpu().numpy() for x in chain.rpn_targets)
        t['mask_rois_xy5'], t['mask_levels'], t['mask_label'] = (x.cpu().numpy() for x in chain.mask_inputs)
        oracle = om.OracleStep(params, tuple(len(s) for s in m.extractor.stages), m.head.n_class, m.head.LOC0)
        img4 = torch.cat([bt['imgs'].cpu().permute(0, 2, 3, 1), torch.zeros((1, S, S, 1))], -1).to(D)
        out = oracle.losses(img4, t)
        names = ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
        sum(out[k] for k in names).backward()
        oracle_grads = {n: (params[n].grad if params[n].grad is not None else torch.zeros_like(params[n])) for n in ps.names()}
        oracle_loss = {k: float(out[k].detach()) for k in names}
        print('oracle (float64, CPU) took %.0f s' % (time.time() - t0))
    gmax = max(float(g.abs().max()) for g in oracle_grads.values())
    worst, wname, errs = 0.0, '', []
    for n in ps.names():
        want = oracle_grads[n]
        got = ps.g(n).cpu().to(D)
        scale = max(float(want.abs().max()), 1e-3 * gmax)
        err = float((got - want).abs().max()) / scale
        errs.append((err, n))
        if err > worst:
            worst, wname = err, n
    errs.sort()
    q = lambda f: errs[int(f * (len(errs) - 1))][0]
    print('   per-tensor gradient error quantiles: median %.2e  90%% %.2e  99%% %.2e   heads: %s' % (
        q(0.5), q(0.9), q(0.99), ', '.join('%s %.1e' % (n.split('/')[-2], e) for e, n in errs if n.startswith('head/') and n.endswith('/W'))))
    lerr = max(abs(obs[k] - oracle_loss[k]) / max(abs(oracle_loss[k]), 1e-3) for k in oracle_loss)
    print('%-40s worst gradient error %.2e (%s)   worst loss error %.2e' % (name, worst, wname, lerr))
_hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, 0))
