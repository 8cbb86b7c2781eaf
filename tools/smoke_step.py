import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import numpy as np, torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
full = len(sys.argv) > 1 and sys.argv[1] == 'full'
dev = torch.device('cuda:0')
shrink = None if full else dict(stages=(1, 1, 1, 1), width_div=2)
m = MaskRCNN(n_fg_class=80, _test_shrink=shrink)
print('params', m.ps.n_params(), 'flat', m.ps.size)
chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss)
opt = MomentumSGD(lr=1e-3, momentum=0.9).setup(chain); opt.add_hook(WeightDecay(0.0005))
N, H, W, G = 2, (1024 if full else 256), (1024 if full else 320), 4
rs = np.random.RandomState(0)
imgs = torch.from_numpy(rs.rand(N, 3, H, W).astype(np.float32)).to(dev)
bb = np.zeros((N, G, 4), np.float32); masks = np.zeros((N, G, H, W), np.uint8)
for i in range(N):
    for g in range(G):
        h, w = rs.uniform(40, H / 2), rs.uniform(40, W / 2)
        y, x = rs.uniform(0, H - h), rs.uniform(0, W - w)
        bb[i, g] = (y, x, y + h, x + w)
        masks[i, g, int(y):int(y + h), int(x):int(x + w)] = 1
labels = torch.from_numpy(rs.randint(0, 80, (N, G)).astype(np.int32)).to(dev)
bb = torch.from_numpy(bb).to(dev); masks = torch.from_numpy(masks).to(dev)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    loss = opt.update(chain, imgs, bb, labels, masks, 1.0)
    torch.cuda.synchronize(); dt = time.time() - t0
    obs = {k: float(v) for k, v in chain.observation.items()}
    print(it, 'ms %.1f' % (dt * 1e3), obs, 'n_pos', chain.targets['n_pos'].tolist(), 'n_rois', chain.rpn_out['n_rois'].tolist())
g = m.ps.grads
print('grad finite', bool(torch.isfinite(g).all()), 'absmax', float(g.abs().max()))
