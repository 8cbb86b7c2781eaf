"""Same-process A/B of a module-level switch on the full training step (boxes differ by ~3 %, so A/B across runs is
meaningless for sub-ms effects).  usage: ab_step.py module.path ATTR valueA valueB"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
mod = importlib.import_module(sys.argv[1]); attr = sys.argv[2]
vals = [eval(v) for v in sys.argv[3:5]]
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for v in vals:
    setattr(mod, attr, v)
    for _ in range(3): opt.update(chain, *args, 1.0)
res = {repr(v): [] for v in vals}
for rep in range(4):
    for v in vals:
        setattr(mod, attr, v)
        opt.update(chain, *args, 1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): opt.update(chain, *args, 1.0)
        torch.cuda.synchronize()
        res[repr(v)].append((time.perf_counter() - t0) * 100)
for k, v in res.items():
    print('%s = %-8s %s ms/step  (mean %.3f)' % (attr, k, ' '.join('%.3f' % x for x in v), sum(v) / len(v)))
