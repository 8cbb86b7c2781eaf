"""Same-process A/B of settings on the full training step (boxes differ by ~3 %, so A/B across runs is meaningless for
sub-ms effects).  usage: ab_step.py 'stmtA' 'stmtB' [...]  - Python statements executed before timing each variant,
with `core` (nn.core), `chain`, `lib` (the C library), `hnn` in scope, e.g.
  ab_step.py 'core.FILTER_GRAD_ON_SIDE_STREAM=True' 'core.FILTER_GRAD_ON_SIDE_STREAM=False'"""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn.nn import core
from chainer_maskrcnn._hip import nn as hnn, lib as _lib
lib = _lib()
vals = sys.argv[1:]
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all', gemm_arithmetic=os.environ.get('MRCNN_GEMM_ARITHMETIC', 'bf16x6_behind_backbone'))
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for v in vals:
    exec(v)
    for _ in range(3): opt.update(chain, *args, 1.0)
res = {v: [] for v in vals}
for rep in range(4):
    for v in vals:
        exec(v)
        opt.update(chain, *args, 1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): opt.update(chain, *args, 1.0)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) * 100)
for k, v in res.items():
    print('%-70s %s ms/step  (mean %.3f)' % (k, ' '.join('%.3f' % x for x in v), sum(v) / len(v)))
