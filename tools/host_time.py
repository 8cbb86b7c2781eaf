"""Host enqueue time vs GPU time of the training step (is the step launch-bound?)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all', gemm_arithmetic=os.environ.get('MRCNN_GEMM_ARITHMETIC', 'bf16x6_behind_backbone'))
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for _ in range(3):
    opt.update(chain, *args, 1.0)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    enq = []
    for _ in range(10):
        a = time.perf_counter()
        opt.update(chain, *args, 1.0)
        enq.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('10 steps back to back: host enqueue %.1f ms/step (min %.1f max %.1f; includes waiting for a free queue slot once the host is a step ahead), '
          'wall incl. final sync %.1f ms/step, tail wait %.1f ms' % ((t1 - t0) * 100, min(enq) * 1e3, max(enq) * 1e3, (t2 - t0) * 100, (t2 - t1) * 1e3))
# the host's own cost: every step enqueued into an EMPTY queue (device idle at the start), so no launch ever waits for the device
for rep in range(3):
    enq = []
    for _ in range(10):
        torch.cuda.synchronize()
        a = time.perf_counter()
        opt.update(chain, *args, 1.0)
        enq.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    enq.sort()
    print('10 steps, each into an empty queue: host enqueue median %.1f ms/step (min %.1f max %.1f)' % (enq[5] * 1e3, enq[0] * 1e3, enq[-1] * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    opt.update(chain, *args, 1.0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
