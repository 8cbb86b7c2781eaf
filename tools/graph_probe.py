"""HIP-graph replay of the whole step against the eager multi-stream step, same process ("re-measure graph replay with a
root cause"): ms per step of both, and - under rocprofv3 --kernel-trace - tools/trace_streams.py shows which queues a replay's kernels
land on.  usage: graph_probe.py [steps=20]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay, GraphedStep
from chainer_maskrcnn.utils.synthetic import make_batch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all', gemm_arithmetic='bf16x6_behind_backbone')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
for _ in range(5):
    opt.update(chain, *args, 1.0)
torch.cuda.synchronize()


def timed(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


eager = [timed(lambda: opt.update(chain, *args, 1.0)) for _ in range(3)]
print('eager, high-priority step stream: %s ms/step' % ' '.join('%.3f' % v for v in eager))
opt.high_priority_stream = False
eager_np = [timed(lambda: opt.update(chain, *args, 1.0)) for _ in range(3)]
print('eager, normal-priority stream:    %s ms/step' % ' '.join('%.3f' % v for v in eager_np))
try:
    g = GraphedStep(opt, chain, args, 1.0)
    rep = [timed(lambda: g(*args)) for _ in range(3)]
    print('graph replay:                     %s ms/step' % ' '.join('%.3f' % v for v in rep))
    t0 = time.perf_counter()
    for _ in range(K):
        g(*args)
    host = (time.perf_counter() - t0) / K * 1e3
    torch.cuda.synchronize()
    print('graph replay host enqueue: %.3f ms/step' % host)
except Exception as e:
    print('graph capture failed: %r' % (e,))
