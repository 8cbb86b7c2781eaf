"""A/B: the training step issued on a HIGH-priority HIP stream (weight-gradient and aux streams at normal priority) vs
everything at normal priority.  The main stream is the critical path; the side streams only have to finish by the end."""
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
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]


def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        opt.update(chain, *args, 1.0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


hi = torch.cuda.Stream(device=dev, priority=-1)
for _ in range(3):
    opt.update(chain, *args, 1.0)
for rep in range(3):
    t0 = run(20)
    hi.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(hi):
        for _ in range(3):
            opt.update(chain, *args, 1.0)
        t1 = run(20)
    torch.cuda.current_stream().wait_stream(hi)
    print('normal priority %.3f ms/step   main stream high priority %.3f ms/step' % (t0, t1))
