"""Phases of the UN-PROFILED step from HIP events on the main stream (the rocprofv3 timeline of this step is host-bound wherever kernels
are short, so its idle windows around the proposal chain are the profiler's): extractor + RPN heads | proposal chain + target sampling |
heads + losses | backward + update.  usage: phase_events.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
from chainer_maskrcnn.utils.synthetic import make_batch
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all', gemm_arithmetic=os.environ.get('MRCNN_GEMM_ARITHMETIC', 'bf16x6_behind_backbone'))
opt = MomentumSGD(lr=1e-3).setup(chain); opt.add_hook(WeightDecay(5e-4))
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
ev = []
def mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(torch.cuda.current_stream(dev)); ev.append((tag, e))
rpn, ptc = model.rpn, chain.proposal_target_creator
fp, sb, call = rpn.forward_padded, ptc.sample_batch, chain.__class__.__call__
def fp2(*a, after_heads=None, **kw):
    def ah(*x):
        mark('heads'); return after_heads(*x)
    return fp(*a, after_heads=ah, **kw)
def sb2(*a, **kw):
    r = sb(*a, **kw); mark('sampled'); return r
rpn.forward_padded, ptc.sample_batch = fp2, sb2
for _ in range(5): opt.update(chain, *args, 1.0)
torch.cuda.synchronize(); ev.clear()
for _ in range(K):
    mark('start'); opt.update(chain, *args, 1.0)
mark('start'); torch.cuda.synchronize()
import collections
acc = collections.defaultdict(list)
for (t0, e0), (t1, e1) in zip(ev, ev[1:]): acc[t0 + ' -> ' + t1].append(e0.elapsed_time(e1))
for k, v in acc.items(): print('%-22s mean %.3f ms  min %.3f  max %.3f' % (k, sum(v) / len(v), min(v), max(v)))
print('step %.3f ms' % (ev[0][1].elapsed_time(ev[-1][1]) / K))
