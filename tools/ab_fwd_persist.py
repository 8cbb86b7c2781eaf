"""Forward pass only (train chain __call__, no backward) with the persistent GEMM off / on, in one process."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.utils.synthetic import make_batch
from chainer_maskrcnn._hip import lib as _lib
lib = _lib()
dev = torch.device('cuda:0')
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
hi = torch.cuda.Stream(device=dev, priority=-1)
for use_hi in (False, True):
    for rep in range(3):
        for on in (0, 1):
            lib.mrcnn_conv2d_set_persistent(on, 1.5)
            with torch.cuda.stream(hi if use_hi else torch.cuda.current_stream()):
                for _ in range(2): chain(*args, 1.0)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10): chain(*args, 1.0)
                torch.cuda.synchronize()
            print('high-priority stream %s persistent %d: forward %.3f ms' % (use_hi, on, (time.perf_counter() - t0) * 100), flush=True)
lib.mrcnn_conv2d_set_persistent(0, 1.5)
