import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import nn as hnn, lib as _lib
lib = _lib(); dev = torch.device('cuda:0')
lib.mrcnn_conv2d_set_winograd_pass_tiles(0, 0, 0)
if os.environ.get('MRCNN_WINO_ORDER'): lib.mrcnn_debug_wino_banded(int(os.environ['MRCNN_WINO_ORDER']))
for (N, H, W, Ci, Co) in [(2, 256, 256, 256, 256), (2, 128, 128, 256, 256), (2, 64, 64, 256, 256), (2, 32, 32, 256, 256), (256, 14, 14, 256, 256), (512, 7, 7, 256, 256), (2, 256, 256, 64, 64), (2, 128, 128, 128, 128), (2, 32, 32, 512, 512)]:
    x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, 3, 3, Ci), device=dev) * 0.05
    b = torch.zeros((Co,), device=dev)
    hnn.conv2d_fwd_raw(x, w, b, 1, 1, True)
    torch.cuda.synchronize()
