"""One convolution geometry, forward + backward-data + backward-filter, a few calls each - for rocprofv3 --pmc passes
(is the A operand of the batched Winograd GEMM fetched once?).  conv_pmc_one.py N H W Cin Cout K"""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import torch
from chainer_maskrcnn._hip import nn as hnn
dev = torch.device('cuda:0')
N, H, W, Ci, Co, K = [int(v) for v in sys.argv[1:7]]
hnn.set_winograd_pass_tiles(0, 0, 0)
x = torch.randn((N, H, W, Ci), device=dev); w = torch.randn((Co, K, K, Ci), device=dev) * 0.02; b = torch.zeros((Co,), device=dev)
gy = torch.randn((N, H, W, Co), device=dev)
for _ in range(3):
    y = hnn.conv2d_fwd_raw(x, w, b, 1, K // 2, False)
    gx = hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, K // 2)
    gw = hnn.conv2d_bwd_filter_raw(x, gy, tuple(w.shape), 1, K // 2, False)
torch.cuda.synchronize()
