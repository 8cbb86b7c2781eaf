"""Winograd input transform check: V = B^T d B of a forward call (F(2x2), kept with keep_v) against a float64 NumPy/torch
reference, plus run-to-run reproducibility of V and of the output.  This is the script that localised the gfx950 store-data hazard
of DESIGN.md section 3.1b (buffer_store_dwordx4 with an SGPR soffset followed by a VALU overwrite of the data registers): wrong
elements showed up as channel % 64 in {49, 53, 57, 61} of 7 of the 16 planes."""
import os, sys
sys.path.insert(0, '/root/repo/chainer-maskrcnn_amd'); sys.path.insert(0, '/root/repo')
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import nn as hnn
DEV='cuda:0'
tile = 2
_hip.check(_hip.lib().mrcnn_conv2d_set_winograd_thresholds(256, 2048, tile))
_hip.check(_hip.lib().mrcnn_conv2d_set_winograd_pass_tiles(0, 0, 0))
N, H, W, Cin, Cout = (2, 48, 48, 256, 256)
g = torch.Generator().manual_seed(1)
x = torch.randn((N, H, W, Cin), generator=g).to(DEV)
w = (torch.randn((Cout, 3, 3, Cin), generator=g) / (9 * Cin) ** 0.5).to(DEV)
b = torch.randn((Cout,), generator=g).to(DEV)
y1, v1 = hnn.conv2d_fwd_raw(x, w, b, 1, 1, False, keep_v=True)
v1 = v1.clone(); torch.cuda.synchronize()
y2, v2 = hnn.conv2d_fwd_raw(x, w, b, 1, 1, False, keep_v=True)
torch.cuda.synchronize()
print('V shape', tuple(v1.shape), 'V mismatches', int((v1 != v2).sum()), 'Y mismatches', int((y1 != y2).sum()))
# reference V: B^T d B for m = 2
xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2).double().cpu(), (1, 1, 1, 1))
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
th, tw = H // 2, W // 2
T = N * th * tw
vv = v1.reshape(-1)[:16 * T * Cin].reshape(16, T, Cin).double().cpu()
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)            # N, C, th, tw, 4, 4
ref = torch.einsum('ij,ncyxjk,lk->ilnyxc', Bt, tiles, Bt).reshape(16, T, Cin)
err = (vv - ref).abs()
print('max |V - ref|', float(err.max()), 'bad elements', int((err > 1e-4).sum()), 'of', err.numel())
bad = (err > 1e-4).nonzero()
if len(bad):
    print('k values', sorted(set(bad[:, 0].tolist())), 'tiles', sorted(set(bad[:, 1].tolist()))[:20], 'channels', sorted(set(bad[:, 2].tolist()))[:12])
tt = bad[:, 1]
n_ = tt // (th * tw); ty = (tt // tw) % th; tx = tt % tw
border = (ty == 0) | (ty == th - 1) | (tx == 0) | (tx == tw - 1)
print('bad on border tiles', int(border.sum()), 'interior', int((~border).sum()))
import collections
print('per k', collections.Counter(bad[:, 0].tolist()))
print('channel mod 64 histogram', sorted(collections.Counter((bad[:, 2] % 64).tolist()).items()))
mm = (v1 != v2).reshape(-1)[:16 * T * Cin].reshape(16, T, Cin).nonzero().cpu()
print('run-to-run: k', sorted(set(mm[:, 0].tolist())), 'chan mod 64', sorted(set((mm[:, 2] % 64).tolist())))
i0 = bad[0]; print('example', i0.tolist(), float(vv[tuple(i0)]), float(ref[tuple(i0)]))
