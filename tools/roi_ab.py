"""ROIAlign backward A/B: variants 1 (barrier-synchronised 8x8 tiles), 2 (independent waves on 4x4 patches) and 3 (table-driven) on
BASELINE configs[1] and on the RoIs of a real training step (FPN levels), interleaved in one process; prints the
difference between the two results (summation order only) and the timings."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.model.head import fpn_roi_mask_head as hd
from chainer_maskrcnn.utils.synthetic import config2_inputs
dev = torch.device('cuda:0')
lib = _hip.lib()


def timed(f, n=50, rounds=5):
    for _ in range(5): f()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return min(out), float(np.median(out))


x, yx, gy = config2_inputs()
N, C, H, W = x.shape
R, _, PH, PW = gy.shape
rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
gx = {v: torch.empty((N, C, H, W), device=dev).contiguous(memory_format=torch.channels_last) for v in (1, 2, 3)}
nb = lib.mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, PH, PW, 2)
ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=dev)
algo = 4 * (N * C * H * W + R * C * PH * PW) + 20 * R


def bwd(v):
    _hip.check(lib.mrcnn_roi_align_set_bwd_variant(v))
    _hip.check(lib.mrcnn_roi_align_bwd_ws_f32(_hip.ptr(gyt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW, 0.25, 2, _hip.ptr(gx[v]), _hip.ptr(ws), nb, _hip.stream_ptr()))


for v in (1, 2, 3):
    gx[v].fill_(float('nan'))
    bwd(v)
torch.cuda.synchronize()
d = (gx[1] - gx[2]).abs().max().item()
print('configs[1]: max |v1 - v2| = %.3e (scale %.3e), NaNs in v2: %d; v3 == v2 bitwise: %s' % (d, gx[1].abs().max().item(), int(torch.isnan(gx[2]).sum()), torch.equal(gx[2], gx[3])))
for rnd in range(2):
    for v in (1, 2, 3):
        mn, med = timed(lambda: bwd(v))
        print('configs[1] bwd variant %d: min %.1f us  median %.1f us   %.0f GB/s algorithmic (%.3f of 8 TB/s)' % (v, mn, med, algo / mn / 1e3, algo / mn / 1e3 / 8000))

# ---- in-step shapes: RoIs sampled by a real step (levels skewed to the coarse maps)
from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
from chainer_maskrcnn.utils.synthetic import make_batch
model = MaskRCNN(n_fg_class=80, device=dev)
chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
b = make_batch(100, 2, 1024, 1024, G=8)
args = [torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks')]
chain(*args, 1.0)
rois, levels, label = chain.mask_inputs
rois, levels = rois.clone(), levels.clone()
del chain, model
torch.cuda.empty_cache()
print('step RoIs per level', torch.bincount(levels.cpu().long(), minlength=5).tolist())
scales = [1 / 4., 1 / 8., 1 / 16., 1 / 32., 1 / 64.]
shapes = [(2, 1024 // s, 1024 // s, 256) for s in (4, 8, 16, 32, 64)]
for P in (7, 14):
    g = torch.randn((rois.shape[0], P, P, 256), device=dev)
    res = {}
    for v in (1, 2, 3):
        _hip.check(lib.mrcnn_roi_align_set_bwd_variant(v))
        gxs = [torch.full(s, float('nan'), device=dev) for s in shapes]
        hd.roi_align_fpn_bwd(g, gxs, rois, levels, P, scales, accumulate=False)
        hd.roi_align_fpn_bwd(g, gxs, rois, levels, P, scales, accumulate=True)
        res[v] = gxs
    torch.cuda.synchronize()
    print('P=%d  max |v1 - v2| per level (after overwrite + accumulate):' % P, ['%.2e' % (a - b_).abs().max().item() for a, b_ in zip(res[1], res[2])],
          'scale %.2e' % max(a.abs().max().item() for a in res[1]), 'v3 == v2 bitwise:', all(torch.equal(a, b_) for a, b_ in zip(res[2], res[3])))
    for v in (1, 2, 3):
        _hip.check(lib.mrcnn_roi_align_set_bwd_variant(v))
        gxs = res[v]
        t_o = timed(lambda: hd.roi_align_fpn_bwd(g, gxs, rois, levels, P, scales, accumulate=False), n=20, rounds=3)
        t_a = timed(lambda: hd.roi_align_fpn_bwd(g, gxs, rois, levels, P, scales, accumulate=True), n=20, rounds=3)
        print('P=%2d variant %d: overwrite %.1f us   accumulate %.1f us' % (P, v, t_o[0], t_a[0]))
_hip.check(lib.mrcnn_roi_align_set_bwd_variant(2))
