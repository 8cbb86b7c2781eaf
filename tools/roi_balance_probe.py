"""How much of the lean ROIAlign backward's time is load imbalance?  configs[1]'s shapes with (a) its own RoIs, (b) 512 equal RoIs on a uniform
grid with about the same number of list entries: same bytes, same arithmetic, every patch list about the same length."""
import ctypes, os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
dev = torch.device('cuda:0')
lib = _hip.lib()


def timed(f, n=50, rounds=7):
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
R, P = gy.shape[0], 7
side = float(sys.argv[1]) if len(sys.argv) > 1 else 176.0
cy, cx = np.meshgrid((np.arange(16) + 0.5) * (H * 4 / 16), (np.arange(32) + 0.5) * (W * 4 / 32), indexing='ij')
uni = np.stack([np.zeros(512), cy.ravel() - side / 2, cx.ravel() - side / 2, cy.ravel() + side / 2, cx.ravel() + side / 2], 1).astype(np.float32)
uni[:, 1] = uni[:, 1].clip(0, H * 4 - 1); uni[:, 3] = uni[:, 3].clip(0, H * 4 - 1); uni[:, 2] = uni[:, 2].clip(0, W * 4 - 1); uni[:, 4] = uni[:, 4].clip(0, W * 4 - 1)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
gx = torch.empty((N, H, W, C), device=dev)
Hs, Ws, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, P, P, 0)
arr = (ctypes.c_void_p * 1)(gx.data_ptr())
for name, rois in (('configs[1] RoIs', yx), ('uniform grid, %.0f-pixel RoIs' % side, uni)):
    rois_xy = torch.from_numpy(rois[:, [0, 2, 1, 4, 3]].copy()).to(dev)
    plan = torch.zeros((pb,), dtype=torch.uint8, device=dev)
    _hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, _hip.ptr(plan), pb, _hip.stream_ptr()))
    torch.cuda.synchronize()
    hdr = plan[:256].view(torch.int32).cpu().numpy()
    nodes_off = int(hdr[12]) * 4
    stride = 4096          # plan_node_stride<112>(): 64-byte header + 112 x 36 bytes
    units = int(hdr[3])
    pi = plan.view(torch.int32)

    cnt = pi[nodes_off // 4:nodes_off // 4 + units * (stride // 4)].view(units, stride // 4)[:, 0].cpu().numpy().astype(np.int64)
    tile = cnt.reshape(-1, 4).sum(1)
    print('%s: entries %d; per patch mean %.1f max %d p99 %d; per tile mean %.1f max %d; pool nodes used %d' % (name, cnt.sum(), cnt.mean(), cnt.max(), np.percentile(cnt, 99), tile.mean(), tile.max(), hdr[2]))

    def planned_v():
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), arr, Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, None, 0,
                                                           _hip.ptr(plan), pb, 1, _hip.stream_ptr()))
    for v, lab in ((8, 'lean 8/8, 256 threads'), (8 + 256, 'no gy loads'), (8 + 768, 'no gy loads, no gx stores')):
        _hip.check(lib.mrcnn_debug_roi_align_lean_variant(v))
        mn, med = timed(planned_v)
        print('    %-28s min %.1f us  median %.1f us' % (lab, mn, med))
    _hip.check(lib.mrcnn_debug_roi_align_lean_variant(8))
