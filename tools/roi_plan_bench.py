"""ROIAlign backward on BASELINE configs[1]: the fused wave kernel against plan + lean backward (ABI v9), interleaved in one process."""
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
for P in (7, 14):
    R = gy.shape[0]
    rs = np.random.RandomState(2)
    g = gy if P == 7 else rs.standard_normal((R, C, P, P)).astype(np.float32)
    rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
    gyt = torch.from_numpy(g).to(dev).contiguous(memory_format=torch.channels_last)
    gx = [torch.empty((N, H, W, C), device=dev) for _ in range(2)]
    Hs, Ws, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
    pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, P, P, 0)
    plan = torch.zeros((pb,), dtype=torch.uint8, device=dev)
    algo = 4 * (N * C * H * W + R * C * P * P) + 20 * R
    arr = [(ctypes.c_void_p * 1)(t.data_ptr()) for t in gx]
    lev0 = torch.zeros((R,), dtype=torch.int32, device=dev)

    def fused():
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_f32(_hip.ptr(gyt), arr[0], Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), _hip.ptr(lev0), R, P, P, 2, 0, None, 0, _hip.stream_ptr()))

    def build():
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, _hip.ptr(plan), pb, _hip.stream_ptr()))

    def planned(verified=0):
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), arr[1], Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, None, 0,
                                                           _hip.ptr(plan), pb, verified, _hip.stream_ptr()))

    def planned_v():
        planned(1)
    for t in gx:
        t.fill_(float('nan'))
    fused(); build(); planned(); torch.cuda.synchronize()
    hdr = plan[:256].view(torch.int32).cpu().numpy()
    print('P=%d: plan bytes %.1f MB, header magic ok %s overflow %d pool nodes used %d units %d; planned == fused bitwise: %s, NaNs %d'
          % (P, pb / 1e6, hdr[0] == 0x4E504C4E, hdr[1], hdr[2], hdr[3], torch.equal(gx[0], gx[1]), int(torch.isnan(gx[1]).sum())))
    for v in (0, 2, 9):
        _hip.check(lib.mrcnn_debug_roi_align_lean_variant(v))
        gx[1].fill_(float('nan')); planned(1); torch.cuda.synchronize()
        print('   lean variant %d == fused bitwise: %s' % (v, torch.equal(gx[0], gx[1])))
    for rnd in range(2):
        for name, f, v in (('fused', fused, None), ('plan build', build, None), ('planned (default)', planned, 8), ('verified 10/8 512 thr', planned_v, 0), ('verified 8/8 512 thr', planned_v, 2),
                           ('verified 8/8 256 thr', planned_v, 8), ('verified 8/8 128 thr', planned_v, 9), ('v 256 thr plain st', planned_v, 8 + 16 * 256), ('v 256 thr no gy gx', planned_v, 8 + 3 * 256)):
            if v is not None:
                _hip.check(lib.mrcnn_debug_roi_align_lean_variant(v))
            mn, med = timed(f)
            print('configs[1] P=%d %-22s: min %.1f us  median %.1f us   %.0f GB/s algorithmic (%.3f of 8 TB/s)' % (P, name, mn, med, algo / mn / 1e3, algo / mn / 1e3 / 8000))
    _hip.check(lib.mrcnn_debug_roi_align_lean_variant(8))
