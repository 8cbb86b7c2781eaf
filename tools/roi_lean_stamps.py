"""Per-wave s_memtime stamps of the lean ROIAlign backward on configs[1] (mrcnn_debug_roi_align_lean_stamps): where does a launch's time go -
dispatch stagger, the first round trip, cycles per list entry as a function of what shares the SIMD, the store tail?"""
import ctypes, os, sys
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
import torch
from chainer_maskrcnn import _hip
from chainer_maskrcnn.utils.synthetic import config2_inputs
dev = torch.device('cuda:0')
lib = _hip.lib()
x, yx, gy = config2_inputs()
N, C, H, W = x.shape
R, P = gy.shape[0], 7
rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
gx = torch.empty((N, H, W, C), device=dev)
Hs, Ws, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, P, P, 0)
arr = (ctypes.c_void_p * 1)(gx.data_ptr())
plan = torch.zeros((pb,), dtype=torch.uint8, device=dev)
_hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, _hip.ptr(plan), pb, _hip.stream_ptr()))
tiles = ((H + 7) // 8) * ((W + 7) // 8) * N
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def planned_v():
    _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), arr, Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, P, P, 2, 0, None, 0,
                                                       _hip.ptr(plan), pb, 1, _hip.stream_ptr()))


_hip.check(lib.mrcnn_debug_roi_align_lean_variant(variant))
for _ in range(5):
    planned_v()
torch.cuda.synchronize()
st = torch.zeros((tiles * 8, 10), dtype=torch.int64, device=dev)
_hip.check(lib.mrcnn_debug_roi_align_lean_stamps(ctypes.c_void_p(st.data_ptr())))
planned_v()
torch.cuda.synchronize()
_hip.check(lib.mrcnn_debug_roi_align_lean_stamps(None))
_hip.check(lib.mrcnn_debug_roi_align_lean_variant(8))
s = st.cpu().numpy()
if len(sys.argv) > 2:
    np.save(sys.argv[2], s)           # raw stamps (tile-major, 8 waves per tile, 10 x u64) for offline analysis
live = s[:, 0] != 0
s = s[live]
t0, t1, t2, t3, n, hw = (s[:, i] for i in range(6))
# s_memtime differs between parts of the chip: durations of one wave come from it (2400 ticks per us), positions in the launch from
# s_memrealtime (100 MHz, one counter for the chip)
r0, r1 = s[:, 6], s[:, 7]
rbase = r0.min()
MHZ = 2400.0          # s_memtime ticks per microsecond on this part (tools/ubench/pk_rate: 2402 - 2407)
us = lambda c: np.asarray(c, dtype=np.float64) / MHZ
print('waves %d (of %d slots), entries %d; first wave\'s entry to last wave\'s end: %.2f us' % (len(s), tiles * 8, n.sum(), (r1.max() - rbase) / 100.0))
print('wave start after the first wave: median %.2f us  p90 %.2f  max %.2f' % (np.median(r0 - rbase) / 100.0, np.percentile(r0 - rbase, 90) / 100.0, (r0 - rbase).max() / 100.0))
tA, tB = s[:, 8], s[:, 9]
print('   of it: kernel arguments back after %.2f us (median; p90 %.2f), the node\'s and the header\'s scalar loads after another %.2f (p90 %.2f), the row indices (vector load) after another %.2f (p90 %.2f)' % (
    us(np.median(tA - t0)), us(np.percentile(tA - t0, 90)), us(np.median(tB - tA)), us(np.percentile(tB - tA, 90)), us(np.median(t1 - tB)), us(np.percentile(t1 - tB, 90))))
print('first round trip (entry -> first node\'s loads back): median %.2f us  p90 %.2f  max %.2f' % (us(np.median(t1 - t0)), us(np.percentile(t1 - t0, 90)), us((t1 - t0).max())))
print('entry loop: median %.2f us  p90 %.2f  max %.2f;  per entry of a wave: median %.0f ticks  (waves with >= 8 entries)' % (
    us(np.median(t2 - t1)), us(np.percentile(t2 - t1, 90)), us((t2 - t1).max()), np.median(((t2 - t1) / np.maximum(n, 1))[n >= 8])))
print('stores acknowledged: median %.2f us  p90 %.2f  max %.2f' % (us(np.median(t3 - t2)), us(np.percentile(t3 - t2, 90)), us((t3 - t2).max())))
print('wave end after the first wave\'s start: median %.2f us  p90 %.2f  max %.2f' % (np.median(r1 - rbase) / 100.0, np.percentile(r1 - rbase, 90) / 100.0, (r1 - rbase).max() / 100.0))
late = (r0 - rbase) >= np.percentile(r0 - rbase, 90)
print('the last tenth of the waves to start: start %.2f us, first round trip %.2f us, loop %.2f us (%.1f entries), stores %.2f us (medians)' % (
    np.median((r0 - rbase)[late]) / 100.0, us(np.median((t1 - t0)[late])), us(np.median((t2 - t1)[late])), np.median(n[late]), us(np.median((t3 - t2)[late]))))
last = np.argsort(-(r1 - rbase))[:5]
for o in last:
    print('   last waves to end: end %.2f us; started %.2f, first round trip %.2f, loop %.2f (%d entries), stores %.2f' % (
        (r1[o] - rbase) / 100.0, (r0[o] - rbase) / 100.0, us(t1[o] - t0[o]), us(t2[o] - t1[o]), n[o], us(t3[o] - t2[o])))
# who shares a SIMD?  HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID low bits of the high word
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 32) & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
skey = key * 4 + simd
print('distinct CUs %d, distinct SIMDs %d; waves per SIMD: median %d max %d; entries per SIMD: median %d max %d' % (
    len(set(key)), len(set(skey)), np.median(np.bincount(np.unique(skey, return_inverse=True)[1])), np.bincount(np.unique(skey, return_inverse=True)[1]).max(),
    np.median(np.bincount(np.unique(skey, return_inverse=True)[1], weights=n)), np.bincount(np.unique(skey, return_inverse=True)[1], weights=n).max()))
inv = np.unique(skey, return_inverse=True)[1]
ent = np.bincount(inv, weights=n)
end = np.zeros(len(ent)); np.maximum.at(end, inv, (r1 - rbase).astype(np.float64))
start = np.full(len(ent), 1e18); np.minimum.at(start, inv, (r0 - rbase).astype(np.float64) + us(t1 - t0) * 100.0)
busy = (end - start) / 100.0
print('per SIMD: (last store - first loop start) median %.2f us max %.2f; correlation with its entries %.2f; ticks per entry median %.0f' % (
    np.median(busy), busy.max(), np.corrcoef(ent, busy)[0, 1], np.median(busy * MHZ / np.maximum(ent, 1))))
order = np.argsort(-busy)[:5]
for o in order:
    print('   slowest SIMD: %.2f us, %d entries, %d waves' % (busy[o], ent[o], (inv == o).sum()))
