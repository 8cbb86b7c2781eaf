"""Where does the hardware put the workgroups of a launch shaped like the ROIAlign backward (tools/roi_balance_model.py assumes: block b
-> XCD b % 8, then round robin over the XCD's 32 CUs)?  usage: python tools/dispatch_census.py [nblocks]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import numpy as np, torch
from chainer_maskrcnn._hip import lib as _lib, check, ptr, stream_ptr
lib = _lib()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 850
for rep in range(3):
    out = torch.zeros((2 * nb,), dtype=torch.int64, device='cuda:0')
    check(lib.mrcnn_debug_dispatch_census(ptr(out), nb, 30, stream_ptr()))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    hw, t0 = o[:nb], o[nb:]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 1) << 4) | (((hwid >> 13) & 7) << 5)       # cu_id, sh_id, se_id within the XCD
    b = np.arange(nb)
    same_xcd = np.mean(xcc == xcc[b % 8][...] ) if False else None
    # is XCD a function of b % 8?
    by_mod = [np.unique(xcc[b % 8 == m]) for m in range(8)]
    print('rep %d: XCC ids per (block %% 8): %s' % (rep, [list(u) for u in by_mod]))
    # within one residue class: the sequence of CU ids in block order
    m0 = b % 8 == 0
    seq = cu[m0]
    print('  CU sequence of blocks 0, 8, 16, ... (first 40):', list(seq[:40]))
    per = [np.sum((xcc == x) & (cu == c)) for x in np.unique(xcc) for c in np.unique(cu[xcc == x])]
    print('  blocks per CU: min %d max %d, distinct (xcc, cu) pairs %d; start spread %.1f us' % (min(per), max(per), len(per), (t0.max() - t0.min()) / 100.0))
    # does CU repeat with period 32 within a residue class?
    j = np.arange(len(seq))
    period_ok = np.mean(seq[j % 32 == (j % 32)] == seq[(j % 32)]) if len(seq) > 32 else None
    print('  fraction of blocks j (within the class) on the same CU as block j %% 32: %.2f' % float(np.mean(seq == seq[j % 32])))
