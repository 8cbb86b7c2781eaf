"""EXPLORATORY: forward-kind GEMMs with three-term split-bf16 operands (mrcnn_conv2d_set_split_bf16) against the float32 MFMA
kernels: error of both against a float64 reference (max |err| / max |ref|) and the timings, per layer shape."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
import torch.nn.functional as F
from chainer_maskrcnn._hip import nn as hnn, lib as _lib, check
lib = _lib()
dev = torch.device('cuda:0')
SHAPES = [  # N, H, W, Cin, Cout, k, pad, fwd tile, reference on the CPU?
    (8, 14, 14, 256, 256, 3, 1, 0, True), (8, 14, 14, 256, 256, 3, 1, 2, True), (1, 64, 64, 64, 256, 1, 0, 2, True), (1, 32, 32, 1024, 256, 1, 0, 2, True),
    (512, 14, 14, 256, 256, 3, 1, 0, False), (2, 256, 256, 256, 256, 3, 1, 0, False), (2, 256, 256, 64, 256, 1, 0, 2, False), (2, 256, 256, 256, 64, 1, 0, 2, False),
    (2, 128, 128, 128, 512, 1, 0, 2, False), (2, 128, 128, 512, 128, 1, 0, 2, False), (2, 64, 64, 256, 1024, 1, 0, 2, False), (2, 64, 64, 1024, 256, 1, 0, 2, False),
    (2, 256, 256, 64, 64, 3, 1, 2, False), (2, 128, 128, 128, 128, 3, 1, 2, False), (2, 64, 64, 256, 256, 3, 1, 2, False), (512, 14, 14, 256, 384, 1, 0, 0, False)]


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
for (N, H, W, Ci, Co, k, p, ft, ref) in SHAPES:
    hnn.set_winograd_pass_tiles(ft, 0, 0)
    g = torch.Generator(device='cpu').manual_seed(N * 7 + Ci)
    x = torch.randn((N, H, W, Ci), generator=g); w = torch.randn((Co, k, k, Ci), generator=g) / (Ci * k * k) ** 0.5
    gy = torch.randn((N, H, W, Co), generator=g)
    xd, wd, gyd = x.to(dev), w.to(dev), gy.to(dev)
    out, tm = {}, {}
    for on in (0, 1, 2, 3):
        check(lib.mrcnn_conv2d_set_split_operands(on, on, on))
        out[on] = (hnn.conv2d_fwd_raw(xd, wd, None, 1, p, False), hnn.conv2d_bwd_data_raw(gyd, wd, tuple(xd.shape), 1, p),
                   hnn.conv2d_bwd_filter_raw(xd, gyd, tuple(wd.shape), 1, p, False)[0])
        tm[on] = (timeit(lambda: hnn.conv2d_fwd_raw(xd, wd, None, 1, p, False)), timeit(lambda: hnn.conv2d_bwd_data_raw(gyd, wd, tuple(xd.shape), 1, p)),
                  timeit(lambda: hnn.conv2d_bwd_filter_raw(xd, gyd, tuple(wd.shape), 1, p, False)))
        if not ref: tot[on] += sum(tm[on])
    err = ''
    if ref:
        y64 = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, 1, p).permute(0, 2, 3, 1)
        gx64 = F.conv_transpose2d(gy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), None, 1, p).permute(0, 2, 3, 1)
        rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
        gw64 = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Co, Ci, k, k), gy.double().permute(0, 3, 1, 2), 1, p).permute(0, 2, 3, 1)
        err = '  err vs fp64 (f32 / bf16x3 / f16x3 / bf16x6): fwd %.1e %.1e %.1e %.1e, bwd_data %.1e %.1e %.1e %.1e, bwd_filter %.1e %.1e %.1e %.1e' % (
            rel(out[0][0], y64), rel(out[1][0], y64), rel(out[2][0], y64), rel(out[3][0], y64), rel(out[0][1], gx64), rel(out[1][1], gx64), rel(out[2][1], gx64), rel(out[3][1], gx64),
            rel(out[0][2], gw64), rel(out[1][2], gw64), rel(out[2][2], gw64), rel(out[3][2], gw64))
    d = float((out[0][0] - out[1][0]).abs().max() / out[0][0].abs().max())
    d2 = float((out[0][0] - out[2][0]).abs().max() / out[0][0].abs().max())
    d3 = float((out[0][0] - out[3][0]).abs().max() / out[0][0].abs().max())
    print('%4dx%3dx%3d %5d->%5d k%d tile %d: fwd %6.1f -> %6.1f / %6.1f / %6.1f us  bwd_data %6.1f -> %6.1f / %6.1f / %6.1f us  bwd_filter %6.1f -> %6.1f / %6.1f / %6.1f us  |x - f32| bf16x3 %.1e f16x3 %.1e bf16x6 %.1e%s' % (N, H, W, Ci, Co, k, ft, tm[0][0], tm[1][0], tm[2][0], tm[3][0], tm[0][1], tm[1][1], tm[2][1], tm[3][1], tm[0][2], tm[1][2], tm[2][2], tm[3][2], d, d2, d3, err), flush=True)
check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0))
hnn.set_winograd_pass_tiles(2, 0, 0)
print('sum over the timed shapes: f32 %.1f us, bf16x3 %.1f us, f16x3 %.1f us, bf16x6 %.1f us' % (tot[0], tot[1], tot[2], tot[3]))
