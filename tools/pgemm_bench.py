"""Plane GEMMs (csrc/planes_gemm.h) standalone: error against float64 and time, F and G kinds, on the Winograd GEMM shapes of the step.
usage: python tools/pgemm_bench.py [quick]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
import torch
from chainer_maskrcnn._hip import lib as _lib, check, ptr, stream_ptr
lib = _lib()
dev = torch.device('cuda:0')


def planes(x, r4=0):
    Rr, C = x.shape
    out = torch.empty((Rr * C * 3,), dtype=torch.int16, device=dev)
    check(lib.mrcnn_debug_split_planes_f32(ptr(x), ptr(out), Rr, C, r4, stream_ptr()))
    return out


def unsplit(pl, Rr, C):
    """P16 planes -> float64 sum of the three planes (exactness check of the split)."""
    v = pl.view(Rr, C // 16, 3, 16).to(torch.int32) & 0xffff
    f = (v << 16).view(torch.float32).double()
    return f.sum(dim=2).reshape(Rr, C)


def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run_f(nb, rows, K, N, bn, check_batches=(0,), kind=0):
    g = torch.Generator(device='cpu').manual_seed(nb * 131 + K)
    A = (torch.randn((nb * rows, K), generator=g) * torch.exp2(torch.randint(-6, 6, (nb * rows, 1), generator=g).float())).to(dev)
    B = (torch.randn((nb * N, K), generator=g) / K ** 0.5).to(dev)
    Ap, Bp = planes(A), planes(B)
    assert float((unsplit(Ap, nb * rows, K) - A.double()).abs().max()) == 0.0, 'split not exact'
    C = torch.full((nb * rows, N), float('nan'), device=dev)
    if kind >= 2:
        Ap = A
    if kind >= 3:
        Bp = planes(B, 1)
    call = lambda: check(lib.mrcnn_debug_planes_gemm(kind, ptr(Ap), ptr(Bp), ptr(C), nb * rows, N, K, rows, nb, 1, 256 if kind >= 3 else 128, bn, stream_ptr()))
    call(); torch.cuda.synchronize()
    err = 0.0
    for b in check_batches:
        ref = A[b * rows:(b + 1) * rows].double() @ B[b * N:(b + 1) * N].double().t()
        err = max(err, float((C[b * rows:(b + 1) * rows].double() - ref).abs().max() / ref.abs().max()))
    us = timeit(call)
    fl = 2.0 * nb * rows * K * N
    print('F%s nb %2d rows %6d K %4d N %4d tile xxxx%-3d: %7.1f us  %6.1f TF/s fp32-equivalent  %5.0f TF/s bf16 executed (%.2f of 2500)  err %.2e' % (
        {0: ' ', 2: 'a', 3: 'B', 4: 'P'}[kind], nb, rows, K, N, bn, us, fl / us * 1e-6, 6 * fl / us * 1e-6, 6 * fl / us * 1e-6 / 2500, err), flush=True)
    return us


def run_g(nb, rows, M, N, ks, bm, bn, check_batches=(0,), kind=1):
    g = torch.Generator(device='cpu').manual_seed(nb * 17 + M)
    A = (torch.randn((nb * rows, M), generator=g) * 1e-3).to(dev)
    B = torch.randn((nb * rows, N), generator=g).to(dev)
    Ap, Bp = (planes(A, 2), B) if kind == 5 else (planes(A), planes(B))
    C = torch.full((ks, M, nb, N), float('nan'), device=dev)
    call = lambda: check(lib.mrcnn_debug_planes_gemm(kind, ptr(Ap), ptr(Bp), ptr(C), M, N, rows, rows, nb, ks, bm, bn, stream_ptr()))
    call(); torch.cuda.synchronize()
    assert not torch.isnan(C).any(), 'unwritten output'
    Cs = C.double().sum(dim=0)
    err = 0.0
    for b in check_batches:
        ref = A[b * rows:(b + 1) * rows].double().t() @ B[b * rows:(b + 1) * rows].double()
        err = max(err, float((Cs[:, b, :] - ref).abs().max() / ref.abs().max()))
    us = timeit(call)
    fl = 2.0 * nb * rows * M * N
    print('G%s nb %2d rows %6d M %4d N %4d ksplit %2d tile %3dx%-3d: %7.1f us  %6.1f TF/s fp32-equivalent  %5.0f TF/s bf16 executed (%.2f of 2500)  err %.2e' % (
        'P' if kind == 5 else ' ', nb, rows, M, N, ks, bm, bn, us, fl / us * 1e-6, 6 * fl / us * 1e-6, 6 * fl / us * 1e-6 / 2500, err), flush=True)
    return us


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'small':
    # the direct 1x1 layers of the ResNet-50 at batch 2 x 1024^2 as plain GEMMs (rows = pixels): what would the 128-wide plane GEMMs do on
    # them?  (k_conv_igemm<.., 3> takes 36-46 us for every one of these: tools/gemm_only_profile.py 3,3,3)
    for rows, K, N in ((131072, 64, 256), (131072, 256, 64), (32768, 128, 512), (32768, 512, 128), (8192, 256, 1024), (8192, 1024, 256),
                       (2048, 512, 2048), (2048, 2048, 512), (2048, 1024, 2048)):
        for kind in (2, 0):
            for bn in (128, 64):
                if N % bn == 0:
                    run_f(1, rows, K, N, bn, kind=kind)
        if rows % 256 == 0 and N % 256 == 0:
            run_f(1, rows, K, N, 256, kind=4)
    sys.exit(0)

if __name__ == '__main__' and not (len(sys.argv) > 1 and sys.argv[1] == 'stamps'):
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    # edge shapes first (correctness): N not a multiple of the tile, 64-wide tiles, short K
    run_f(4, 128, 32, 96, 128, (0, 3)); run_f(4, 128, 64, 96, 64, (0, 3)); run_f(16, 256, 64, 64, 64, (0, 15)); run_f(3, 384, 512, 160, 128, (1, 2))
    run_f(4, 128, 32, 96, 128, (0, 3), 2); run_f(4, 128, 64, 96, 64, (0, 3), 2); run_f(16, 256, 64, 64, 64, (0, 15), 2); run_f(3, 384, 512, 160, 128, (1, 2), 2)
    run_f(4, 256, 32, 96, 256, (0, 3), 3); run_f(4, 256, 64, 256, 256, (0, 3), 3); run_f(3, 512, 512, 160, 256, (1, 2), 3); run_f(2, 512, 128, 512, 256, (0, 1), 3)
    run_f(4, 256, 32, 96, 256, (0, 3), 4); run_f(4, 256, 64, 256, 256, (0, 3), 4); run_f(3, 512, 512, 160, 256, (1, 2), 4); run_f(2, 512, 128, 512, 256, (0, 1), 4)
    run_g(4, 128, 96, 160, 2, 256, 256, (0, 3), 5); run_g(3, 384, 256, 256, 3, 256, 256, (0, 2), 5); run_g(2, 512, 512, 256, 2, 256, 256, (0, 1), 5); run_g(2, 256, 256, 512, 1, 256, 256, (0, 1), 5)
    run_g(4, 128, 96, 160, 2, 128, 128, (0, 3)); run_g(4, 128, 64, 64, 1, 64, 64, (0, 3)); run_g(16, 256, 64, 256, 2, 64, 128, (0, 15)); run_g(3, 384, 256, 96, 3, 128, 64, (1, 2))
    if len(sys.argv) > 1 and sys.argv[1] == 'parts':
        from chainer_maskrcnn._hip import nn as hnn
        for nm, m in (('whole', 0), ('no MFMA', 1), ('L2-resident operands', 2), ('no epilogue', 4), ('no loads', 8), ('no loads, no epilogue', 12), ('L2 + no epilogue', 6), ('no MFMA, no epilogue', 5),
                      ('no MFMA, no loads, no epilogue', 13), ('no MFMA, no loads', 9), ('whole', 0)):
            check(lib.mrcnn_debug_conv_parts(m << 4))
            print('%-24s' % nm, end=' ')
            run_f(36, 8192, 256, 256, 128)
            print('%-24s' % nm, end=' ')
            run_f(36, 8192, 256, 256, 128, kind=2)
            print('%-24s' % nm, end=' ')
            run_f(36, 8192, 256, 256, 256, kind=3)
            print('%-24s' % nm, end=' ')
            run_f(36, 8192, 256, 256, 256, kind=4)
        check(lib.mrcnn_debug_conv_parts(0))
        for ks in (4, 7, 8):
            run_g(36, 8192, 256, 256, ks, 256, 256, kind=5)
        for nm, m in (('no MFMA', 1), ('no epilogue', 4), ('no loads', 8), ('no loads, no epilogue', 12), ('no MFMA, no epilogue', 5), ('no MFMA, no loads, no epilogue', 13)):
            check(lib.mrcnn_debug_conv_parts(m << 4)); print('%-24s' % nm, end=' '); run_g(36, 8192, 256, 256, 7, 256, 256, kind=5)
        check(lib.mrcnn_debug_conv_parts(0))
        # the in-kernel-split kernels on the same GEMM (GEMM launch only) and the float32 MFMA
        x = torch.randn((512, 14, 14, 256), device=dev); w = torch.randn((256, 3, 3, 256), device=dev) * 0.02
        hnn.set_winograd_pass_tiles(0, 0, 0)
        import torch.nn.functional as F
        xs, ws = x[:8].cpu(), w.cpu()
        y64 = F.conv2d(xs.double().permute(0, 3, 1, 2), ws.double().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        gy = torch.randn((512, 14, 14, 256), device=dev)
        gx64 = F.conv_transpose2d(gy[:8].cpu().double().permute(0, 3, 1, 2), ws.double().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
        for mode, big in ((0, 0), (3, 0), (3, 1)):
            check(lib.mrcnn_conv2d_set_split_operands(mode, mode, mode))
            check(lib.mrcnn_debug_conv_parts(0 if big else 0x100))
            y = hnn.conv2d_fwd_raw(x, w, None, 1, 1, False)
            gx = hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, 1)
            e1 = float((y[:8].double().cpu() - y64).abs().max() / y64.abs().max()); e2 = float((gx[:8].double().cpu() - gx64).abs().max() / gx64.abs().max())
            t_all = timeit(lambda: hnn.conv2d_fwd_raw(x, w, None, 1, 1, False)); t_bd = timeit(lambda: hnn.conv2d_bwd_data_raw(gy, w, tuple(x.shape), 1, 1))
            check(lib.mrcnn_conv2d_set_debug_skip(2))
            t_g = timeit(lambda: hnn.conv2d_fwd_raw(x, w, None, 1, 1, False))
            check(lib.mrcnn_conv2d_set_debug_skip(0))
            print('conv 512x14x14x256->256, split mode %d, plane GEMM %d: fwd whole call %.1f us (GEMM launch only %.1f us), bwd-data %.1f us; err vs fp64 fwd %.2e bwd-data %.2e' % (mode, big, t_all, t_g, t_bd, e1, e2))
        check(lib.mrcnn_conv2d_set_split_operands(0, 0, 0)); check(lib.mrcnn_debug_conv_parts(0))
        sys.exit(0)
    if not quick:
        # mask head 512 x 14 x 14 x 256 -> 256, F(4x4): 36 GEMMs of 8192 x 256 x 256; FPN 2 x 256^2 the same; F(2x2) ResNet / small maps
        run_f(36, 8192, 256, 256, 128); run_f(36, 8192, 256, 256, 64)
        run_f(36, 8192, 256, 256, 128, kind=2); run_f(36, 8192, 256, 256, 64, kind=2); run_f(36, 2048, 256, 256, 128, kind=2); run_f(16, 32768, 64, 64, 64, kind=2); run_f(16, 8192, 128, 128, 128, kind=2); run_f(36, 512, 512, 512, 128, kind=2)
        run_f(36, 2048, 256, 256, 128); run_f(16, 32768, 64, 64, 64); run_f(16, 8192, 128, 128, 128); run_f(36, 512, 512, 512, 128)
        for ks in (2, 4, 8, 16):
            run_g(36, 8192, 256, 256, ks, 128, 128)
        run_g(36, 2048, 256, 256, 2, 128, 128); run_g(36, 8192, 64, 64, 8, 64, 64); run_g(36, 2048, 128, 128, 4, 128, 128)


def stamps_report(kind=0, dbg=0):
    """Per-workgroup s_memtime / s_memrealtime / hardware ids of one launch of the mask-head F GEMM: concurrency per CU, clock."""
    import numpy as np
    nb, rows, K, N = 36, 8192, 256, 256
    A = torch.randn((nb * rows, K), device=dev); B = torch.randn((nb * N, K), device=dev) / 16
    Ap, Bp = (A if kind >= 2 else planes(A)), planes(B, 1 if kind >= 3 else 0)
    C = torch.empty((nb * rows, N), device=dev)
    T = 256 if kind >= 3 else 128
    nwg = (nb * rows // T) * (N // T)
    st = torch.zeros((nwg, 5), dtype=torch.int64, device=dev)
    check(lib.mrcnn_debug_conv_parts(dbg << 4))
    for it in range(3):
        check(lib.mrcnn_debug_planes_gemm_stamps(ptr(st) if it == 2 else None))
        check(lib.mrcnn_debug_planes_gemm(kind, ptr(Ap), ptr(Bp), ptr(C), nb * rows, N, K, rows, nb, 1, T, T, stream_ptr()))
    torch.cuda.synchronize()
    check(lib.mrcnn_debug_planes_gemm_stamps(None)); check(lib.mrcnn_debug_conv_parts(0))
    s = st.cpu().numpy().astype(np.int64)
    s = s[s[:, 3] > 0]
    nwg = len(s)
    t0, t1, r0, r1, hw = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4]
    hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
    cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 0x1) << 4) | (((hwid >> 13) & 0x7) << 5) | (xcc << 8)
    dur = (r1 - r0) / 100.0          # us (100 MHz)
    clk = (t1 - t0) / np.maximum(r1 - r0, 1) * 100.0     # MHz
    span = (r1.max() - r0.min()) / 100.0
    ucu = np.unique(cu)
    conc = []
    for c_ in ucu[:64]:
        m = cu == c_
        ev = sorted([(a, 1) for a in r0[m]] + [(b, -1) for b in r1[m]])
        cur = 0; last = ev[0][0]; acc = 0.0
        for tt, d in ev:
            acc += cur * (tt - last); last = tt; cur += d
        conc.append(acc / max(1, ev[-1][0] - ev[0][0]))
    print('kind %d dbg %d: %d workgroups on %d distinct CU ids, span %.1f us, workgroup duration median %.2f us (min %.2f max %.2f), clock median %.0f MHz, '
          'mean concurrent workgroups per CU %.2f, workgroups per CU %.1f' % (kind, dbg, nwg, len(ucu), span, np.median(dur), dur.min(), dur.max(), np.median(clk),
                                                                      float(np.mean(conc)), nwg / len(ucu)), flush=True)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'stamps':
    for kind in (4, 3):
        for dbg in (0, 1, 8, 4, 5, 12, 13):
            stamps_report(kind, dbg)
