"""Oracle: ROIAlign forward / backward (TEST INFRASTRUCTURE - see oracle/__init__.py).

Reference call sites (relative to /root/reference):
  chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7   column shim (idx,y1,x1,y2,x2)->(idx,x1,y1,x2,y2)
  chainer_maskrcnn/model/head/fpn_roi_mask_head.py:59-61,75-77,93-94   per-RoI calls, 7x7 and 14x14
  chainer_maskrcnn/model/head/fpn_roi_keypoint_head.py:63-68,85-86,102-103

The operator body lives in the git submodule katotetsuro/roi_align (.gitmodules:1-3),
which is ABSENT from /root/reference (empty directory, SHA unrecoverable).
=> PARITY UNPINNED for the arithmetic.  The algorithm restated here is the published
Caffe2/Detectron ``RoIAlign`` (legacy, non-"aligned" variant; identical to Chainer's
later ``F.roi_average_align_2d``), SURVEY.md Appendix A-1, evaluated in float32 with
one rounding per written operation and NO fused multiply-add:

    x1f=x1*s, y1f=y1*s, x2f=x2*s, y2f=y2*s
    rw=max(x2f-x1f,1), rh=max(y2f-y1f,1);  bw=rw/PW, bh=rh/PH
    gh = sr>0 ? sr : ceil(rh/PH)   (gw likewise)
    y(ph,iy) = (y1f + ph*bh) + ((iy+0.5)*bh)/gh      x likewise
    sample is void (contributes 0) if y<-1 or y>H or x<-1 or x>W
    y=max(y,0); yl=(int)y; if yl>=H-1: yh=yl=H-1, y=yl  else yh=yl+1
    ly=y-yl, hy=1-ly (x likewise)
    v = hy*hx*f[yl,xl] + hy*lx*f[yl,xh] + ly*hx*f[yh,xl] + ly*lx*f[yh,xh]
    out = (sum over iy (outer), ix (inner) of v) / (gh*gw)
    backward: gx[cell] += (gy*w)/(gh*gw) for the same four cells of every non-void sample.

Deliberate pin: the reference signature has no ``sampling_ratio``; the Python shim
defaults to 2 (Mask R-CNN paper / Detectron configs); 0 selects the adaptive grid.
"""
import numpy as np

F = np.float32


def roi_geometry(roi, H, W, outh, outw, spatial_scale, sampling_ratio):
    """Per-axis sample tables of ONE RoI ``(idx, x1, y1, x2, y2)``.

    Returns (gh, gw, ytab, xtab); each tab is a dict of arrays over k = p*grid+i:
      valid (bool), lo, hi (int32 cell index, -1 if void), wl, wh (float32 weights of
      the lo / hi cell, i.e. ``hy`` and ``ly`` of the formula above).
    """
    s = F(spatial_scale)
    x1f = F(roi[1]) * s
    y1f = F(roi[2]) * s
    x2f = F(roi[3]) * s
    y2f = F(roi[4]) * s
    rw = np.maximum(x2f - x1f, F(1))
    rh = np.maximum(y2f - y1f, F(1))
    bw = F(rw / F(outw))
    bh = F(rh / F(outh))
    if sampling_ratio > 0:
        gh = gw = int(sampling_ratio)
    else:
        gh = int(np.ceil(F(rh / F(outh))))
        gw = int(np.ceil(F(rw / F(outw))))

    def axis(start, bin_sz, P, grid, size):
        p = np.repeat(np.arange(P), grid).astype(F)
        i = np.tile(np.arange(grid), P).astype(F)
        c = (start + p * bin_sz) + ((i + F(0.5)) * bin_sz) / F(grid)
        c = c.astype(F)
        valid = ~((c < F(-1)) | (c > F(size)))
        c = np.maximum(c, F(0))
        lo = c.astype(np.int32)  # truncation == floor for c >= 0
        edge = lo >= size - 1
        lo = np.where(edge, size - 1, lo).astype(np.int32)
        hi = np.where(edge, size - 1, lo + 1).astype(np.int32)
        c = np.where(edge, lo.astype(F), c).astype(F)
        wh = (c - lo.astype(F)).astype(F)      # weight of the hi cell  (ly / lx)
        wl = (F(1) - wh).astype(F)             # weight of the lo cell  (hy / hx)
        lo = np.where(valid, lo, -1).astype(np.int32)
        hi = np.where(valid, hi, -1).astype(np.int32)
        return dict(valid=valid, lo=lo, hi=hi, wl=wl, wh=wh)

    return gh, gw, axis(y1f, bh, outh, gh, H), axis(x1f, bw, outw, gw, W)


def roi_align_fwd(x, rois, outh, outw, spatial_scale, sampling_ratio=2):
    """x (N,C,H,W) f32, rois (R,5) f32 ``(idx,x1,y1,x2,y2)`` -> y (R,C,outh,outw) f32."""
    x = np.asarray(x, dtype=F)
    rois = np.asarray(rois, dtype=F)
    N, C, H, W = x.shape
    R = rois.shape[0]
    y = np.zeros((R, C, outh, outw), dtype=F)
    for r in range(R):
        n = int(rois[r, 0])
        gh, gw, yt, xt = roi_geometry(rois[r], H, W, outh, outw, spatial_scale, sampling_ratio)
        valid = yt['valid'][:, None] & xt['valid'][None, :]
        yl = np.maximum(yt['lo'], 0); yh = np.maximum(yt['hi'], 0)
        xl = np.maximum(xt['lo'], 0); xh = np.maximum(xt['hi'], 0)
        f = x[n]
        w1 = yt['wl'][:, None] * xt['wl'][None, :]
        w2 = yt['wl'][:, None] * xt['wh'][None, :]
        w3 = yt['wh'][:, None] * xt['wl'][None, :]
        w4 = yt['wh'][:, None] * xt['wh'][None, :]
        v = (w1 * f[:, yl][:, :, xl] + w2 * f[:, yl][:, :, xh]
             + w3 * f[:, yh][:, :, xl] + w4 * f[:, yh][:, :, xh])
        v = np.where(valid[None], v, F(0)).astype(F)
        v = v.reshape(C, outh, gh, outw, gw)
        acc = np.zeros((C, outh, outw), dtype=F)
        for iy in range(gh):
            for ix in range(gw):
                acc += v[:, :, iy, :, ix]
        y[r] = acc / F(gh * gw)
    return y


def roi_align_bwd(gy, rois, x_shape, spatial_scale, sampling_ratio=2):
    """gy (R,C,outh,outw) -> gx (N,C,H,W): exact adjoint of :func:`roi_align_fwd`."""
    gy = np.asarray(gy, dtype=F)
    rois = np.asarray(rois, dtype=F)
    N, C, H, W = x_shape
    R, _, outh, outw = gy.shape
    gx = np.zeros((N, C, H, W), dtype=F)
    for r in range(R):
        n = int(rois[r, 0])
        gh, gw, yt, xt = roi_geometry(rois[r], H, W, outh, outw, spatial_scale, sampling_ratio)
        cnt = F(gh * gw)
        vy = np.nonzero(yt['valid'])[0]
        vx = np.nonzero(xt['valid'])[0]
        if vy.size == 0 or vx.size == 0:
            continue
        g = gy[r][:, vy // gh][:, :, vx // gw]          # (C, SY, SX)
        for (ya, wy) in ((yt['lo'][vy], yt['wl'][vy]), (yt['hi'][vy], yt['wh'][vy])):
            for (xa, wx) in ((xt['lo'][vx], xt['wl'][vx]), (xt['hi'][vx], xt['wh'][vx])):
                w = wy[:, None] * wx[None, :]
                contrib = ((g * w[None]) / cnt).astype(F)
                # dense per-RoI patch, then one add: same sums as a per-tap scatter
                y0, x0 = int(ya.min()), int(xa.min())
                patch = np.zeros((C, int(ya.max()) - y0 + 1, int(xa.max()) - x0 + 1), dtype=F)
                # rows first (np.add.at keeps duplicate indices), then columns
                tmp = np.zeros((C, patch.shape[1], contrib.shape[2]), dtype=F)
                np.add.at(tmp, (slice(None), ya - y0), contrib)
                np.add.at(patch, (slice(None), slice(None), xa - x0), tmp)
                gx[n, :, y0:y0 + patch.shape[1], x0:x0 + patch.shape[2]] += patch
    return gx


def roi_align_sample_tables(rois, H, W, outh, outw, spatial_scale, sampling_ratio, smax):
    """Integer corner indices and float weights of every sample, padded to ``smax``.

    Returns (cnt (R,2) i32 = [PH*gh, PW*gw], idx (R,2,smax,2) i32, wgt (R,2,smax,2) f32)
    with axis 0 = y, axis 1 = x; idx = (lo, hi) or (-1,-1) for a void sample,
    (-2,-2) beyond cnt; wgt = (wl, wh), 0 for void samples and beyond cnt.  This is the "indices
    bit-exact" contract of BASELINE.json's north_star.
    """
    rois = np.asarray(rois, dtype=F)
    R = rois.shape[0]
    cnt = np.zeros((R, 2), np.int32)
    idx = np.full((R, 2, smax, 2), -2, np.int32)
    wgt = np.zeros((R, 2, smax, 2), F)
    for r in range(R):
        gh, gw, yt, xt = roi_geometry(rois[r], H, W, outh, outw, spatial_scale, sampling_ratio)
        for a, (t, k) in enumerate(((yt, outh * gh), (xt, outw * gw))):
            cnt[r, a] = k
            m = min(k, smax)
            idx[r, a, :m, 0] = t['lo'][:m]
            idx[r, a, :m, 1] = t['hi'][:m]
            wgt[r, a, :m, 0] = np.where(t['valid'][:m], t['wl'][:m], F(0))
            wgt[r, a, :m, 1] = np.where(t['valid'][:m], t['wh'][:m], F(0))
    return cnt, idx, wgt


def roi_align_2d_yx(x, indices_and_rois, outh, outw, spatial_scale, sampling_ratio=2):
    """Restates chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7 (column permutation)."""
    xy = np.asarray(indices_and_rois, dtype=F)[:, [0, 2, 1, 4, 3]]
    return roi_align_fwd(x, xy, outh, outw, spatial_scale, sampling_ratio)


def roi_align_fwd_scalar(x, rois, outh, outw, spatial_scale, sampling_ratio=2):
    """Literal per-sample scalar loop of the formula in the module docstring.

    Slow; used only to pin :func:`roi_align_fwd` (the vectorised form) on small cases.
    """
    x = np.asarray(x, dtype=F)
    rois = np.asarray(rois, dtype=F)
    N, C, H, W = x.shape
    R = rois.shape[0]
    y = np.zeros((R, C, outh, outw), dtype=F)
    s = F(spatial_scale)
    for r in range(R):
        n = int(rois[r, 0])
        x1f = rois[r, 1] * s; y1f = rois[r, 2] * s
        x2f = rois[r, 3] * s; y2f = rois[r, 4] * s
        rw = max(F(x2f - x1f), F(1)); rh = max(F(y2f - y1f), F(1))
        bw = F(rw / F(outw)); bh = F(rh / F(outh))
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(F(rh / F(outh))))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(F(rw / F(outw))))
        for c in range(C):
            f = x[n, c]
            for ph in range(outh):
                for pw in range(outw):
                    acc = F(0)
                    for iy in range(gh):
                        yy = F(F(y1f + F(F(ph) * bh)) + F(F(F(iy) + F(0.5)) * bh) / F(gh))
                        for ix in range(gw):
                            xx = F(F(x1f + F(F(pw) * bw)) + F(F(F(ix) + F(0.5)) * bw) / F(gw))
                            if yy < -1 or yy > H or xx < -1 or xx > W:
                                continue
                            y_ = max(yy, F(0)); x_ = max(xx, F(0))
                            yl = int(y_); xl = int(x_)
                            if yl >= H - 1:
                                yh = yl = H - 1; y_ = F(yl)
                            else:
                                yh = yl + 1
                            if xl >= W - 1:
                                xh = xl = W - 1; x_ = F(xl)
                            else:
                                xh = xl + 1
                            ly = F(y_ - F(yl)); lx = F(x_ - F(xl))
                            hy = F(F(1) - ly); hx = F(F(1) - lx)
                            v = F(F(hy * hx) * f[yl, xl])
                            v = F(v + F(F(hy * lx) * f[yl, xh]))
                            v = F(v + F(F(ly * hx) * f[yh, xl]))
                            v = F(v + F(F(ly * lx) * f[yh, xh]))
                            acc = F(acc + v)
                    y[r, c, ph, pw] = F(acc / F(gh * gw))
    return y
