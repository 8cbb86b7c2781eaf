"""Box + mask head over FPN levels on gfx950 kernels.

Mirror of chainer_maskrcnn/model/head/fpn_roi_mask_head.py:11-102 (layers :24-49, forward :55-88,
predict_mask :90-102).  Kept: conv3x3 BEFORE the two FCs in the box branch (:65), class-agnostic
4-vector regressor (:28), no ReLU between deconv and the final 1x1 conv (:83).
Mechanism differences (results identical):
  * the two per-RoI Python loops (:59-61, :75-77) are ONE batched multi-level ROIAlign launch each;
  * ``cls_loc`` (4) and ``score`` (n_class) are one fused linear layer: output columns [0,n_class) =
    score, [LOC0, LOC0+4) = loc;
  * the 2x2/2 deconvolution is a 1x1 convolution to 4*256 channels + a pixel shuffle that adds the bias.
"""
import ctypes

import numpy as np
import torch

from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr, ops
from chainer_maskrcnn._hip.nn import workspace
from chainer_maskrcnn.nn.core import Conv, ParamStore, normal, pad_to


def _level_args(xs, scales):
    L = len(xs)
    arr_p = (ctypes.c_void_p * L)(*[x.data_ptr() for x in xs])
    Hs = (ctypes.c_int * L)(*[x.shape[1] for x in xs])
    Ws = (ctypes.c_int * L)(*[x.shape[2] for x in xs])
    sc = (ctypes.c_float * L)(*[float(s) for s in scales])
    return L, arr_p, Hs, Ws, sc


def roi_align_fpn_fwd(xs, rois_xy5, levels, out_size, scales, sampling_ratio=2):
    """xs: NHWC levels (N,H_l,W_l,C); rois_xy5 (R,5) (idx,x1,y1,x2,y2); levels (R,) int32 -> (R,P,P,C)."""
    _hip.require_cuda(rois_xy5, levels, *xs)
    L, arr_p, Hs, Ws, sc = _level_args(xs, scales)
    N, C = xs[0].shape[0], xs[0].shape[3]
    R = rois_xy5.shape[0]
    y = torch.empty((R, out_size, out_size, C), dtype=torch.float32, device=rois_xy5.device)
    nb = lib().mrcnn_roi_align_fwd_workspace_bytes(R)
    ws = workspace(nb, y.device) if nb else None          # the map-order permutation of the RoIs
    check(lib().mrcnn_roi_align_fpn_fwd_ws_f32(arr_p, Hs, Ws, sc, L, N, C, ptr(rois_xy5), ptr(levels), R, out_size,
                                               out_size, sampling_ratio, ptr(y), ptr(ws), ws.numel() if ws is not None else 0,
                                               stream_ptr()))
    return y


# OPT-IN for the training step: the backward's entry lists built beside the FORWARD pass (the RoIs are all they need:
# mrcnn_roi_align_fpn_bwd_plan_f32 on the weight-gradient stream, idle then) and the backward following them
# (mrcnn_roi_align_fpn_bwd_planned_f32: same bits).  On configs[1] the backward goes 27.6 -> 20.8 us (bench.py --workload roialign); in the
# step the two ROIAlign backward calls are 0.2 of 21.8 ms and the plan's launches and buffers (~100 MB each) eat the gain: same-process
# A/B 21.88 (on) against 21.82 ms (off), tools/ab_step.py - so the step keeps the fused backward.
PLAN_BWD_IN_FORWARD = False


def roi_align_fpn_bwd_plan(xs, rois_xy5, levels, out_size, scales, sampling_ratio=2, stream=None):
    """Entry-list plan of the ROIAlign backward for these RoIs over maps shaped like xs (NHWC levels), enqueued on `stream` (default: the
    current one).  Returns the plan buffer (uint8) for roi_align_fpn_bwd(..., plan=)."""
    L, _, Hs, Ws, sc = _level_args(xs, scales)
    N, C = xs[0].shape[0], xs[0].shape[3]
    R = rois_xy5.shape[0]
    nb = lib().mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, L, N, R, out_size, out_size, 1)
    if nb == 0:
        return None
    plan = torch.empty((nb,), dtype=torch.uint8, device=rois_xy5.device)
    st = _hip.raw_stream(rois_xy5.device.index) if stream is None else stream.cuda_stream
    check(lib().mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, L, N, C, ptr(rois_xy5), ptr(levels), R, out_size, out_size, sampling_ratio, 1,
                                                 ptr(plan), plan.numel(), ctypes.c_void_p(st)))
    return plan


def _plan_beside_forward(xs, rois_xy5, levels, out_size, scales):
    """The plan on the weight-gradient stream, behind everything enqueued on the current stream so far (the RoIs); returns (plan, event)."""
    if not PLAN_BWD_IN_FORWARD or not rois_xy5.is_cuda or rois_xy5.shape[0] == 0:
        return None
    from chainer_maskrcnn._hip import nn as hnn
    dev = rois_xy5.device
    main, side = torch.cuda.current_stream(dev), hnn.side_stream(dev)
    side.wait_stream(main)
    plan = roi_align_fpn_bwd_plan(xs, rois_xy5, levels, out_size, scales, stream=side)
    if plan is None:
        return None
    ev = torch.cuda.Event()
    ev.record(side)
    for t_ in (plan, rois_xy5, levels):
        if t_ is not None:
            t_.record_stream(side)
    return plan, ev


def roi_align_bwd_plan_status(plan):
    """(header valid, tiles flagged, pool nodes used) of a plan buffer - WAITS for the device (mrcnn_roi_align_bwd_plan_status).  A caller
    that found (True, 0, _) may pass verified=True to roi_align_fpn_bwd: the lean kernel alone, without the launch behind it."""
    st3 = (ctypes.c_int * 3)()
    check(lib().mrcnn_roi_align_bwd_plan_status(ptr(plan), plan.numel(), st3, stream_ptr()))
    return bool(st3[0]), int(st3[1]), int(st3[2])


def roi_align_fpn_bwd(gy, gxs, rois_xy5, levels, out_size, scales, accumulate, sampling_ratio=2, plan=None, verified=False):
    """plan: (buffer, event) of _plan_beside_forward / a buffer of roi_align_fpn_bwd_plan for the SAME rois, levels, maps: the lean backward."""
    L, arr_p, Hs, Ws, sc = _level_args(gxs, scales)
    N, C = gxs[0].shape[0], gxs[0].shape[3]
    nb = lib().mrcnn_roi_align_fpn_bwd_workspace_bytes(Hs, Ws, L, N, C, rois_xy5.shape[0], out_size, out_size, sampling_ratio)
    ws = workspace(nb, gy.device) if nb else None
    if plan is not None:
        if isinstance(plan, tuple):
            plan, ev = plan
            torch.cuda.current_stream(gy.device).wait_event(ev)
        check(lib().mrcnn_roi_align_fpn_bwd_planned_f32(ptr(gy), arr_p, Hs, Ws, sc, L, N, C, ptr(rois_xy5), ptr(levels), rois_xy5.shape[0], out_size,
                                                        out_size, sampling_ratio, int(accumulate), ptr(ws), ws.numel() if ws is not None else 0,
                                                        ptr(plan), plan.numel(), int(bool(verified)), stream_ptr()))
        return
    check(lib().mrcnn_roi_align_fpn_bwd_f32(ptr(gy), arr_p, Hs, Ws, sc, L, N, C, ptr(rois_xy5), ptr(levels),
                                            rois_xy5.shape[0], out_size, out_size, sampling_ratio, int(accumulate),
                                            ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()))


class FPNRoIMaskHead(object):
    mask_size = 28

    def __init__(self, n_class, roi_size_box, roi_size_mask, loc_initialW=None, score_initialW=None,
                 mask_initialW=None, ps=None, prefix='head', in_channels=256, fc_channels=1024,
                 n_mask_convs=4, mask_out_channels=None, upsample2x=False, mask_conv_names=None):
        self.ps = ps if ps is not None else ParamStore()
        self.n_class, self.roi_size_box, self.roi_size_mask = n_class, roi_size_box, roi_size_mask
        c = in_channels
        p = prefix + '/'
        self.conv1 = Conv(self.ps, p + 'conv1', c, c, 3, 1, 1, relu=True, fwd_tile=0)
        self.fc1 = Conv(self.ps, p + 'fc1', c * roi_size_box * roi_size_box, fc_channels, relu=True)
        self.fc2 = Conv(self.ps, p + 'fc2', fc_channels, fc_channels, relu=True)
        # fused [score | loc] linear layer
        self.LOC0 = pad_to(n_class, 8)
        self.out_p = pad_to(self.LOC0 + 4, 32)
        li = normal(0.001 if loc_initialW is None else loc_initialW)((4, 1, 1, fc_channels))
        si = normal(0.01 if score_initialW is None else score_initialW)((n_class, 1, 1, fc_channels))
        self.box_out = Conv(self.ps, p + 'score_cls_loc', fc_channels, n_class + 4, 1, cout_p=self.out_p,
                            cout_index=list(range(n_class)) + list(range(self.LOC0, self.LOC0 + 4)),
                            init=lambda shape: (lambda rs: np.concatenate([si(rs), li(rs)], 0)))
        mi = 0.01 if mask_initialW is None else mask_initialW
        names = mask_conv_names or ['mask%d' % (i + 1) for i in range(n_mask_convs)]
        self.mask_convs = [Conv(self.ps, p + nm, c, c, 3, 1, 1, relu=True, fwd_tile=0) for nm in names]
        self.mask_out_channels = (n_class - 1) if mask_out_channels is None else mask_out_channels
        self.upsample2x = upsample2x
        # deconv1: W (Cin, Cout, 2, 2) in Chainer == 1x1 conv weight ((a*2+b)*Cout + o, Cin); bias added by the shuffle
        self.deconv1 = Conv(self.ps, p + 'deconv1', c, 4 * c, 1, bias=False, init=normal(mi))
        self.ps.register(p + 'deconv1/b', (c,), lambda rs: np.zeros((c,), np.float32))
        self.deconv_b = p + 'deconv1/b'
        self.conv2 = Conv(self.ps, p + 'conv2', c, self.mask_out_channels, 1, init=normal(mi))
        self.channels = c
        # No non-linearity sits between deconv1 and conv2 in the reference (:83), so the two run as ONE 2x2/2
        # deconvolution to mask_out_channels with composed weights (include/mrcnn_hip.h, mrcnn_deconv_merge_*): same
        # function, same parameters and gradients, 4x fewer MACs on the 28x28 maps.  False = layer by layer.
        self.merge_deconv = (c % 8 == 0)

    # ---- forward --------------------------------------------------------------------------------
    def box_branch(self, xs, rois_xy5, levels, spatial_scales):
        pool = roi_align_fpn_fwd(xs, rois_xy5, levels, self.roi_size_box, spatial_scales)
        h, t1 = self.conv1.fwd(pool)
        R = h.shape[0]
        h, t2 = self.fc1.fwd(h.view(R, 1, 1, -1))
        h, t3 = self.fc2.fwd(h)
        o, t4 = self.box_out.fwd(h)
        from chainer_maskrcnn.nn import core
        plan = _plan_beside_forward(xs, rois_xy5, levels, self.roi_size_box, spatial_scales) if core.TRAIN else None
        self.box_tape = (t1, t2, t3, t4, tuple(pool.shape), rois_xy5, levels, spatial_scales, plan)
        self.last_box_out = o.view(R, self.out_p)
        return o.view(R, self.out_p)          # [:, :n_class] scores, [:, LOC0:LOC0+4] loc

    def mask_branch(self, xs, rois_xy5, levels, spatial_scales):
        pool = roi_align_fpn_fwd(xs, rois_xy5, levels, self.roi_size_mask, spatial_scales)
        h, tapes = pool, []
        for cv in self.mask_convs:
            h, t = cv.fwd(h)
            tapes.append(t)
        if self.merge_deconv:
            m, td, t2 = self._merged_deconv_fwd(h)
        else:
            d, td = self.deconv1.fwd(h)
            up = ops.pixel_shuffle2x(d, bias=self.ps.p(self.deconv_b))
            m, t2 = self.conv2.fwd(up)
        if self.upsample2x:                    # keypoint head: F.resize_images x2 (fpn_roi_keypoint_head.py:80-81)
            m = ops.bilinear2x_fwd(m)
        from chainer_maskrcnn.nn import core
        plan = _plan_beside_forward(xs, rois_xy5, levels, self.roi_size_mask, spatial_scales) if core.TRAIN else None
        self.mask_tape = (tapes, td, t2, rois_xy5, levels, spatial_scales, plan)
        return m                               # (Rm, mask_size, mask_size, pad32(mask_out_channels)) NHWC

    def compose_deconv(self, device):
        """Composed weights of deconv1 and conv2 for the current parameters.  The train chain calls this at the start of
        the step on its second stream (the parameters are final then), so the small composition kernel is off the
        critical path; ``mask_branch`` composes on demand otherwise."""
        dc, c2, C = self.deconv1, self.conv2, self.channels
        K2 = c2.cout_p
        wm = torch.empty((4 * K2, 1, 1, dc.cin_p), dtype=torch.float32, device=device)
        bm = torch.empty((K2,), dtype=torch.float32, device=device)
        check(lib().mrcnn_deconv_merge_fwd_f32(ptr(dc.W), ptr(self.ps.p(self.deconv_b)), ptr(c2.W), ptr(c2.b), ptr(wm), ptr(bm),
                                               C, dc.cin_p, K2, c2.cin_p, stream_ptr()))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self._composed = (wm, bm, ev)
        return wm, bm, ev

    def _merged_deconv_fwd(self, h):
        from chainer_maskrcnn._hip import nn as hnn
        dc, c2 = self.deconv1, self.conv2
        composed = getattr(self, '_composed', None)
        self._composed = None                      # valid for one forward pass: the parameters change every step
        wm, bm, ev = composed if composed is not None else self.compose_deconv(h.device)
        self._composed = None                      # (compose_deconv parks its result there for the train chain's early call)
        cur = torch.cuda.current_stream(h.device)
        cur.wait_event(ev)                         # composed on another stream at the start of the step
        wm.record_stream(cur)
        bm.record_stream(cur)
        hnn.LOGICAL = (dc.cin, 4 * c2.cout)
        try:
            d = hnn.conv2d_fwd_raw(h, wm, None, 1, 0, False)
        finally:
            hnn.LOGICAL = None
        m = ops.pixel_shuffle2x(d, bias=bm)
        return m, (h, wm), None

    def _merged_deconv_bwd(self, td, g_mask):
        """Gradients of deconv1 / conv2 parameters from the merged layer; returns the gradient of the deconv input."""
        from chainer_maskrcnn._hip import nn as hnn
        dc, c2, C = self.deconv1, self.conv2, self.channels
        x_in, wm = td
        K2 = c2.cout_p
        g4 = ops.pixel_shuffle2x(g_mask, inverse=True)
        hnn.LOGICAL = (dc.cin, 4 * c2.cout)
        try:
            G, gb4 = hnn.conv2d_bwd_filter_raw(x_in, g4, tuple(wm.shape), 1, 0, True)
            # x_in is the ReLU output of the last mask conv: its ReLU backward is fused here
            g = hnn.conv2d_bwd_data_raw(g4, wm, tuple(x_in.shape), 1, 0, relu_x=x_in if self.mask_convs else None)
        finally:
            hnn.LOGICAL = None
        ps = self.ps
        check(lib().mrcnn_deconv_merge_bwd_f32(ptr(G), ptr(gb4), ptr(dc.W), ptr(ps.p(self.deconv_b)), ptr(c2.W),
                                               ptr(ps.g(dc.name + '/W')), ptr(ps.g(self.deconv_b)), ptr(ps.g(c2.name + '/W')),
                                               ptr(ps.g(c2.name + '/b')) if c2.has_bias else None, C, dc.cin_p, K2, c2.cin_p,
                                               stream_ptr()))
        return g

    def __call__(self, x, indices_and_rois, levels, spatial_scales, train=True):
        """Reference signature (:55): x = pyramid levels, indices_and_rois (R,5) (idx,y1,x1,y2,x2),
        levels (R,).  Returns (roi_cls_locs (R,4), roi_scores (R,n_class), mask (R,n_class-1,28,28) NCHW view)."""
        xy5 = indices_and_rois[:, [0, 2, 1, 4, 3]].contiguous()
        lv = levels.to(torch.int32).contiguous()
        o = self.box_branch(x, xy5, lv, spatial_scales)
        locs, scores = o[:, self.LOC0:self.LOC0 + 4], o[:, :self.n_class]
        if not train:
            self.x = x
            return locs, scores
        m = self.mask_branch(x, xy5, lv, spatial_scales)
        return locs, scores, m[..., :self.mask_out_channels].permute(0, 3, 1, 2)

    def predict_mask(self, levels, indices_and_rois, spatial_scales):
        xy5 = indices_and_rois[:, [0, 2, 1, 4, 3]].contiguous()
        m = self.mask_branch(self.x, xy5, levels.to(torch.int32).contiguous(), spatial_scales)
        return m[..., :self.mask_out_channels].permute(0, 3, 1, 2)

    # ---- backward -------------------------------------------------------------------------------
    def backward(self, g_box_out, g_mask, g_feats):
        """g_box_out (R, out_p), g_mask (Rm,28,28,Cm) or None; g_feats: per-level gradient buffers, fully
        overwritten by the first ROIAlign backward and accumulated into by the second."""
        self.backward_box(g_box_out, g_feats)
        if g_mask is not None:
            self.backward_mask_pool(self.backward_mask_convs(g_mask), g_feats)

    def backward_box(self, g_box_out, g_feats, accumulate=False):
        t1, t2, t3, t4, pool_shape, rois, levels, scales, plan = self.box_tape
        R = g_box_out.shape[0]
        # every layer's input is the ReLU output of the layer below: the ReLU backward rides in the data-gradient epilogue
        g = self.box_out.bwd(t4, g_box_out.view(R, 1, 1, -1), mask_gx=True)
        g = self.fc2.bwd(t3, g, gy_masked=True, mask_gx=True)
        g = self.fc1.bwd(t2, g, gy_masked=True, mask_gx=True)
        g = self.conv1.bwd(t1, g.view(pool_shape), gy_masked=True)
        # accumulate: g_feats already hold a gradient (the RPN's, computed early) - the backward then only touches the patches RoIs land on
        roi_align_fpn_bwd(g, g_feats, rois, levels, self.roi_size_box, scales, accumulate=accumulate, plan=plan)
        self.box_tape = None

    def backward_mask_convs(self, g_mask):
        """Mask / keypoint branch backward down to the gradient of its pooled input."""
        from chainer_maskrcnn._hip import nn as hnn
        tapes, td, tc2, rois, levels, scales, _ = self.mask_tape
        if self.upsample2x:
            g_mask = ops.bilinear2x_bwd(g_mask)
        if self.merge_deconv:
            g = self._merged_deconv_bwd(td, g_mask)
            n = len(self.mask_convs)
            for i in range(n - 1, -1, -1):          # inputs of convs 1.. are ReLU outputs; conv 0 reads the pooled features
                g = self.mask_convs[i].bwd(tapes[i], g, gy_masked=True, mask_gx=(i > 0))
            return g
        g = self.conv2.bwd(tc2, g_mask)
        # deconv bias gradient = column sums of g over all output pixels: the filter-gradient call on the shuffled
        # tensor yields gb4 of length 4*C, summed over the 4 sub-pixel copies
        g4 = ops.pixel_shuffle2x(g, inverse=True)
        C = self.channels
        gb4 = torch.empty((4 * C,), dtype=torch.float32, device=g.device)
        gw = self.ps.g(self.deconv1.name + '/W')
        x_in = td[0]
        hnn.conv2d_bwd_filter_raw(x_in, g4, tuple(gw.shape), 1, 0, True, gw=gw, gb=gb4, accumulate=False)
        gb = self.ps.g(self.deconv_b)
        ops.add(gb4[0:C], gb4[C:2 * C], out=gb)
        ops.add(gb, gb4[2 * C:3 * C], out=gb)
        ops.add(gb, gb4[3 * C:4 * C], out=gb)
        g = hnn.conv2d_bwd_data_raw(g4, self.deconv1.W, tuple(x_in.shape), 1, 0)
        for cv, t in zip(reversed(self.mask_convs), reversed(tapes)):
            g = cv.bwd(t, g)
        return g

    def backward_mask_pool(self, g_pool, g_feats):
        tapes, td, tc2, rois, levels, scales, plan = self.mask_tape
        roi_align_fpn_bwd(g_pool, g_feats, rois, levels, self.roi_size_mask, scales, accumulate=True, plan=plan)
        self.mask_tape = None
