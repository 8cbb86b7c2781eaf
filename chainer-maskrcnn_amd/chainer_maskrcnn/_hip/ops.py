"""1:1 torch-tensor wrappers over the C ABI (include/mrcnn_hip.h) for everything except the
convolutions (``nn.py``) and ROIAlign (``functions/roi_align``).  Tensors are device-memory handles
only: every function checks the device, allocates outputs with ``torch.empty`` and launches HIP
kernels on the current stream.  No arithmetic happens in torch.
"""
import ctypes

import torch

from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr
from chainer_maskrcnn._hip.nn import workspace

f32 = torch.float32
i32 = torch.int32


def _empty(shape, dev, dtype=f32):
    return torch.empty(shape, dtype=dtype, device=dev)


def _ck(*ts):
    _hip.require_cuda(*ts)
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise ValueError('non-contiguous tensor passed to a HIP op')


# ---- batch norm ---------------------------------------------------------------------------------
def bn_train_fwd(x, gamma, beta, residual=None, relu=False, running_mean=None, running_var=None,
                 eps=2e-5, decay=0.9):
    """x (..., C) NHWC.  Returns (y, save_mean, save_invstd)."""
    _ck(x, gamma, beta, residual)
    C = x.shape[-1]
    P = x.numel() // C
    y = torch.empty_like(x)
    mean = _empty((C,), x.device)
    invstd = _empty((C,), x.device)
    nb = lib().mrcnn_bn_workspace_bytes(P, C)
    ws = workspace(nb, x.device)
    check(lib().mrcnn_bn_train_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(y), ptr(mean), ptr(invstd),
                                       ptr(running_mean), ptr(running_var), P, C, eps, decay, int(relu), ptr(ws),
                                       ws.numel(), stream_ptr()))
    return y, mean, invstd


def bn_train_fwd_stats(x, part, gamma, beta, residual=None, relu=False, running_mean=None, running_var=None, eps=2e-5, decay=0.9):
    """bn_train_fwd from the partial statistics (rows, 2, C) the producing convolution's epilogue left (hnn.conv2d_fwd_bnstats_raw)."""
    _ck(x, part, gamma, beta, residual)
    C = x.shape[-1]
    P = x.numel() // C
    y = torch.empty_like(x)
    mean = _empty((C,), x.device)
    invstd = _empty((C,), x.device)
    check(lib().mrcnn_bn_train_fwd_stats_f32(ptr(x), ptr(part), part.shape[0], ptr(gamma), ptr(beta), ptr(residual), ptr(y), ptr(mean),
                                             ptr(invstd), ptr(running_mean), ptr(running_var), P, C, eps, decay, int(relu), stream_ptr()))
    return y, mean, invstd


def bn_train_bwd(gy, x, y, gamma, mean, invstd, relu=False, want_gres=False, beta=None):
    """y None + beta given (BN + ReLU without residual): the ReLU mask is recomputed from x."""
    _ck(gy, x, y, gamma, mean, invstd, beta)
    C = x.shape[-1]
    P = x.numel() // C
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if want_gres else None
    gg = _empty((C,), x.device)
    gb = _empty((C,), x.device)
    ws = workspace(lib().mrcnn_bn_workspace_bytes(P, C), x.device)
    check(lib().mrcnn_bn_train_bwd_f32(ptr(gy), ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(gx), ptr(gres),
                                       ptr(gg), ptr(gb), P, C, int(relu), ptr(ws), ws.numel(), stream_ptr()))
    return gx, gres, gg, gb


def bn_infer_fwd(x, gamma, beta, mean, var, residual=None, relu=False, eps=2e-5):
    _ck(x, gamma, beta, mean, var, residual)
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(lib().mrcnn_bn_infer_fwd_f32(ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(var), ptr(residual), ptr(y),
                                       x.numel() // C, C, eps, int(relu), stream_ptr()))
    return y


def relu_bwd(gy, y, out=None):
    _ck(gy, y)
    out = torch.empty_like(gy) if out is None else out
    check(lib().mrcnn_relu_bwd_f32(ptr(gy), ptr(y), ptr(out), gy.numel(), stream_ptr()))
    return out


def add(a, b, out=None):
    _ck(a, b)
    out = torch.empty_like(a) if out is None else out
    check(lib().mrcnn_add_f32(ptr(a), ptr(b), ptr(out), a.numel(), stream_ptr()))
    return out


def maxpool2x2_fwd(x):
    _ck(x)
    N, H, W, C = x.shape
    y = _empty((N, (H + 1) // 2, (W + 1) // 2, C), x.device)
    check(lib().mrcnn_maxpool2x2_fwd_f32(ptr(x), ptr(y), N, H, W, C, stream_ptr()))
    return y


def maxpool3x3s2_fwd(x):
    """F.max_pooling_2d(ksize=3, stride=2) with cover_all (C4Backbone pool1)."""
    _ck(x)
    N, H, W, C = x.shape
    y = _empty((N, (H - 2) // 2 + 1, (W - 2) // 2 + 1, C), x.device)
    check(lib().mrcnn_maxpool3x3s2_fwd_f32(ptr(x), ptr(y), N, H, W, C, stream_ptr()))
    return y


def global_avg_pool(x):
    """x (R,H,W,C) -> (R,C): mean over the spatial positions."""
    _ck(x)
    R, H, W, C = x.shape
    y = _empty((R, C), x.device)
    check(lib().mrcnn_global_avg_pool_fwd_f32(ptr(x), ptr(y), R, H * W, C, stream_ptr()))
    return y


def relu(x, out=None):
    _ck(x)
    out = torch.empty_like(x) if out is None else out
    check(lib().mrcnn_relu_fwd_f32(ptr(x), ptr(out), x.numel(), stream_ptr()))
    return out


def maxpool2x2_bwd(x, gy):
    _ck(x, gy)
    N, H, W, C = x.shape
    gx = torch.empty_like(x)
    check(lib().mrcnn_maxpool2x2_bwd_f32(ptr(x), ptr(gy), ptr(gx), N, H, W, C, stream_ptr()))
    return gx


def upsample2x_add_fwd(top, lat):
    _ck(top, lat)
    N, H, W, C = lat.shape
    out = torch.empty_like(lat)
    check(lib().mrcnn_upsample2x_add_fwd_f32(ptr(top), ptr(lat), ptr(out), N, H, W, top.shape[1], top.shape[2], C,
                                             stream_ptr()))
    return out


def upsample2x_bwd(gout, gtop=None, top_shape=None):
    """gtop given => accumulate into it; else allocate (top_shape) and overwrite."""
    _ck(gout, gtop)
    N, H, W, C = gout.shape
    acc = gtop is not None
    if gtop is None:
        gtop = _empty(top_shape, gout.device)
    check(lib().mrcnn_upsample2x_bwd_f32(ptr(gout), ptr(gtop), N, H, W, gtop.shape[1], gtop.shape[2], C, int(acc),
                                         stream_ptr()))
    return gtop


def subsample_bwd(gsub, x_shape, stride, gx=None, relu_x=None):
    """relu_x (x_shape, nullable): the scattered positions are zeroed where relu_x <= 0 (after the accumulation); positions off
    the lattice keep what gx held - pass a gx that is already masked."""
    _ck(gsub, gx, relu_x)
    N, H, W, C = x_shape
    acc = gx is not None
    if gx is None:
        gx = _empty(x_shape, gsub.device)
    check(lib().mrcnn_subsample_bwd_f32(ptr(gsub), ptr(gx), N, H, W, C, stride, int(acc), ptr(relu_x), stream_ptr()))
    return gx


def pixel_shuffle2x(t, bias=None, inverse=False):
    """forward: t (N,H,W,4*C) [+ bias (C)] -> (N,2H,2W,C); inverse: (N,2H,2W,C) -> (N,H,W,4*C)."""
    _ck(t, bias)
    if not inverse:
        N, H, W, C4 = t.shape
        C = C4 // 4
        out = _empty((N, 2 * H, 2 * W, C), t.device)
    else:
        N, H2, W2, C = t.shape
        H, W = H2 // 2, W2 // 2
        out = _empty((N, H, W, 4 * C), t.device)
    check(lib().mrcnn_pixel_shuffle2x_f32(ptr(t), ptr(bias), ptr(out), N, H, W, C, int(inverse), stream_ptr()))
    return out


def bilinear2x_fwd(x):
    _ck(x)
    N, H, W, C = x.shape
    y = _empty((N, 2 * H, 2 * W, C), x.device)
    check(lib().mrcnn_bilinear2x_fwd_f32(ptr(x), ptr(y), N, H, W, C, stream_ptr()))
    return y


def bilinear2x_bwd(gy):
    _ck(gy)
    N, OH, OW, C = gy.shape
    gx = _empty((N, OH // 2, OW // 2, C), gy.device)
    check(lib().mrcnn_bilinear2x_bwd_f32(ptr(gy), ptr(gx), N, OH // 2, OW // 2, C, stream_ptr()))
    return gx


def image_nchw3_to_nhwc4(x):
    _ck(x)
    N, C, H, W = x.shape
    if C != 3:
        raise ValueError('expected (N,3,H,W) images')
    y = _empty((N, H, W, 4), x.device)
    check(lib().mrcnn_image_nchw3_to_nhwc4_f32(ptr(x), ptr(y), N, H, W, stream_ptr()))
    return y


def image_resize_f32(img, oh, ow, div=1.0):
    """img (C,H,W) float32 -> (C,oh,ow): cv2.resize INTER_LINEAR float rule, then / div (MaskRCNN.prepare)."""
    _ck(img)
    C, H, W = img.shape
    out = _empty((C, oh, ow), img.device)
    check(lib().mrcnn_image_resize_f32(ptr(img), C, H, W, ptr(out), oh, ow, oh, ow, float(div), stream_ptr()))
    return out


def random_keys(shape, seed, device):
    """uint32 sampler keys stored in an int32 tensor."""
    out = _empty(shape, device, i32)
    check(lib().mrcnn_random_keys_u32(ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), stream_ptr()))
    return out


def seed_state(seed, device):
    """Device-resident sampler seed (int64 tensor of one element)."""
    return torch.tensor([int(seed) & (2 ** 63 - 1)], dtype=torch.int64, device=device)


def random_keys_dev(shape, state):
    """uint32 keys from the device seed state; advances the state (graph-replay safe)."""
    out = _empty(shape, state.device, i32)
    check(lib().mrcnn_random_keys_dev_u32(ptr(out), out.numel(), ptr(state), stream_ptr()))
    return out


def sgd_momentum_wd(p, g, v, lr, momentum=0.9, weight_decay=5e-4):
    _ck(p, g, v)
    check(lib().mrcnn_sgd_momentum_wd_f32(ptr(p), ptr(g), ptr(v), p.numel(), lr, momentum, weight_decay, stream_ptr()))


# ---- losses -------------------------------------------------------------------------------------
def _loss_ws(dev):
    return workspace(lib().mrcnn_loss_workspace_bytes(), dev)


def softmax_ce_fills_gradient(M, K, xmap, gmap=None, Kfill=0):
    """True when mrcnn_softmax_ce_f32 takes its channel-interleaved path for these element maps (rows = (group, channel) over an NHWC
    (G, K, C) tensor: the keypoint loss) - the callee then writes EVERY element of gx (zeros in the padded channels and the ignored
    rows), so the caller need not zero-fill it.  Asks the library's own dispatch predicate (mrcnn_softmax_ce_fills_gx)."""
    A, gs, rs, es = xmap
    _, ggs, grs, ges = gmap or xmap
    return bool(lib().mrcnn_softmax_ce_fills_gx(int(M), int(K), int(A), int(gs), int(rs), int(es), int(ggs), int(grs), int(ges), int(Kfill)))


def softmax_ce(x, t, M, K, xmap, gmap=None, Kfill=0, want_grad=True, gx=None, ignore_label=-1, out=None):
    """x: device tensor holding a logical (M,K) matrix; xmap = (A, gs, rs, es) element map
    (see include/mrcnn_hip.h).  Returns (loss_out (2,), gx)."""
    _ck(t)
    _hip.require_cuda(x)
    out = _empty((2,), x.device) if out is None else out
    gmap = gmap or xmap
    if want_grad and gx is None:
        gx = torch.empty_like(x)
    ws = _loss_ws(x.device)
    check(lib().mrcnn_softmax_ce_f32(ptr(x), xmap[0], xmap[1], xmap[2], xmap[3], ptr(t), M, K, ignore_label, ptr(out),
                                     ptr(gx) if want_grad else None, gmap[1], gmap[2], gmap[3], Kfill, ptr(ws),
                                     ws.numel(), stream_ptr()))
    return out, gx


def smooth_l1(x, ldx, t, label, M, sigma, want_grad=True, gfill=0, col0=0, gx=None, out=None):
    """x: (M, ldx) buffer; the 4 predictions of row r start at column col0.  gx (same buffer shape) receives
    columns [col0, col0+max(4,gfill))."""
    _ck(x, t, label, gx)
    out = _empty((2,), x.device) if out is None else out
    if want_grad and gx is None:
        gx = torch.empty_like(x)
    ws = _loss_ws(x.device)
    xp = ctypes.c_void_p(x.data_ptr() + 4 * col0)
    gp = ctypes.c_void_p(gx.data_ptr() + 4 * col0) if want_grad else None
    check(lib().mrcnn_smooth_l1_f32(xp, ldx, ptr(t), ptr(label), M, sigma, ptr(out), gp, ldx, gfill, ptr(ws),
                                    ws.numel(), stream_ptr()))
    return out, gx


def mask_bce(x, gt, label, want_grad=True, out=None):
    """x (Rm,H,W,Cm) NHWC logits, gt (Rm,H,W) int32, label (Rm,) int32."""
    _ck(x, gt, label)
    Rm, H, W, Cm = x.shape
    out = _empty((2,), x.device) if out is None else out
    gx = torch.empty_like(x) if want_grad else None
    ws = _loss_ws(x.device)
    check(lib().mrcnn_mask_bce_f32(ptr(x), ptr(gt), ptr(label), Rm, H * W, Cm, ptr(out), ptr(gx), ptr(ws), ws.numel(),
                                   stream_ptr()))
    return out, gx


def loss_total(losses):
    """losses (n,2) device pairs -> (1,) total."""
    _ck(losses)
    out = _empty((1,), losses.device)
    check(lib().mrcnn_loss_total_f32(ptr(losses), losses.shape[0], ptr(out), stream_ptr()))
    return out


# ---- RPN proposal path --------------------------------------------------------------------------
def rpn_pack(head, A, locs, scores, a_off):
    _ck(head, locs, scores)
    N, H, W, Cp = head.shape
    check(lib().mrcnn_rpn_pack_f32(ptr(head), N, H * W, Cp, A, ptr(locs), ptr(scores), a_off, locs.shape[1],
                                   stream_ptr()))


def rpn_pack_levels(heads, A, locs, scores):
    """All levels' head outputs (NHWC, same N and padded channel count) into locs / scores in ONE launch; level order = anchor order."""
    import ctypes
    _ck(locs, scores, *heads)
    L = len(heads)
    N, _, _, Cp = heads[0].shape
    ptrs = (ctypes.c_void_p * L)(*[h.data_ptr() for h in heads])
    hws = (ctypes.c_int * L)(*[h.shape[1] * h.shape[2] for h in heads])
    check(lib().mrcnn_rpn_pack_levels_f32(ptrs, hws, L, N, Cp, A, ptr(locs), ptr(scores), locs.shape[1], stream_ptr()))


def rpn_unpack_grad_levels(glocs, gscores, head_shapes, A):
    """The backward of rpn_pack_levels: one gradient tensor per level, one launch."""
    import ctypes
    _ck(glocs, gscores)
    L = len(head_shapes)
    N, _, _, Cp = head_shapes[0]
    gheads = [_empty(sh, glocs.device) for sh in head_shapes]
    ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in gheads])
    hws = (ctypes.c_int * L)(*[sh[1] * sh[2] for sh in head_shapes])
    check(lib().mrcnn_rpn_unpack_grad_levels_f32(ptr(glocs), ptr(gscores), ptrs, hws, L, N, Cp, A, glocs.shape[1], stream_ptr()))
    return gheads


def rpn_unpack_grad(glocs, gscores, head_shape, A, a_off):
    _ck(glocs, gscores)
    N, H, W, Cp = head_shape
    ghead = _empty(head_shape, glocs.device)
    check(lib().mrcnn_rpn_unpack_grad_f32(ptr(glocs), ptr(gscores), N, H * W, Cp, A, ptr(ghead), a_off, glocs.shape[1],
                                          stream_ptr()))
    return ghead


def rpn_proposals(locs, scores, anchors, img_size, min_size, n_pre, n_post, nms_thresh, debug=False, per_image=None):
    """locs (N,A,4), scores (N,A,2), anchors (A,4).  Returns dict of padded device outputs.
    per_image: optional (N,3) f32 device tensor (h, w, min_size * scale) - every image clipped / filtered with its own."""
    _ck(locs, scores, anchors, per_image)
    N, A, _ = locs.shape
    dev = locs.device
    rois = _empty((N * n_post, 4), dev)
    idx = _empty((N * n_post,), dev, i32)
    lev = _empty((N * n_post,), dev)
    cnt = _empty((N,), dev, i32)
    npre = min(n_pre, A)
    dbg = [None, None, None]
    if debug:
        dbg = [torch.full((N * npre,), -1, dtype=i32, device=dev), _empty((N * n_post,), dev, i32), _empty((N,), dev, i32)]
    ws = workspace(lib().mrcnn_rpn_proposals_workspace_bytes(N, A, n_pre, n_post), dev)
    check(lib().mrcnn_rpn_proposals_f32(ptr(locs), ptr(scores), ptr(anchors), N, A, float(img_size[0]),
                                        float(img_size[1]), float(min_size), ptr(per_image), n_pre, n_post, float(nms_thresh), ptr(rois),
                                        ptr(idx), ptr(lev), ptr(cnt), ptr(dbg[0]), ptr(dbg[1]), ptr(dbg[2]), ptr(ws),
                                        ws.numel(), stream_ptr()))
    return dict(rois=rois, roi_indices=idx, levels=lev, n_rois=cnt, sorted_anchor=dbg[0], keep=dbg[1], n_pre=dbg[2])


def softmax2(scores):
    """(..., 2) class scores -> softmax probabilities (the fg score of ChainerCV's single-level RPN)."""
    _ck(scores)
    out = torch.empty_like(scores)
    check(lib().mrcnn_softmax2_f32(ptr(scores), ptr(out), scores.numel() // 2, stream_ptr()))
    return out


def nms(boxes, thresh, max_keep=None):
    """Greedy NMS in the given order; returns (keep (max_keep,) int32 padded, n_keep (1,) int32)."""
    _ck(boxes)
    n = boxes.shape[0]
    max_keep = max_keep or max(n, 1)
    keep = torch.full((max_keep,), -1, dtype=i32, device=boxes.device)
    nk = _empty((1,), boxes.device, i32)
    ws = workspace(lib().mrcnn_nms_workspace_bytes(n), boxes.device)
    check(lib().mrcnn_nms_f32(ptr(boxes), n, float(thresh), max_keep, ptr(keep), ptr(nk), ptr(ws), ws.numel(),
                              stream_ptr()))
    return keep, nk


def map_rois_to_fpn_levels(rois, k_min=0, k_max=4):
    _ck(rois)
    lev = _empty((rois.shape[0],), rois.device)
    check(lib().mrcnn_map_rois_to_fpn_levels_f32(ptr(rois), rois.shape[0], k_min, k_max, ptr(lev), stream_ptr()))
    return lev


# ---- targets ------------------------------------------------------------------------------------
def proposal_target(rois, roi_levels, n_rois, gt_boxes, gt_labels, n_gt, keys, n_sample=256, pos_ratio=0.25,
                    pos_iou_thresh=0.5, neg_hi=0.5, neg_lo=0.0, mean=(0., 0., 0., 0.), std=(0.1, 0.1, 0.2, 0.2),
                    pos_order=None, neg_order=None):
    """rois (N*roi_cap,4) padded, gt_boxes (N,gt_cap,4), gt_labels (N,gt_cap) i32, keys (N,roi_cap+gt_cap) u32 (as int32
    storage).  Returns dict of per-row outputs (N*n_sample rows) + 'n_cand' (N,2) candidate-set sizes.
    pos_order / neg_order (N,n_sample) int32: reference-order mode (see include/mrcnn_hip.h), keys may be None."""
    _ck(rois, roi_levels, n_rois, gt_boxes, gt_labels, n_gt, keys, pos_order, neg_order)
    N, gt_cap = gt_labels.shape
    roi_cap = rois.shape[0] // N
    dev = rois.device
    R = N * n_sample
    o = dict(sample_roi=_empty((R, 4), dev), rois_xy5=_empty((R, 5), dev), sample_levels=_empty((R,), dev, i32),
             gt_roi_loc=_empty((R, 4), dev), gt_roi_label=_empty((R,), dev, i32), gt_assign=_empty((R,), dev, i32),
             sample_src=_empty((R,), dev, i32), n_pos=_empty((N,), dev, i32), n_sampled=_empty((N,), dev, i32),
             n_cand=_empty((N, 2), dev, i32))
    m4 = (ctypes.c_float * 4)(*mean)
    s4 = (ctypes.c_float * 4)(*std)
    import numpy as np
    n_pos_max = int(np.round(n_sample * pos_ratio))
    check(lib().mrcnn_proposal_target_f32(ptr(rois), ptr(roi_levels), ptr(n_rois), roi_cap, ptr(gt_boxes), ptr(gt_labels),
                                          ptr(n_gt), gt_cap, ptr(keys), N, n_sample, n_pos_max, pos_iou_thresh, neg_hi,
                                          neg_lo, ctypes.cast(m4, ctypes.c_void_p), ctypes.cast(s4, ctypes.c_void_p),
                                          ptr(o['sample_roi']), ptr(o['rois_xy5']), ptr(o['sample_levels']),
                                          ptr(o['gt_roi_loc']), ptr(o['gt_roi_label']), ptr(o['gt_assign']),
                                          ptr(o['sample_src']), ptr(o['n_pos']), ptr(o['n_sampled']), ptr(pos_order),
                                          ptr(neg_order), ptr(o['n_cand']), stream_ptr()))
    return o


def mask_target(masks, sample_roi, gt_assign, n_pos, n_sample, pos_cap, mask_size):
    """masks (N,gt_cap,H,W) uint8 -> (N*pos_cap, mask_size, mask_size) int32 (-1 rows for unused slots)."""
    _ck(masks, sample_roi, gt_assign, n_pos)
    N, gt_cap, H, W = masks.shape
    out = _empty((N * pos_cap, mask_size, mask_size), masks.device, i32)
    check(lib().mrcnn_mask_target_u8(ptr(masks), N, gt_cap, H, W, ptr(sample_roi), ptr(gt_assign), ptr(n_pos), n_sample,
                                     pos_cap, mask_size, ptr(out), stream_ptr()))
    return out


def keypoint_target(kps, sample_roi, gt_assign, n_pos, n_sample, pos_cap, mask_size, inplace_quirk=False):
    """kps (N,gt_cap,K,3) f32 -> (N*pos_cap, K) int32.  inplace_quirk: the reference's in-place gt mutation (App. B-11)."""
    _ck(kps, sample_roi, gt_assign, n_pos)
    N, gt_cap, K, _ = kps.shape
    out = _empty((N * pos_cap, K), kps.device, i32)
    check(lib().mrcnn_keypoint_target_f32(ptr(kps), N, gt_cap, K, ptr(sample_roi), ptr(gt_assign), ptr(n_pos), n_sample,
                                          pos_cap, mask_size, int(inplace_quirk), ptr(out), stream_ptr()))
    return out


def count_valid_labels(labels):
    """labels (N,G) int32 (-1 = padding row) -> per-image gt counts (N,) int32, on the device."""
    _ck(labels)
    N, G = labels.shape
    out = _empty((N,), labels.device, i32)
    check(lib().mrcnn_count_valid_labels_i32(ptr(labels), N, G, ptr(out), stream_ptr()))
    return out


def anchor_target(anchors, gt_boxes, n_gt, img_size, keys=None, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3,
                  pos_ratio=0.5, per_image_hw=None):
    """anchors (A,4), gt_boxes (N,gt_cap,4), keys (N,A) u32 or None (= no subsampling).  Returns (loc (N,A,4), label (N,A)).
    per_image_hw: optional (N,2) f32 device tensor - each image's own size for the inside test."""
    _ck(anchors, gt_boxes, n_gt, keys, per_image_hw)
    A = anchors.shape[0]
    N, gt_cap, _ = gt_boxes.shape
    dev = anchors.device
    loc = _empty((N, A, 4), dev)
    label = _empty((N, A), dev, i32)
    ws = workspace(lib().mrcnn_anchor_target_workspace_bytes(N, A), dev)
    check(lib().mrcnn_anchor_target_f32(ptr(anchors), A, ptr(gt_boxes), ptr(n_gt), gt_cap, N, float(img_size[0]),
                                        float(img_size[1]), ptr(per_image_hw), ptr(keys), n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio,
                                        int(keys is not None), ptr(loc), ptr(label), ptr(ws), ws.numel(), stream_ptr()))
    return loc, label


# ---- inference post-processing --------------------------------------------------------------------
def detect_decode(rois, box_out, n_class, loc0, scale, mean, std, size):
    """rois (R,4), box_out (R,ld) -> cls_bbox (R,4) in original-image pixels, prob (R,n_class)."""
    _ck(rois, box_out)
    R, ld = box_out.shape
    cls_bbox = _empty((R, 4), rois.device)
    prob = _empty((R, n_class), rois.device)
    m4 = (ctypes.c_float * 4)(*mean)
    s4 = (ctypes.c_float * 4)(*std)
    check(lib().mrcnn_detect_decode_f32(ptr(rois), R, ptr(box_out), ld, n_class, loc0, float(scale),
                                        ctypes.cast(m4, ctypes.c_void_p), ctypes.cast(s4, ctypes.c_void_p), float(size[0]),
                                        float(size[1]), ptr(cls_bbox), ptr(prob), stream_ptr()))
    return cls_bbox, prob


def class_nms(cls_bbox, prob, l_begin, l_end, score_thresh, nms_thresh):
    """-> keep_idx (n_class,R) int32, keep_cnt (n_class,) int32."""
    _ck(cls_bbox, prob)
    R, n_class = prob.shape
    keep_idx = torch.full((n_class, max(R, 1)), -1, dtype=i32, device=prob.device)
    keep_cnt = torch.zeros((n_class,), dtype=i32, device=prob.device)
    if R > 0:
        check(lib().mrcnn_class_nms_f32(ptr(cls_bbox), ptr(prob), R, n_class, l_begin, l_end, float(score_thresh),
                                        float(nms_thresh), ptr(keep_idx), ptr(keep_cnt), stream_ptr()))
    return keep_idx, keep_cnt


def mask_paste(mask_logits, label, bbox, size):
    """mask_logits (D,S,S,Cm) NHWC, label (D,) int32, bbox (D,4) -> (D,H,W) uint8."""
    _ck(mask_logits, label, bbox)
    D, S, _, Cm = mask_logits.shape
    out = torch.empty((D, size[0], size[1]), dtype=torch.uint8, device=mask_logits.device)
    check(lib().mrcnn_mask_paste_f32(ptr(mask_logits), D, S, Cm, ptr(label), ptr(bbox), size[0], size[1], ptr(out),
                                     stream_ptr()))
    return out
