"""Oracle: box codecs, IoU, anchors, FPN level map, NMS (TEST INFRASTRUCTURE).

Third-party routines restated here are NOT on disk under /root/reference (ChainerCV is
an uninstalled pip dependency with no pinned version: README.md:43-47) => restated from
the published ChainerCV 0.9-0.10 sources, SURVEY.md Appendix A-2..A-4, "parity unpinned".
Reference call sites are cited per function.  All arithmetic is float32.
"""
import numpy as np

F = np.float32


def loc2bbox(src_bbox, loc):
    """ChainerCV ``loc2bbox``; call sites: utils/proposal_creator.py:131 (mirror of the
    ChainerCV ProposalCreator used at model/rpn/multilevel_region_proposal_network.py:157),
    model/maskrcnn.py:196."""
    src_bbox = np.asarray(src_bbox, F)
    loc = np.asarray(loc, F)
    if src_bbox.shape[0] == 0:
        return np.zeros((0, 4), dtype=F)
    h = src_bbox[:, 2] - src_bbox[:, 0]
    w = src_bbox[:, 3] - src_bbox[:, 1]
    cy = src_bbox[:, 0] + F(0.5) * h
    cx = src_bbox[:, 1] + F(0.5) * w
    ncy = loc[:, 0] * h + cy
    ncx = loc[:, 1] * w + cx
    nh = np.exp(loc[:, 2]) * h
    nw = np.exp(loc[:, 3]) * w
    out = np.empty(loc.shape, F)
    out[:, 0] = ncy - F(0.5) * nh
    out[:, 1] = ncx - F(0.5) * nw
    out[:, 2] = ncy + F(0.5) * nh
    out[:, 3] = ncx + F(0.5) * nw
    return out


def bbox2loc(src_bbox, dst_bbox):
    """ChainerCV ``bbox2loc``; call site utils/proposal_target_creator.py:88 and the
    ChainerCV AnchorTargetCreator used at model/fpn_maskrcnn_train_chain.py:81-82."""
    src_bbox = np.asarray(src_bbox, F)
    dst_bbox = np.asarray(dst_bbox, F)
    h = src_bbox[:, 2] - src_bbox[:, 0]
    w = src_bbox[:, 3] - src_bbox[:, 1]
    cy = src_bbox[:, 0] + F(0.5) * h
    cx = src_bbox[:, 1] + F(0.5) * w
    bh = dst_bbox[:, 2] - dst_bbox[:, 0]
    bw = dst_bbox[:, 3] - dst_bbox[:, 1]
    bcy = dst_bbox[:, 0] + F(0.5) * bh
    bcx = dst_bbox[:, 1] + F(0.5) * bw
    eps = np.finfo(F).eps
    h = np.maximum(h, eps)
    w = np.maximum(w, eps)
    dy = (bcy - cy) / h
    dx = (bcx - cx) / w
    dh = np.log(bh / h)
    dw = np.log(bw / w)
    return np.stack((dy, dx, dh, dw), axis=1).astype(F)


def bbox_iou(a, b):
    """ChainerCV ``bbox_iou``; call site utils/proposal_target_creator.py:55."""
    a = np.asarray(a, F)
    b = np.asarray(b, F)
    tl = np.maximum(a[:, None, :2], b[None, :, :2])
    br = np.minimum(a[:, None, 2:], b[None, :, 2:])
    area_i = np.prod(br - tl, axis=2) * (tl < br).all(axis=2)
    area_a = np.prod(a[:, 2:] - a[:, :2], axis=1)
    area_b = np.prod(b[:, 2:] - b[:, :2], axis=1)
    return (area_i / (area_a[:, None] + area_b - area_i)).astype(F)


def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32)):
    """ChainerCV ``generate_anchor_base``; call site
    model/rpn/multilevel_region_proposal_network.py:70-71 (one scale per level)."""
    py = base_size / 2.
    px = base_size / 2.
    ab = np.zeros((len(ratios) * len(anchor_scales), 4), dtype=F)
    for i in range(len(ratios)):
        for j in range(len(anchor_scales)):
            h = base_size * anchor_scales[j] * np.sqrt(ratios[i])
            w = base_size * anchor_scales[j] * np.sqrt(1. / ratios[i])
            k = i * len(anchor_scales) + j
            ab[k, 0] = py - h / 2.
            ab[k, 1] = px - w / 2.
            ab[k, 2] = py + h / 2.
            ab[k, 3] = px + w / 2.
    return ab


def enumerate_shifted_anchor(anchor_base, feat_stride, height, width):
    """ChainerCV ``_enumerate_shifted_anchor``; call site
    model/rpn/multilevel_region_proposal_network.py:128-129."""
    shift_y = np.arange(0, height * feat_stride, feat_stride)
    shift_x = np.arange(0, width * feat_stride, feat_stride)
    shift_x, shift_y = np.meshgrid(shift_x, shift_y)
    shift = np.stack((shift_y.ravel(), shift_x.ravel(),
                      shift_y.ravel(), shift_x.ravel()), axis=1)
    A = anchor_base.shape[0]
    K = shift.shape[0]
    anchor = anchor_base.reshape((1, A, 4)) + shift.reshape((1, K, 4)).transpose((1, 0, 2))
    return anchor.reshape((K * A, 4)).astype(F)


def fpn_anchors(feat_shapes, feat_strides=(4, 8, 16, 32, 64),
                anchor_sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1, 2)):
    """Concatenated anchors of all levels, model/rpn/multilevel_region_proposal_network.py:126-152
    with anchor_scales = anchor_sizes/16 (model/extractor/feature_pyramid_network.py:43-44)."""
    out = []
    for (hh, ww), st, sz in zip(feat_shapes, feat_strides, anchor_sizes):
        base = generate_anchor_base(anchor_scales=[sz / 16.], ratios=ratios)
        out.append(enumerate_shifted_anchor(base, st, hh, ww))
    return np.concatenate(out, axis=0)


def map_rois_to_fpn_levels(rois, k_min=0, k_max=4):
    """model/rpn/multilevel_region_proposal_network.py:16-31 (in-tree, PINNED by
    tests/golden/levels_reference.npz which was produced by the reference function).
    rois (R,4) yx; returns float32 levels."""
    rois = np.asarray(rois, F)
    area = np.prod(rois[:, 2:] - rois[:, :2], axis=1)
    s = np.sqrt(area)
    s0 = 224
    lvl0 = 4
    target = np.floor(lvl0 + np.log2(s / s0 + 1e-6))
    return np.clip(target, k_min, k_max)


def nms(bbox, thresh):
    """Greedy NMS in the given order, ChainerCV ``non_maximum_suppression`` (GPU kernel
    formula ``devIoU``; the CPU path uses the identical float32 expression), called by the
    ProposalCreator at model/rpn/multilevel_region_proposal_network.py:157-158 and by
    model/maskrcnn.py:300.  Suppress i iff some kept j has
    ``area_i/(area_a+area_b-area_i) >= thresh``.  Returns int32 indices."""
    bbox = np.asarray(bbox, F)
    n = bbox.shape[0]
    if n == 0:
        return np.zeros((0,), np.int32)
    area = ((bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1])).astype(F)
    keep = []
    kb = np.zeros((0, 4), F)
    ka = np.zeros((0,), F)
    cap = 0
    nk = 0
    for i in range(n):
        if nk:
            top = np.maximum(bbox[i, 0], kb[:nk, 0])
            left = np.maximum(bbox[i, 1], kb[:nk, 1])
            bottom = np.minimum(bbox[i, 2], kb[:nk, 2])
            right = np.minimum(bbox[i, 3], kb[:nk, 3])
            hgt = np.maximum(bottom - top, F(0))
            wid = np.maximum(right - left, F(0))
            ai = hgt * wid
            with np.errstate(invalid='ignore', divide='ignore'):
                iou = ai / ((area[i] + ka[:nk]) - ai)
            if (iou >= F(thresh)).any():
                continue
        if nk == cap:
            cap = max(64, cap * 2)
            kb = np.concatenate([kb, np.zeros((cap - kb.shape[0], 4), F)])
            ka = np.concatenate([ka, np.zeros((cap - ka.shape[0],), F)])
        kb[nk] = bbox[i]
        ka[nk] = area[i]
        nk += 1
        keep.append(i)
    return np.asarray(keep, np.int32)


def argsort_desc_pinned(score):
    """``score.argsort()[::-1]`` (utils/proposal_creator.py:148) with the tie order PINNED
    to (score desc, index desc) = a stable ascending sort reversed (NumPy's default
    introsort leaves ties unspecified)."""
    return np.argsort(np.asarray(score), kind='stable')[::-1]
