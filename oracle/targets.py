"""Oracle: training-target creators (TEST INFRASTRUCTURE - see oracle/__init__.py).

  ProposalTargetCreator  <- utils/proposal_target_creator.py:12-137 (in-tree; its CONTROL
                            FLOW is pinned by tests/golden/ptc_reference.npz, produced by
                            running the reference class with this oracle's bbox_iou /
                            bbox2loc / cv2_resize_linear_u8 injected for the absent
                            ChainerCV / OpenCV modules)
  AnchorTargetCreator    <- ChainerCV (third-party, absent), default
                            instance used at model/fpn_maskrcnn_train_chain.py:21,81-82
  cv2_resize_linear_u8   <- OpenCV ``cv2.resize`` default interpolation on uint8
                            (utils/proposal_target_creator.py:102-103); restated from the
                            published resize.cpp fixed-point path ("parity unpinned":
                            cv2 is not installed here)

Random draws: the reference uses the *global* ``np.random`` (``np.random.choice(...,
replace=False)``).  Here an explicit ``RandomState`` is injected; seeding it like the
global state reproduces the same draws.
"""
import numpy as np

from .boxes import bbox_iou, bbox2loc, map_rois_to_fpn_levels

F = np.float32


def cv2_resize_linear_u8(src, dsize):
    """``cv2.resize(src, (dw, dh))`` for a 2-D uint8 image, INTER_LINEAR fixed-point path
    (INTER_RESIZE_COEF_BITS = 11): half-pixel centres, border clamp, horizontal pass in
    int32 then ``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`` vertically."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dw, dh = dsize
    if sh == 0 or sw == 0:
        raise ValueError('cv2.resize: empty source (ssize.empty())')
    if (dh, dw) == (sh, sw):
        return src.copy()
    ONE = 2048

    def coeffs(ssize, dsize_):
        scale = 1.0 / (float(dsize_) / float(ssize))   # resize.cpp: scale_x = 1./inv_scale_x
        ofs = np.zeros(dsize_, np.int64)
        a = np.zeros((dsize_, 2), np.int64)
        for d in range(dsize_):
            f = F((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            f = F(f - F(s))
            if s < 0:
                f = F(0); s = 0
            if s >= ssize - 1:
                f = F(0); s = ssize - 1
            ofs[d] = s
            # saturate_cast<short>(float * 2048) == cvRound (round-half-even)
            a[d, 0] = int(np.rint(F(F(1) - f) * F(ONE)))
            a[d, 1] = int(np.rint(f * F(ONE)))
        return ofs, a

    xo, xa = coeffs(sw, dw)
    yo, ya = coeffs(sh, dh)
    s32 = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    rows = s32[:, xo] * xa[:, 0][None, :] + s32[:, x1] * xa[:, 1][None, :]   # (sh, dw)
    y1 = np.minimum(yo + 1, sh - 1)
    S0 = rows[yo]
    S1 = rows[y1]
    b0 = ya[:, 0][:, None]
    b1 = ya[:, 1][:, None]
    out = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


class ProposalTargetCreator(object):
    """utils/proposal_target_creator.py:12-137."""

    def __init__(self, sizes=(16,), n_sample=256, pos_ratio=0.25, pos_iou_thresh=0.5,
                 neg_iou_thresh_hi=0.5, neg_iou_thresh_lo=0.0):
        self.sizes = sizes                      # stored, never used (:19)
        self.n_sample = n_sample
        self.pos_ratio = pos_ratio
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh_hi = neg_iou_thresh_hi
        self.neg_iou_thresh_lo = neg_iou_thresh_lo

    def __call__(self, roi, bbox, label, mask, levels,
                 loc_normalize_mean=(0., 0., 0., 0.), loc_normalize_std=(0.1, 0.1, 0.2, 0.2),
                 mask_size=14, binary_mask=True, rng=None, return_debug=False, mutate_gt=True):
        rng = rng if rng is not None else np.random
        roi = np.asarray(roi, F)
        bbox = np.asarray(bbox, F)
        roi = np.concatenate((roi, bbox), axis=0)                           # :48
        bbox_levels = map_rois_to_fpn_levels(bbox)                           # :51
        levels = np.concatenate([np.asarray(levels), bbox_levels])           # :52
        pos_roi_per_image = np.round(self.n_sample * self.pos_ratio)         # :54
        iou = bbox_iou(roi, bbox)                                            # :55
        gt_assignment = iou.argmax(axis=1)                                   # :56
        max_iou = iou.max(axis=1)                                            # :57
        gt_roi_label = np.asarray(label)[gt_assignment] + 1                  # :60
        pos_index = np.where(max_iou >= self.pos_iou_thresh)[0]              # :63
        pos_roi_per_this_image = int(min(pos_roi_per_image, pos_index.size))
        if pos_index.size > 0:
            pos_index = rng.choice(pos_index, size=pos_roi_per_this_image, replace=False)  # :66
        neg_index = np.where((max_iou < self.neg_iou_thresh_hi) &
                             (max_iou >= self.neg_iou_thresh_lo))[0]         # :71-72
        neg_roi_per_this_image = self.n_sample - pos_roi_per_this_image
        neg_roi_per_this_image = int(min(neg_roi_per_this_image, neg_index.size))
        if neg_index.size > 0:
            neg_index = rng.choice(neg_index, size=neg_roi_per_this_image, replace=False)  # :77
        keep_index = np.append(pos_index, neg_index)                         # :81
        gt_roi_label = gt_roi_label[keep_index]
        gt_roi_label[pos_roi_per_this_image:] = 0                            # :83
        sample_roi = roi[keep_index]
        sample_levels = levels[keep_index]
        gt_roi_loc = bbox2loc(sample_roi, bbox[gt_assignment[keep_index]])   # :88
        gt_roi_loc = ((gt_roi_loc - np.array(loc_normalize_mean, F)) /
                      np.array(loc_normalize_std, F))                        # :89-90
        gt_roi_mask = []
        if binary_mask:
            _, h, w = mask.shape
            for i, idx in enumerate(gt_assignment[pos_index]):               # :96-103
                A = mask[idx,
                         max(int(sample_roi[i, 0]), 0):min(int(sample_roi[i, 2]), h),
                         max(int(sample_roi[i, 1]), 0):min(int(sample_roi[i, 3]), w)]
                gt_roi_mask.append(cv2_resize_linear_u8(A, (mask_size, mask_size)).astype(np.int32))
        else:
            mask = np.asarray(mask)
            for i, idx in enumerate(gt_assignment[pos_index]):               # :105-127
                y0, x0, y1, x1 = list(map(int, sample_roi[i, :4]))
                kp = mask[idx]      # view: mutated IN PLACE like the reference (:112-115)
                if not mutate_gt:   # the device kernel's deliberate fix: work on a copy
                    kp = kp.copy()
                kp[:, :2] = (kp[:, :2] - [y0, x0]) / [max(y1 - y0, 1), max(x1 - x0, 1)] * mask_size
                keypoint_labels = np.zeros(kp.shape[0], dtype=np.int32)
                for j, r in enumerate(kp):
                    y, x, v = list(map(int, r))
                    if v == 2 and 0 <= y and y < mask_size and 0 <= x and x < mask_size:
                        keypoint_labels[j] = y * mask_size + x
                    else:
                        keypoint_labels[j] = -1
                gt_roi_mask.append(keypoint_labels)
        gt_roi_mask = np.array(gt_roi_mask)
        out = (sample_roi, sample_levels, gt_roi_loc.astype(F), gt_roi_label, gt_roi_mask)
        if return_debug:
            return out, dict(keep_index=keep_index, gt_assignment=gt_assignment, max_iou=max_iou,
                             n_pos=pos_roi_per_this_image)
        return out


def proposal_targets_from_keys(roi, bbox, label, keys, n_sample=256, pos_ratio=0.25,
                               pos_iou_thresh=0.5, neg_iou_thresh_hi=0.5, neg_iou_thresh_lo=0.0):
    """Key-driven sampler used to pin the DEVICE sampler: identical candidate sets to
    ProposalTargetCreator (:63-78) but the random subset is "the k candidates with the
    smallest (key, index)" for caller-supplied uint32 ``keys`` (one per roi+gt row), and the
    selected rows keep ascending index order.  Uniform keys give the same distribution as
    ``np.random.choice(replace=False)``.  Returns (keep_index, n_pos, gt_assignment, max_iou)."""
    roi = np.concatenate((np.asarray(roi, F), np.asarray(bbox, F)), axis=0)
    iou = bbox_iou(roi, bbox)
    gt_assignment = iou.argmax(axis=1)
    max_iou = iou.max(axis=1)
    keys = np.asarray(keys, np.uint64)
    comp = (keys << np.uint64(32)) | np.arange(roi.shape[0], dtype=np.uint64)

    def pick(cand, k):
        if cand.size <= k:
            return cand
        sel = cand[np.argsort(comp[cand], kind='stable')[:k]]
        return np.sort(sel)

    pos = np.where(max_iou >= F(pos_iou_thresh))[0]
    n_pos = int(min(np.round(n_sample * pos_ratio), pos.size))
    pos = pick(pos, n_pos)
    neg = np.where((max_iou < F(neg_iou_thresh_hi)) & (max_iou >= F(neg_iou_thresh_lo)))[0]
    n_neg = int(min(n_sample - n_pos, neg.size))
    neg = pick(neg, n_neg)
    return np.append(pos, neg), n_pos, gt_assignment, max_iou


class AnchorTargetCreator(object):
    """ChainerCV ``AnchorTargetCreator``."""

    def __init__(self, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3, pos_ratio=0.5):
        self.n_sample = n_sample
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh = neg_iou_thresh
        self.pos_ratio = pos_ratio

    def labels_before_sampling(self, bbox, anchor, img_size):
        img_H, img_W = img_size
        anchor = np.asarray(anchor, F)
        inside = np.where((anchor[:, 0] >= 0) & (anchor[:, 1] >= 0) &
                          (anchor[:, 2] <= img_H) & (anchor[:, 3] <= img_W))[0]
        a = anchor[inside]
        ious = bbox_iou(a, bbox)
        argmax_ious = ious.argmax(axis=1)
        max_ious = ious[np.arange(len(inside)), argmax_ious]
        gt_argmax = ious.argmax(axis=0)
        gt_max = ious[gt_argmax, np.arange(ious.shape[1])]
        gt_argmax = np.where(ious == gt_max)[0]
        label = np.full((len(inside),), -1, np.int32)
        label[max_ious < self.neg_iou_thresh] = 0
        label[gt_argmax] = 1
        label[max_ious >= self.pos_iou_thresh] = 1
        return inside, a, argmax_ious, max_ious, label

    def __call__(self, bbox, anchor, img_size, rng=None):
        rng = rng if rng is not None else np.random
        bbox = np.asarray(bbox, F)
        n_anchor = len(anchor)
        inside, a, argmax_ious, _, label = self.labels_before_sampling(bbox, anchor, img_size)
        n_pos = int(self.pos_ratio * self.n_sample)
        pos_index = np.where(label == 1)[0]
        if len(pos_index) > n_pos:
            disable = rng.choice(pos_index, size=(len(pos_index) - n_pos), replace=False)
            label[disable] = -1
        n_neg = self.n_sample - np.sum(label == 1)
        neg_index = np.where(label == 0)[0]
        if len(neg_index) > n_neg:
            disable = rng.choice(neg_index, size=(len(neg_index) - n_neg), replace=False)
            label[disable] = -1
        loc = bbox2loc(a, bbox[argmax_ious])
        full_label = np.full((n_anchor,), -1, np.int32)
        full_label[inside] = label
        full_loc = np.zeros((n_anchor, 4), F)
        full_loc[inside] = loc
        return full_loc, full_label


def anchor_targets_from_keys(bbox, anchor, img_size, keys, n_sample=256, pos_iou_thresh=0.7,
                             neg_iou_thresh=0.3, pos_ratio=0.5):
    """Key-driven AnchorTargetCreator used to pin the DEVICE sampler: identical labels before
    sampling; the random "disable" draws of the reference are replaced by "keep the k candidates
    with the smallest (key, anchor index)" for caller-supplied uint32 ``keys`` (one per anchor).
    Returns (loc (A,4) f32, label (A,) i32)."""
    atc = AnchorTargetCreator(n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio)
    bbox = np.asarray(bbox, F)
    n_anchor = len(anchor)
    inside, a, argmax_ious, _, label = atc.labels_before_sampling(bbox, anchor, img_size)
    comp = (np.asarray(keys, np.uint64)[inside] << np.uint64(31)) | inside.astype(np.uint64)

    def keep_smallest(cand, k):
        if cand.size > k:
            order = cand[np.argsort(comp[cand], kind='stable')]
            label[order[k:]] = -1

    n_pos = int(pos_ratio * n_sample)
    keep_smallest(np.where(label == 1)[0], n_pos)
    n_neg = n_sample - int(np.sum(label == 1))
    keep_smallest(np.where(label == 0)[0], n_neg)
    loc = bbox2loc(a, bbox[argmax_ious])
    full_label = np.full((n_anchor,), -1, np.int32)
    full_label[inside] = label
    full_loc = np.zeros((n_anchor, 4), F)
    full_loc[inside] = loc
    return full_loc, full_label
