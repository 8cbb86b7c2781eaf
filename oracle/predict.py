"""Oracle: inference post-processing (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates model/maskrcnn.py:178-210 (decode, clip, softmax), :278-312 (``_suppress``) and :231-246 (mask paste) in
NumPy.  Third-party pieces ("parity unpinned"): ChainerCV ``loc2bbox`` and
``non_maximum_suppression(bbox, thresh, score)`` (sorts by descending score - tie order pinned like
boxes.argsort_desc_pinned - then greedy NMS, returns indices in that order), ``F.softmax``, and OpenCV
``cv2.resize`` on float32 (INTER_LINEAR: half-pixel centres, edge clamp, float coefficients, horizontal then vertical).
"""
import numpy as np

from .boxes import loc2bbox, nms, argsort_desc_pinned

F = np.float32


def decode(rois, roi_cls_loc, roi_score, scale, size, n_class, mean=(0., 0., 0., 0.), std=(0.1, 0.1, 0.2, 0.2)):
    """maskrcnn.py:178-205 for the class-agnostic regressor: every class shares the box.  Returns (bbox (R,4), prob)."""
    roi = (np.asarray(rois, F) / F(scale)).astype(F)
    loc = (np.asarray(roi_cls_loc, F) * np.asarray(std, F) + np.asarray(mean, F)).astype(F)
    bbox = loc2bbox(roi, loc)
    bbox[:, 0::2] = np.clip(bbox[:, 0::2], 0, size[0])
    bbox[:, 1::2] = np.clip(bbox[:, 1::2], 0, size[1])
    x = np.asarray(roi_score, F)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    prob = (e / e.sum(axis=1, keepdims=True, dtype=F)).astype(F)
    return bbox, prob


def nms_with_score(bbox, thresh, score):
    order = argsort_desc_pinned(score)
    sel = nms(bbox[order], thresh)
    return order[sel]


def suppress(cls_bbox, prob, n_class, nms_thresh, score_thresh, predict_mask=True):
    """maskrcnn.py:278-312.  Returns (selected RoI indices, labels) concatenated over classes."""
    idx, lab = [], []
    for l in range(1, n_class):
        if predict_mask and l == n_class - 1:
            continue
        m = np.nonzero(prob[:, l] > F(score_thresh))[0]
        keep = nms_with_score(cls_bbox[m], nms_thresh, prob[m, l])
        idx.append(m[keep])
        lab.append(np.full(len(keep), l - 1, np.int32))
    return np.concatenate(idx).astype(np.int64), np.concatenate(lab)


def prepare(img, min_size=600, max_size=1000):
    """MaskRCNN.prepare (maskrcnn.py:261-276): (3,H,W) float32 0..255 -> resized so that the short side is min_size unless the
    long side would exceed max_size (chainercv.transforms.resize = cv2.resize INTER_LINEAR per channel), then / 255.  The size
    rule and the scaling are pinned by tests/golden/prepare_reference.npz (the reference method executed in the build
    container); the interpolation is this oracle's cv2 restatement."""
    _, H, W = img.shape
    scale = min_size / min(H, W)
    if scale * max(H, W) > max_size:
        scale = max_size / max(H, W)
    oh, ow = int(H * scale), int(W * scale)
    out = np.stack([cv2_resize_linear_f32(img[c], (ow, oh)) for c in range(img.shape[0])])
    return out.astype(np.float32) / 255


def cv2_resize_linear_f32(src, dsize):
    """cv2.resize(src, (dw, dh)) for a 2-D float32 image."""
    src = np.asarray(src, F)
    sh, sw = src.shape
    dw, dh = dsize

    def coef(ssize, dsz):
        scale = 1.0 / (float(dsz) / float(ssize))
        i0 = np.zeros(dsz, np.int64)
        f = np.zeros(dsz, F)
        for d in range(dsz):
            v = F((d + 0.5) * scale - 0.5)
            s = int(np.floor(v))
            v = F(v - F(s))
            if s < 0:
                v, s = F(0), 0
            if s >= ssize - 1:
                v, s = F(0), ssize - 1
            i0[d], f[d] = s, v
        return i0, f
    x0, fx = coef(sw, dw)
    y0, fy = coef(sh, dh)
    x1 = np.minimum(x0 + 1, sw - 1)
    y1 = np.minimum(y0 + 1, sh - 1)
    rows = (src[:, x0] * (F(1) - fx)[None, :] + src[:, x1] * fx[None, :]).astype(F)
    return (rows[y0] * (F(1) - fy)[:, None] + rows[y1] * fy[:, None]).astype(F)


def paste_masks(mask_logits_nchw, label, bbox, size):
    """maskrcnn.py:231-246.  mask_logits (D, n_fg, S, S); returns (D, H, W) bool."""
    D = len(bbox)
    out = np.zeros((D,) + tuple(size), bool)
    for i in range(D):
        m = (F(1) / (F(1) + np.exp(-mask_logits_nchw[i, label[i]].astype(F)))).astype(F)
        b = bbox[i]
        w, h = int(b[3] - b[1]), int(b[2] - b[0])
        if w <= 0 or h <= 0:
            continue
        mm = (cv2_resize_linear_f32(m, (w, h)) * F(255)).astype(np.uint8) > 127
        s, t = int(b[0]), int(b[1])
        hh, ww = min(h, size[0] - s), min(w, size[1] - t)
        out[i, s:s + hh, t:t + ww] = mm[:hh, :ww]
    return out
