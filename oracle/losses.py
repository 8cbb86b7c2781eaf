"""Oracle: the five training losses (TEST INFRASTRUCTURE - see oracle/__init__.py).

Third-party (Chainer / ChainerCV, absent => "parity unpinned"):
  smooth_l1_loss / fast_rcnn_loc_loss <- chainercv faster_rcnn_train_chain._smooth_l1_loss,
        _fast_rcnn_loc_loss; used at model/fpn_maskrcnn_train_chain.py:83-84,100-101
  softmax_cross_entropy               <- chainer.functions.softmax_cross_entropy
        (normalize=True, ignore_label=-1); used at :85,:102 and train_keypoints.py:24-27
  sigmoid_cross_entropy               <- chainer.functions.sigmoid_cross_entropy
        (normalize=True); used by calc_mask_loss train.py:50-58
Each function returns (loss, gradient w.r.t. the prediction) in float32.
"""
import numpy as np

F = np.float32


def smooth_l1_loss(x, t, in_weight, sigma):
    sigma2 = F(sigma) ** 2
    diff = in_weight * (x - t)
    abs_diff = np.abs(diff)
    flag = (abs_diff < F(1.) / sigma2).astype(F)
    y = flag * (sigma2 / F(2.)) * np.square(diff) + (F(1) - flag) * (abs_diff - F(0.5) / sigma2)
    gdiff = flag * sigma2 * diff + (F(1) - flag) * np.sign(diff)
    return np.sum(y, dtype=F), (gdiff * in_weight).astype(F)


def fast_rcnn_loc_loss(pred_loc, gt_loc, gt_label, sigma):
    pred_loc = np.asarray(pred_loc, F)
    gt_loc = np.asarray(gt_loc, F)
    in_weight = np.zeros_like(gt_loc)
    in_weight[np.asarray(gt_label) > 0] = 1
    loss, g = smooth_l1_loss(pred_loc, gt_loc, in_weight, sigma)
    norm = F(np.sum(np.asarray(gt_label) >= 0))
    return F(loss / norm), (g / norm).astype(F)


def softmax_cross_entropy(x, t, ignore_label=-1):
    x = np.asarray(x, F)
    t = np.asarray(t)
    m = x.max(axis=1, keepdims=True)
    z = x - m
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True, dtype=F))
    logp = z - lse
    valid = t != ignore_label
    count = max(int(valid.sum()), 1)
    tt = np.where(valid, t, 0)
    nll = -logp[np.arange(x.shape[0]), tt] * valid
    loss = F(nll.sum(dtype=F) / F(count))
    g = np.exp(logp)
    g[np.arange(x.shape[0]), tt] -= 1
    g = g * valid[:, None] / F(count)
    return loss, g.astype(F)


def sigmoid_cross_entropy(x, t, ignore_label=-1):
    x = np.asarray(x, F)
    t = np.asarray(t)
    valid = (t != ignore_label)
    count = max(int(valid.sum()), 1)
    tf = t.astype(F)
    loss_e = -(x * (tf - (x >= 0)) - np.log1p(np.exp(-np.abs(x))))
    loss = F((loss_e * valid).sum(dtype=F) / F(count))
    sig = F(1) / (F(1) + np.exp(-x))
    g = (sig - tf) * valid / F(count)
    return loss, g.astype(F)


def calc_mask_loss(roi_cls_mask, gt_roi_mask, gt_roi_label):
    """train.py:50-58: channel ``gt_label-1`` of each RoI, first n_pos rows, BCE-with-logits."""
    R = roi_cls_mask.shape[0]
    n_pos = gt_roi_mask.shape[0]
    sel = roi_cls_mask[np.arange(R), np.asarray(gt_roi_label) - 1][:n_pos]
    loss, g = sigmoid_cross_entropy(sel, gt_roi_mask)
    gfull = np.zeros_like(roi_cls_mask, dtype=F)
    gfull[np.arange(n_pos), np.asarray(gt_roi_label)[:n_pos] - 1] = g
    return loss, gfull


def calc_keypoint_loss(roi_cls_mask, gt_roi_mask):
    """train_keypoints.py:21-27: softmax-CE over H*W positions per (positive RoI, keypoint)."""
    n_pos, K = gt_roi_mask.shape[:2]
    x = roi_cls_mask[:n_pos].reshape(n_pos * K, -1)
    loss, g = softmax_cross_entropy(x, gt_roi_mask.reshape(-1))
    gfull = np.zeros_like(roi_cls_mask, dtype=F)
    gfull[:n_pos] = g.reshape((n_pos,) + roi_cls_mask.shape[1:])
    return loss, gfull
