#!/usr/bin/env python3
"""Golden vector of ONE WHOLE INFERENCE CALL made by executing the reference's own code (build container only).

Run from the repo root:   python tests/golden/make_predict_reference.py          (needs /root/reference; ~1 min)

Executed from /root/reference, unmodified:  MaskRCNN.predict (chainer_maskrcnn/model/maskrcnn.py:157-259) and everything
it calls in the tree - MaskRCNN.prepare (:261-276), MaskRCNN.__call__ in inference mode (:135-155), the FPN extractor, the
multilevel RPN with the test-time proposal counts, FPNRoIMaskHead.__call__ (box branch, caches the pyramid) and
.predict_mask (:92-104), MaskRCNN._suppress (:278-312), the mask paste loop (:231-246).

Third-party calls are served by tests/golden/mini_chainer.py (inference-mode BatchNormalization = running statistics,
using_config, softmax, sigmoid) and this repo's oracle (ProposalCreator with the test preset, loc2bbox, NMS with scores,
ROIAlign, chainercv resize / cv2.resize as the float32 INTER_LINEAR restatement, cv2.threshold): see make_step_reference.py.
The running statistics are the batch statistics of a training-mode pass over the same prepared image (what training leaves
behind, up to the decay), taken once and stored in the fixture, so the inference-mode activations stay in range with
seeded weights.  The fixture pins the WIRING of the inference path: prepare's size rule, scale handling (RoIs in network
coordinates, boxes in image coordinates), class-agnostic loc tiling, de-normalisation, clipping, soft-max, per-class
suppression incl. the skipped last class, mask head on detections * scale, channel = label, paste offsets.

Only data is stored: predict_reference.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = '/root/reference'

import make_step_reference as gs        # noqa: E402
import mini_chainer as mc               # noqa: E402
from weights import chainer_weights, layer_list     # noqa: E402
from oracle import predict as opredict  # noqa: E402
from oracle import roi_align as oroi    # noqa: E402
from oracle import targets as otargets  # noqa: E402

WEIGHT_SEED = 20261


def main():
    gs.install()
    np.bool = bool                       # removed from NumPy 1.24+; maskrcnn.py:221,246 use it
    cv2 = sys.modules['cv2']

    def cv2_resize(a, dsize):            # INTER_LINEAR: uint8 (targets) or float32 (mask paste, maskrcnn.py:236)
        dsize = (int(dsize[0]), int(dsize[1]))
        a = np.asarray(a)
        return otargets.cv2_resize_linear_u8(a, dsize) if a.dtype == np.uint8 else opredict.cv2_resize_linear_f32(a.astype(np.float32), dsize)
    cv2.resize = cv2_resize
    cv2.THRESH_BINARY = 0
    cv2.threshold = lambda m, thresh, maxval, kind: (float(thresh), np.where(m > thresh, maxval, 0).astype(m.dtype))
    sys.path.insert(0, REF)
    from chainer_maskrcnn.model import maskrcnn as ref_m
    import chainer_maskrcnn.functions.roi_align_2d_yx as ref_yx
    ref_yx.roi_align_2d = lambda x, rois_xy, outh, outw, s: mc.V(oroi.roi_align_fwd(np.asarray(x), np.asarray(rois_xy), outh, outw, s, 2))
    ref_m.non_maximum_suppression = lambda bbox, thresh, score=None, limit=None: opredict.nms_with_score(np.asarray(bbox, np.float32), thresh, np.asarray(score, np.float32))
    ref_m.resize = lambda img, size: np.stack([opredict.cv2_resize_linear_f32(img[c], (size[1], size[0])) for c in range(img.shape[0])])

    H, W = 96, 128
    rs = np.random.RandomState(91)
    img = np.floor(rs.rand(3, H, W) * 256).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for cy, cx, ry, rx, v in ((30, 40, 18, 25, 90.0), (60, 95, 25, 20, -70.0), (75, 25, 12, 16, 60.0)):      # a few blobs
        img[:, ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1] += v
    img = np.clip(img, 0, 255).astype(np.float32)

    t0 = time.time()
    weights = chainer_weights(WEIGHT_SEED)
    model = ref_m.MaskRCNN(n_fg_class=80, min_size=128, max_size=256)
    gs.assign(model, weights)
    model.use_preset('evaluate')

    # ---- running statistics: one training-mode pass of the extractor over the prepared image
    bn_names = [n for n, kind, _, _ in layer_list() if kind == 'bn']
    prepared = model.prepare(img)
    mc.config.train = True
    model.extractor(mc.V(prepared[None]))
    stats = {}
    for n in bn_names:
        bn = model
        for p in n.split('/'):
            bn = getattr(bn, p)
        stats[n + '/avg_mean'] = bn.last_mean.astype(np.float32)
        stats[n + '/avg_var'] = bn.last_var.astype(np.float32)
        bn.avg_mean, bn.avg_var = mc.V(stats[n + '/avg_mean']), mc.V(stats[n + '/avg_var'])

    # ---- a first inference call to place the score threshold where ~40 (RoI, class) pairs pass (random weights: the class
    #      probabilities sit near 1/81); the attribute is a plain float of the reference model
    cap = {}

    def tap(name, fn):
        def f(*a, **k):
            out = fn(*a, **k)
            cap[name] = (a, out)
            return out
        return f
    with mc.using_config('train', False):
        roi_cls_locs, roi_scores, rois, roi_indices, levels = model(mc.V(prepared[None]), scale=prepared.shape[2] / W)
    prob = np.asarray(mc.softmax(roi_scores))
    model.score_thresh = float(np.sort(prob[:, 1:80].ravel())[-120])
    print('proposals', rois.shape, 'score_thresh', model.score_thresh)

    model._suppress = tap('suppress', model._suppress)
    call = type(model).__call__
    type(model).__call__ = lambda self, x, scale=1.: tap('call', lambda: call(self, x, scale=scale))()
    mc.config.train = True               # predict() switches to inference by itself (using_config)
    masks, labels, scores = model.predict([img])
    type(model).__call__ = call
    print('reference predict executed in %.1f s' % (time.time() - t0))
    roi_cls_locs, roi_scores, rois, roi_indices, levels = cap['call'][1]
    bbox, label, score, det_roi, det_level = cap['suppress'][1]
    assert mc.config.train is True and len(label) >= 10 and len(np.unique(label)) >= 3, (len(label), np.unique(label))
    assert np.array_equal(labels[0], label) and np.array_equal(scores[0], score)
    mk = masks[0]
    assert mk.shape == (len(label), H, W) and mk.dtype == bool and 0 < mk.sum() < mk.size
    f32 = lambda a: np.asarray(a, np.float32)
    out = {'in_img': img, 'in_min_max': np.array([128, 256]), 'in_weight_seed': np.int64(WEIGHT_SEED), 'in_score_thresh': np.float64(model.score_thresh),
           'in_nms_thresh': np.float64(model.nms_thresh), 'prepared': f32(prepared), 'rois': f32(rois), 'levels': np.asarray(levels, np.int32),
           'roi_cls_locs': f32(roi_cls_locs), 'roi_scores': f32(roi_scores), 'bbox': f32(bbox), 'label': np.asarray(label, np.int32),
           'score': f32(score), 'det_level': np.asarray(det_level, np.int32), 'masks': np.packbits(mk, axis=-1), 'masks_shape': np.array(mk.shape)}
    for k, v in stats.items():
        out['bn/' + k] = v
    path = os.path.join(HERE, 'predict_reference.npz')
    np.savez_compressed(path, **out)
    print('predict_reference.npz %.2f MB' % (os.path.getsize(path) / 2 ** 20), 'prepared', prepared.shape, 'rois', rois.shape, 'detections', len(label),
          'classes', len(np.unique(label)), 'mask pixels', int(mk.sum()))


if __name__ == '__main__':
    main()
