"""Device ``MaskRCNN.predict`` against a whole inference call EXECUTED BY THE REFERENCE'S OWN CODE
(tests/golden/make_predict_reference.py -> tests/golden/predict_reference.npz; the oracle's pin on the same fixture is
tests/test_predict_reference_cpu.py).  Full-width ResNet-50 FPN, seeded Chainer-layout weights + the stored running
statistics loaded through ChainerNpzMap.from_chainer, the reference's image, its score threshold.

The reference ran in float64; the device runs float32 kernels, so proposals / detections whose score order or IoU test sits
inside that difference may differ - detections are matched by (label, box) and the bars are: >= 90 % of the reference's
detections found with boxes within 0.5 px and scores within 2 % (random weights: the class scores are steep functions of
the box), their pasted masks differing in <= 2 % of the mask pixels; the prepared image is compared bit for bit, the RoIs
by matching, the box head on the reference's own RoIs at 1e-3."""
import os

import numpy as np
import pytest
import torch

from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.nn import core
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap
from test_predict_reference_cpu import load_predict_golden, golden_arrays

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_device_predict_equals_reference_executed_predict():
    d = load_predict_golden(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    img = d['in_img']
    H, W = img.shape[1:]
    mn, mx = (int(v) for v in d['in_min_max'])
    m = MaskRCNN(n_fg_class=80, device=DEV, seed=1, min_size=mn, max_size=mx)
    arrays = golden_arrays(d)
    assert set(ChainerNpzMap(m).from_chainer(arrays, strict=False)) == set(arrays)
    m.use_preset('evaluate')
    m.score_thresh = float(d['in_score_thresh'])
    assert m.nms_thresh == float(d['in_nms_thresh'])
    keep_train = core.TRAIN
    try:
        masks, labels, scores = m.predict([torch.from_numpy(img)])
    finally:
        core.TRAIN = keep_train
    # ---- prepare: bit-exact
    np.testing.assert_array_equal(m.prepare(torch.from_numpy(img).to(DEV)).cpu().numpy(), d['prepared'])
    # ---- proposals of the inference-mode RPN (6000 -> 300, here all that survive NMS)
    rois = m.last_rois.cpu().numpy()
    dist = np.abs(rois[:, None, :] - d['rois'][None, :, :]).max(-1)
    hit = dist.min(1) < 0.05
    print('proposals: device %d, reference %d, matched %d' % (len(rois), len(d['rois']), int(hit.sum())))
    assert abs(len(rois) - len(d['rois'])) <= 0.03 * len(d['rois']) and hit.mean() >= 0.97
    # ---- box head of the same call on the REFERENCE's proposals (the device's own differ by up to the matching distance,
    #      which moves pooled features by more than the 1e-3 bar): the pyramid predict() cached, the reference's RoIs
    R = len(d['rois'])
    dev = torch.device(DEV)
    xy5 = torch.from_numpy(np.concatenate([np.zeros((R, 1), np.float32), d['rois'][:, [1, 0, 3, 2]]], 1)).to(dev).contiguous()
    box = m.head.box_branch(m.head.x, xy5, torch.from_numpy(d['levels']).to(dev), m.extractor.spatial_scales).cpu().numpy()
    nc, l0 = m.head.n_class, m.head.LOC0
    err_s = float(np.abs(box[:, :nc] - d['roi_scores']).max()) / max(float(np.abs(d['roi_scores']).max()), 1e-30)
    err_l = float(np.abs(box[:, l0:l0 + 4] - d['roi_cls_locs']).max()) / max(float(np.abs(d['roi_cls_locs']).max()), 1e-30)
    print('box head on the reference RoIs: scores %.2e, locs %.2e of scale' % (err_s, err_l))
    assert err_s < 1e-3 and err_l < 1e-3
    # ---- detections
    bbox = m.last_bboxes[0].cpu().numpy()
    label, score, mk = labels[0].cpu().numpy(), scores[0].cpu().numpy(), masks[0].cpu().numpy()
    assert mk.shape[1:] == (H, W) and mk.dtype == bool
    found, mask_diff, mask_pix, berr, serr = 0, 0, 0, [], []
    for i in range(len(d['label'])):
        cand = np.nonzero(label == d['label'][i])[0]
        if not len(cand):
            continue
        e = np.abs(bbox[cand] - d['bbox'][i]).max(1)
        k = cand[e.argmin()]
        berr.append(float(e.min())); serr.append(abs(float(score[k]) - float(d['score'][i])) / float(d['score'][i]))
        if e.min() < 0.5 and serr[-1] <= 2e-2:
            found += 1
            mask_diff += int((mk[k] != d['masks'][i]).sum())
            mask_pix += int(d['masks'][i].sum())
    print('detections: device %d, reference %d, matched %d; mask pixels differing %d of %d' % (len(label), len(d['label']), found, mask_diff, mask_pix))
    print('nearest same-label box distance (px): median %.3g, p90 %.3g, max %.3g; score deviation: median %.2e, p90 %.2e, max %.2e'
          % (np.median(berr), np.percentile(berr, 90), max(berr), np.median(serr), np.percentile(serr, 90), max(serr)))
    assert found >= 0.9 * len(d['label']) and abs(len(label) - len(d['label'])) <= 0.1 * len(d['label'])
    assert mask_diff <= 0.02 * max(mask_pix, 1)
