"""CPU pin of the oracle's INFERENCE path (oracle/model.py in inference mode + oracle/predict.py + oracle/proposal.py - what
tests/test_config0_gpu.py and tests/test_predict_gpu.py check the device against) on a whole MaskRCNN.predict call EXECUTED
BY THE REFERENCE'S OWN CODE (tests/golden/make_predict_reference.py -> tests/golden/predict_reference.npz: prepare, __call__
in inference mode, FPNRoIMaskHead.__call__ / predict_mask, _suppress, the paste loop; Chainer / ChainerCV / cv2 primitives
served by float64 stand-ins and this repo's oracle)."""
import os
import sys

import numpy as np
import torch

from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap
from oracle import boxes as oboxes
from oracle import predict as opredict
from oracle import proposal as oproposal
from oracle.model import OracleStep, D

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from weights import chainer_weights      # noqa: E402


def load_predict_golden(golden_dir):
    d = dict(np.load(os.path.join(golden_dir, 'predict_reference.npz')))
    shape = tuple(d['masks_shape'])
    d['masks'] = np.unpackbits(d['masks'], axis=-1)[..., :shape[-1]].reshape(shape).astype(bool)
    return d


def golden_arrays(d):
    """Chainer-layout snapshot: seeded weights + the stored BatchNorm running statistics."""
    arrays = chainer_weights(int(d['in_weight_seed']))
    arrays.update({k[3:]: v for k, v in d.items() if k.startswith('bn/')})
    return arrays


def _rel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    return float(np.abs(got - want).max()) / max(float(np.abs(want).max()), 1e-30)


def test_oracle_inference_equals_reference_executed_predict(golden_dir):
    d = load_predict_golden(golden_dir)
    img = d['in_img']
    H, W = img.shape[1:]
    mn, mx = (int(v) for v in d['in_min_max'])
    m = MaskRCNN(n_fg_class=80, device='cpu', seed=1, min_size=mn, max_size=mx)
    arrays = golden_arrays(d)
    assert set(ChainerNpzMap(m).from_chainer(arrays, strict=False)) == set(arrays)
    # ---- prepare (maskrcnn.py:261-276)
    x = opredict.prepare(img, mn, mx)
    np.testing.assert_array_equal(x, d['prepared'])
    Hp, Wp = x.shape[1:]
    scale = Wp / W                                                                   # maskrcnn.py:173
    # ---- __call__ in inference mode (:135-155)
    ps = m.ps
    params = {n: ps.p(n).detach().to(D) for n in ps.names()}
    o = OracleStep(params, (3, 4, 6, 3), m.head.n_class, m.head.LOC0, bn_buffers={k: v.detach() for k, v in ps.buffers.items()})
    img4 = torch.cat([torch.from_numpy(x[None]).permute(0, 2, 3, 1), torch.zeros((1, Hp, Wp, 1))], -1).to(D)
    with torch.no_grad():
        feats = o.extractor(img4)
        locs, scores = o.rpn(feats)
        anchor = oboxes.fpn_anchors([tuple(f.shape[1:3]) for f in feats])
        rois = oproposal.ProposalCreator()(locs[0].numpy().astype(np.float32), scores[0, :, 1].numpy().astype(np.float32), anchor, (Hp, Wp),
                                           scale=scale, train=False)
        np.testing.assert_allclose(rois, d['rois'], rtol=0, atol=1e-3)
        levels = np.clip(oboxes.map_rois_to_fpn_levels(rois), 0, 4).astype(np.int32)         # :141
        np.testing.assert_array_equal(levels, d['levels'])
        R = len(rois)
        xy5 = np.concatenate([np.zeros((R, 1), np.float32), d['rois'][:, [1, 0, 3, 2]]], 1)
        box = o.head_box(feats, xy5, levels).numpy()
        nc, l0 = m.head.n_class, m.head.LOC0
        assert _rel(box[:, :nc], d['roi_scores']) < 1e-5 and _rel(box[:, l0:l0 + 4], d['roi_cls_locs']) < 1e-5
        # ---- decode, soft-max, per-class suppression (:176-216, 278-312) on the reference's own head outputs
        cls_bbox, prob = opredict.decode(d['rois'], d['roi_cls_locs'], d['roi_scores'], scale, (H, W), nc)
        idx, lab = opredict.suppress(cls_bbox, prob, nc, float(d['in_nms_thresh']), float(d['in_score_thresh']), predict_mask=True)
        np.testing.assert_array_equal(lab, d['label'])
        np.testing.assert_allclose(cls_bbox[idx], d['bbox'], rtol=0, atol=1e-3)
        np.testing.assert_allclose(prob[idx, lab + 1], d['score'], rtol=1e-5, atol=1e-7)
        np.testing.assert_array_equal(levels[idx], d['det_level'])
        # ---- mask head on the detections (in network coordinates) and the paste (:218-246)
        Dn = len(lab)
        dxy5 = np.concatenate([np.zeros((Dn, 1), np.float32), (d['bbox'] * np.float32(scale))[:, [1, 0, 3, 2]]], 1).astype(np.float32)
        wm = o.head_mask(feats, dxy5, d['det_level'])[..., :80].permute(0, 3, 1, 2).numpy()
        masks = opredict.paste_masks(wm, d['label'], d['bbox'], (H, W))
    assert masks.shape == d['masks'].shape
    diff = int((masks != d['masks']).sum())
    assert diff <= 1e-3 * max(int(d['masks'].sum()), 1), (diff, int(d['masks'].sum()))
