#!/usr/bin/env python3
"""Golden vector of ONE WHOLE TRAINING-STEP FORWARD made by executing the reference's own model code (build container only).

Run from the repo root:   python tests/golden/make_step_reference.py [mask|keypoint]      (needs /root/reference; ~1 min each;
                          one kind per process - the reference's modules are imported once)

Executed from /root/reference, unmodified, in this process:
    MaskRCNN.__init__                              chainer_maskrcnn/model/maskrcnn.py:26-135
    FeaturePyramidNetwork.__init__/__call__        chainer_maskrcnn/model/extractor/feature_pyramid_network.py:18-72
    MultilevelRegionProposalNetwork.__call__       chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:90-170
    map_rois_to_fpn_levels                         ... :16-31
    FPNRoIMaskHead.__init__/__call__               chainer_maskrcnn/model/head/fpn_roi_mask_head.py:13-90
    _roi_align_2d_yx                               chainer_maskrcnn/functions/roi_align_2d_yx.py:4-7
    FPNMaskRCNNTrainChain.__init__/__call__        chainer_maskrcnn/model/fpn_maskrcnn_train_chain.py:15-123
    ProposalTargetCreator.__call__                 chainer_maskrcnn/utils/proposal_target_creator.py:26-137
    calc_mask_loss                                 train.py:49-57

The third-party packages they call (Chainer, ChainerCV, cv2, the absent roi_align submodule) are served by
tests/golden/mini_chainer.py (float64 restatements of the documented Chainer / ChainerCV semantics) and by this repo's
oracle (anchors, ProposalCreator, AnchorTargetCreator, bbox utilities, ROIAlign, cv2 resize).  The fixture therefore pins
the reference's WIRING of a training step - layer order, activations, pooling, top-down pathway, anchor enumeration
order, loc/score reshapes, proposal -> level -> sampling -> pooling flow, row/channel selection of the four head losses
and of the mask loss - against which oracle/model.py (the CPU restatement the device step is compared with) and the
device model are checked.  Weights come from tests/golden/weights.py (seeded NumPy), stored in Chainer's layouts and
assigned to the reference's link attributes by their snapshot key ('extractor/resnet/res2/a/conv1/W' ...).

Only data (inputs, sampled targets, activations, losses) is stored: step_reference.npz."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = '/root/reference'

import make_reference_vectors as g      # noqa: E402  (placeholder machinery, synthetic case maker)
import mini_chainer as mc               # noqa: E402
from weights import chainer_weights     # noqa: E402
from oracle import boxes as oboxes      # noqa: E402
from oracle import roi_align as oroi    # noqa: E402
from oracle import targets as otargets  # noqa: E402

WEIGHT_SEED, NP_SEED = 20260, 515


def install():
    cuda = g._mod('chainer.cuda', get_array_module=lambda *a: np, to_cpu=lambda x: np.asarray(x), to_gpu=lambda x: x, available=False)
    backends = g._mod('chainer.backends', cuda=cuda)
    g._mod('chainer.backends.cuda', get_array_module=lambda *a: np)
    links = g._mod('chainer.links', Convolution2D=mc.Convolution2D, Linear=mc.Linear, Deconvolution2D=mc.Deconvolution2D,
                   BatchNormalization=mc.BatchNormalization)
    g._mod('chainer.links.model'); g._mod('chainer.links.model.vision')
    g._mod('chainer.links.model.vision.resnet', ResNet50Layers=mc.ResNet50Layers, BuildingBlock=mc.BuildingBlock,
           _global_average_pooling_2d=mc._global_average_pooling_2d)
    functions = g._mod('chainer.functions', relu=mc.relu, max_pooling_2d=mc.max_pooling_2d, unpooling_2d=mc.unpooling_2d,
                       concat=mc.concat, softmax_cross_entropy=mc.softmax_cross_entropy, sigmoid_cross_entropy=mc.sigmoid_cross_entropy,
                       softmax=mc.softmax, sigmoid=mc.sigmoid, resize_images=mc.resize_images)
    reporter = g._mod('chainer.reporter', report=mc.report)
    initializers = g._mod('chainer.initializers', Normal=mc._Normal)
    g._mod('chainer', cuda=cuda, backends=backends, links=links, functions=functions, reporter=reporter, initializers=initializers,
           Chain=mc.Chain, ChainList=mc.ChainList, Variable=mc.Var, config=mc.config, using_config=mc.using_config)
    base = 'chainercv.links.model.faster_rcnn'
    for n in ('chainercv', 'chainercv.links', 'chainercv.links.model', base, base + '.utils', 'chainercv.utils', 'chainercv.utils.bbox'):
        g._mod(n)
    g._mod(base + '.faster_rcnn', FasterRCNN=mc.FasterRCNN)
    g._mod(base + '.faster_rcnn_train_chain', FasterRCNNTrainChain=mc.FasterRCNNTrainChain, _smooth_l1_loss=mc._smooth_l1_loss,
           _fast_rcnn_loc_loss=mc._fast_rcnn_loc_loss)
    g._mod(base + '.region_proposal_network', _enumerate_shifted_anchor=oboxes.enumerate_shifted_anchor)
    g._mod(base + '.utils.generate_anchor_base', generate_anchor_base=oboxes.generate_anchor_base)
    g._mod(base + '.utils.proposal_creator', ProposalCreator=mc.ProposalCreator)
    g._mod(base + '.utils.anchor_target_creator', AnchorTargetCreator=otargets.AnchorTargetCreator)
    g._mod(base + '.utils.bbox2loc', bbox2loc=oboxes.bbox2loc)
    g._mod(base + '.utils.loc2bbox', loc2bbox=oboxes.loc2bbox)
    g._mod('chainercv.utils.bbox.bbox_iou', bbox_iou=oboxes.bbox_iou)
    g._mod('cv2', resize=lambda a, dsize: otargets.cv2_resize_linear_u8(a, dsize))
    sys.meta_path.insert(0, g._PlaceholderFinder())
    if not hasattr(np, 'asscalar'):                  # removed from NumPy 1.23+; the train chain calls it (:39)
        np.asscalar = lambda a: np.asarray(a).item()


def assign(model, weights):
    """chainer.serializers.load_npz semantics: 'a/b/c/W' -> model.a.b.c.W"""
    for key, arr in weights.items():
        obj = model
        parts = key.split('/')
        for p in parts[:-1]:
            obj = getattr(obj, p)
        assert hasattr(obj, parts[-1]), key
        setattr(obj, parts[-1], mc.V(arr))


def main(kind='mask'):
    """kind 'mask': train.py's model (FPNRoIMaskHead, 80 classes, calc_mask_loss) -> step_reference.npz;
    kind 'keypoint': train_keypoints.py's (FPNRoIKeypointHead, 1 class, 17 keypoints, 8 convolutions, its calc_mask_loss =
    soft-max cross-entropy over the 56x56 positions, binary_mask=False) -> step_keypoint_reference.npz."""
    install()
    sys.path.insert(0, REF)
    from chainer_maskrcnn.model.maskrcnn import MaskRCNN
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain
    import chainer_maskrcnn.functions.roi_align_2d_yx as ref_yx
    if kind == 'mask':
        import train as ref_train
    else:
        import train_keypoints as ref_train
    assert os.path.dirname(os.path.abspath(ref_train.__file__)) == REF
    K = 17

    def roi_align_2d(x, rois_xy, outh, outw, spatial_scale):       # the absent submodule's operator: this repo's oracle
        return mc.V(oroi.roi_align_fwd(np.asarray(x), np.asarray(rois_xy), outh, outw, spatial_scale, 2))
    ref_yx.roi_align_2d = roi_align_2d

    H, W, G = 128, 160, 4
    _, bbox, label, mask = g.synth_case(81, 8, G, H, W)
    if kind == 'keypoint':
        _, _, _, kps = g.synth_case(81, 8, G, H, W, keypoints=True)         # same boxes (same seed), (y, x, v) keypoints inside them
        kps[:, :, 2] = np.where(kps[:, :, 2] == 0, 2, kps[:, :, 2])           # mostly visible: enough labelled positions
        label = np.zeros(G, np.int32)
    img = np.floor(np.random.RandomState(82).rand(1, 3, H, W) * 256).astype(np.float32)
    img -= np.array([122.7717, 115.9465, 102.9801], np.float32)[None, :, None, None]
    for gi in range(G):           # the objects are visible in the image, so that the features are not pure noise
        img[0][:, mask[gi] > 0] += np.float32(40.0 * (gi + 1) / G)

    t0 = time.time()
    if kind == 'mask':
        weights = chainer_weights(WEIGHT_SEED)
        model = MaskRCNN(n_fg_class=80)
        assign(model, weights)
        chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=ref_train.calc_mask_loss)
        gt_in = mask
    else:
        weights = chainer_weights(WEIGHT_SEED + 1, n_fg_class=1, n_keypoints=K)
        model = MaskRCNN(n_fg_class=1, n_keypoints=K, head_arch='fpn_keypoint')          # train_keypoints.py:119-120
        assign(model, weights)
        chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=ref_train.calc_mask_loss, binary_mask=False)
        gt_in = kps.copy()               # the reference's ProposalTargetCreator writes into it

    cap = {}

    class Tap(object):
        def __init__(self, fn, name):
            self.fn, self.name = fn, name

        def __call__(self, *a, **k):
            out = self.fn(*a, **k)
            cap[self.name] = (a, out)
            return out

        def __getattr__(self, n):
            return getattr(self.fn, n)
    # record what flows between the reference's components (arguments and results), without touching them
    model.extractor, model.rpn, model.head = Tap(model.extractor, 'extractor'), Tap(model.rpn, 'rpn'), Tap(model.head, 'head')
    chain.proposal_target_creator = Tap(chain.proposal_target_creator, 'ptc')
    chain.anchor_target_creator = Tap(chain.anchor_target_creator, 'atc')

    np.random.seed(NP_SEED)
    loss = chain(mc.V(img), bbox[None], label[None], gt_in[None], np.array(1.0, np.float32))
    print('reference step executed in %.1f s' % (time.time() - t0), dict(mc.REPORTED))

    feats = cap['extractor'][1]
    rpn_locs, rpn_scores, rois, roi_indices, anchor, levels = cap['rpn'][1]
    sample_roi, sample_levels, gt_roi_loc, gt_roi_label, gt_roi_mask = cap['ptc'][1]
    gt_rpn_loc, gt_rpn_label = cap['atc'][1]
    (_, indices_and_rois, head_levels, _), (roi_cls_locs, roi_scores, roi_cls_mask) = cap['head']
    n_pos = gt_roi_mask.shape[0]
    assert n_pos >= 4 and (gt_rpn_label == 1).sum() >= 2 and len(np.unique(np.asarray(sample_levels))) >= 2, \
        (n_pos, (gt_rpn_label == 1).sum(), np.unique(np.asarray(sample_levels)))
    assert abs(float(loss) - sum(v for k, v in mc.REPORTED.items() if k != 'loss')) < 1e-9
    f32 = lambda a: np.asarray(a, np.float32)
    out = {
        'in_img': img, 'in_bbox': bbox, 'in_label': label, 'in_weight_seed': np.int64(WEIGHT_SEED + (kind == 'keypoint')), 'in_np_seed': np.int64(NP_SEED),
        'anchor': f32(anchor), 'rois': f32(rois), 'levels': np.asarray(levels, np.int32),
        'sample_roi': f32(sample_roi), 'sample_levels': np.asarray(sample_levels, np.int32), 'gt_roi_loc': f32(gt_roi_loc),
        'gt_roi_label': np.asarray(gt_roi_label, np.int32), 'gt_roi_mask': np.asarray(gt_roi_mask, np.int8),
        'gt_rpn_loc': f32(gt_rpn_loc), 'gt_rpn_label': np.asarray(gt_rpn_label, np.int32),
        'indices_and_rois': f32(indices_and_rois),
        'p6': f32(feats[4]), 'p5': f32(feats[3]), 'p4_sub': f32(feats[2][:, ::4]), 'p3_sub': f32(feats[1][:, ::8, ::2, ::2]),
        'p2_sub': f32(feats[0][:, ::8, ::4, ::4]),
        'rpn_locs': f32(rpn_locs), 'rpn_scores': f32(rpn_scores),
        'roi_cls_locs': f32(roi_cls_locs), 'roi_scores': f32(roi_scores),
    }
    if kind == 'mask':
        out.update({'in_mask': np.packbits(mask, axis=-1), 'in_mask_shape': np.array(mask.shape),
                    'roi_mask_pos': f32(np.asarray(roi_cls_mask)[np.arange(n_pos), np.asarray(gt_roi_label)[:n_pos] - 1]),
                    'roi_mask_sub': f32(np.asarray(roi_cls_mask)[:, ::16, ::2, ::2])})
    else:
        assert roi_cls_mask.shape[1:] == (K, 56, 56) and gt_roi_mask.shape == (n_pos, K) and (np.asarray(gt_roi_mask) >= 0).sum() >= 20
        out.update({'in_keypoints': kps, 'roi_mask_sub': f32(np.asarray(roi_cls_mask)[:n_pos, :, ::4, ::4]),
                    'roi_mask_at_label': f32(np.asarray(roi_cls_mask)[:n_pos].reshape(n_pos, K, -1)[
                        np.arange(n_pos)[:, None], np.arange(K)[None, :], np.maximum(np.asarray(gt_roi_mask), 0)])})
        out['gt_roi_mask'] = np.asarray(gt_roi_mask, np.int32)
    for k, v in mc.REPORTED.items():
        out['loss_' + k] = np.float64(v)
    path = os.path.join(HERE, 'step_reference.npz' if kind == 'mask' else 'step_keypoint_reference.npz')
    np.savez_compressed(path, **out)
    print(os.path.basename(path) + ' %.2f MB' % (os.path.getsize(path) / 2 ** 20), 'rois', rois.shape, 'samples', sample_roi.shape, 'positives', n_pos,
          'rpn positives', int((gt_rpn_label == 1).sum()), 'levels', np.bincount(np.asarray(sample_levels, np.int64), minlength=5))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'mask')
