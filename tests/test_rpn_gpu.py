"""GPU parity tests: device proposal path (rpn.hip) against the NumPy oracle (oracle/boxes.py,
oracle/proposal.py).  Integer outputs (sort order, NMS keep lists, FPN levels) are bit-exact given
identical float inputs; decoded boxes involve exp() and are compared to 2 ulp-level tolerance, so
the end-to-end bit-exact case uses dh = dw = 0 (exp(0) == 1 on both sides)."""
import os

import numpy as np
import pytest
import torch

from oracle import boxes as ob
from oracle.proposal import ProposalCreator

pytestmark = pytest.mark.gpu

from chainer_maskrcnn._hip import ops  # noqa: E402

DEV = 'cuda:0'
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _rand_boxes(rs, n, size=400.0):
    c = rs.uniform(0, size, (n, 2))
    hw = np.exp(rs.uniform(np.log(8), np.log(200), (n, 2)))
    return np.concatenate([c - hw / 2, c + hw / 2], 1).astype(np.float32)


# 5000 / 12000 boxes = 79 / 188 chunks of 64: the reduction crosses several 48-chunk windows, with light (0.7) and heavy
# (0.3) suppression
@pytest.mark.parametrize('n,thresh', [(1, 0.7), (63, 0.7), (64, 0.5), (65, 0.3), (1000, 0.7), (5000, 0.7), (5000, 0.3), (12000, 0.5)])
def test_nms_keep_list_bit_exact(n, thresh):
    rs = np.random.RandomState(n)
    b = _rand_boxes(rs, n)
    if n > 100:      # duplicates and threshold-equal pairs
        b[10] = b[3]
        b[11] = [0, 0, 10, 10]
        b[12] = [0, 0, 10, 7]        # IoU exactly 0.7 with box 11
    want = ob.nms(b, thresh)
    keep, nk = ops.nms(torch.from_numpy(b).to(DEV), thresh)
    nk = int(nk.item())
    assert nk == len(want)
    np.testing.assert_array_equal(keep.cpu().numpy()[:nk], want)


# The walk works in super-chunks of 256 boxes inside windows of 48 chunks (3072 boxes): sizes on and around those
# boundaries, the largest supported input, sparse boxes (almost nothing suppressed: every chunk keeps 64) and dense ones, and
# max_keep values that stop the walk inside a chunk, at a chunk end and at a super-chunk end.
@pytest.mark.parametrize('n,spread,thresh,max_keep', [(255, 400, 0.5, None), (256, 400, 0.5, None), (257, 400, 0.5, None),
                                                      (3071, 3000, 0.7, None), (3072, 3000, 0.7, None), (3073, 3000, 0.7, None),
                                                      (6200, 20000, 0.7, None), (16384, 4000, 0.6, None), (16384, 60000, 0.7, 2000),
                                                      (9000, 2000, 0.7, 1), (9000, 2000, 0.7, 64), (9000, 2000, 0.7, 65),
                                                      (9000, 2000, 0.7, 256), (9000, 2000, 0.7, 300)])
def test_nms_walk_boundaries(n, spread, thresh, max_keep):
    rs = np.random.RandomState(n + (max_keep or 0))
    b = _rand_boxes(rs, n, float(spread))
    want = ob.nms(b, thresh)
    if max_keep:
        want = want[:max_keep]
    keep, nk = ops.nms(torch.from_numpy(b).to(DEV), thresh, max_keep=max_keep)
    nk = int(nk.item())
    assert nk == len(want)
    np.testing.assert_array_equal(keep.cpu().numpy()[:nk], want)


def test_nms_max_keep_and_empty():
    rs = np.random.RandomState(2)
    b = _rand_boxes(rs, 500)
    want = ob.nms(b, 0.7)[:50]
    keep, nk = ops.nms(torch.from_numpy(b).to(DEV), 0.7, max_keep=50)
    assert int(nk.item()) == 50
    np.testing.assert_array_equal(keep.cpu().numpy(), want)
    keep, nk = ops.nms(torch.zeros((0, 4), device=DEV), 0.7)
    assert int(nk.item()) == 0


def test_fpn_levels_match_reference_golden():
    """tests/golden/levels_reference.npz was produced by the REFERENCE function
    (model/rpn/multilevel_region_proposal_network.py:16-31) in the build container."""
    z = np.load(os.path.join(GOLDEN, 'levels_reference.npz'))
    rois, want = z['rois'], z['levels']
    got = ops.map_rois_to_fpn_levels(torch.from_numpy(np.ascontiguousarray(rois, np.float32)).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, want.astype(np.float32))
    rs = np.random.RandomState(0)
    b = _rand_boxes(rs, 5000, 1000.0)
    np.testing.assert_array_equal(ops.map_rois_to_fpn_levels(torch.from_numpy(b).to(DEV)).cpu().numpy(),
                                  ob.map_rois_to_fpn_levels(b))


def _rpn_case(seed, N, feat_shapes, exact):
    rs = np.random.RandomState(seed)
    anchors = ob.fpn_anchors(feat_shapes)
    A = anchors.shape[0]
    locs = (rs.standard_normal((N, A, 4)) * 0.3).astype(np.float32)
    if exact:
        locs[:, :, 2:] = 0
    scores = rs.standard_normal((N, A, 2)).astype(np.float32)
    scores[:, 5:40, 1] = scores[0, 5, 1]          # tied scores: order must follow the (score desc, index desc) pin
    return anchors, locs, scores


# The top-n_pre selection is a radix select + an in-LDS bitonic sort whose shape depends on n_pre: 8192 keys (8 per thread),
# 1024 (1 per thread, partner through LDS), 4096 (4 per thread), 128 (fewer threads than the block), and - with the larger
# pyramid, A = 23,025 anchors - the training size: 12,000 of them in a 16,384-key sort (16 per thread).
@pytest.mark.parametrize('n_pre,n_post,big', [(12000, 2000, False), (600, 100, False), (3000, 300, False), (100, 20, False),
                                              (12000, 2000, True), (2048, 300, False), (2049, 300, False), (16384, 2000, True)])
def test_proposals_end_to_end_bit_exact(n_pre, n_post, big):
    N = 2
    feat = [(72, 80), (36, 40), (18, 20), (9, 10), (5, 5)] if big else [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)]
    anchors, locs, scores = _rpn_case(1, N, feat, exact=True)
    img_size = (288, 320) if big else (160, 192)
    o = ops.rpn_proposals(torch.from_numpy(locs).to(DEV), torch.from_numpy(scores).to(DEV), torch.from_numpy(anchors).to(DEV),
                          img_size, 16.0, n_pre, n_post, 0.7, debug=True)
    pc = ProposalCreator(n_train_pre_nms=n_pre, n_train_post_nms=n_post)
    for i in range(N):
        want, dbg = pc(locs[i], scores[i, :, 1], anchors, img_size, return_debug=True)
        npre = int(o['n_pre'][i].item())
        assert npre == len(dbg['pre_nms_index'])
        np.testing.assert_array_equal(o['sorted_anchor'].cpu().numpy().reshape(N, -1)[i, :npre], dbg['pre_nms_index'])
        nk = int(o['n_rois'][i].item())
        assert nk == len(want)
        np.testing.assert_array_equal(o['keep'].cpu().numpy().reshape(N, -1)[i, :nk], dbg['nms_keep'])
        got = o['rois'].cpu().numpy().reshape(N, n_post, 4)[i]
        np.testing.assert_array_equal(got[:nk], want)
        assert np.all(got[nk:] == 0)
        idx = o['roi_indices'].cpu().numpy().reshape(N, n_post)[i]
        assert np.all(idx[:nk] == i) and np.all(idx[nk:] == -1)
        np.testing.assert_array_equal(o['levels'].cpu().numpy().reshape(N, n_post)[i, :nk], ob.map_rois_to_fpn_levels(want))


def test_proposals_per_image_size_and_scale_equal_separate_calls():
    """A padded batch of two images of different sizes / resize factors: with per_image (h, w, min_size * scale) every
    image's proposals are bitwise those of a batch-1 call with its own size - and they are checked against the oracle's
    ProposalCreator called the way the reference calls it (rpn/multilevel_region_proposal_network.py:156-164: that
    image's img_size and scale)."""
    N, n_pre, n_post = 2, 3000, 300
    feat = [(40, 48), (20, 24), (10, 12), (5, 6), (3, 3)]
    anchors, locs, scores = _rpn_case(4, N, feat, exact=True)
    sizes = np.array([[160, 192], [117, 150]], np.float32)
    scales = np.array([1.0, 1.75], np.float32)
    per = torch.from_numpy(np.concatenate([sizes, (16.0 * scales)[:, None]], 1).astype(np.float32)).to(DEV)
    dl, ds, da = (torch.from_numpy(v).to(DEV) for v in (locs, scores, anchors))
    o = ops.rpn_proposals(dl, ds, da, (160, 192), 16.0, n_pre, n_post, 0.7, per_image=per)
    for i in range(N):
        one = ops.rpn_proposals(dl[i:i + 1].contiguous(), ds[i:i + 1].contiguous(), da, tuple(sizes[i]), 16.0 * float(scales[i]), n_pre, n_post, 0.7)
        nk = int(one['n_rois'][0].item())
        assert nk == int(o['n_rois'][i].item())
        got = o['rois'].cpu().numpy().reshape(N, n_post, 4)[i]
        np.testing.assert_array_equal(got, one['rois'].cpu().numpy())
        want = ProposalCreator(n_train_pre_nms=n_pre, n_train_post_nms=n_post)(locs[i], scores[i, :, 1], anchors, tuple(sizes[i]), scale=float(scales[i]))
        np.testing.assert_array_equal(got[:nk], want)
        assert got[:nk, 2].max() <= sizes[i, 0] and got[:nk, 3].max() <= sizes[i, 1]
    assert not np.array_equal(o['rois'].cpu().numpy(), ops.rpn_proposals(dl, ds, da, (160, 192), 16.0, n_pre, n_post, 0.7)['rois'].cpu().numpy())


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_proposals_equal_reference_golden(ci):
    """tests/golden/pc_reference.npz: RoIs produced by the REFERENCE's in-tree ProposalCreator (utils/proposal_creator.py)
    executed in the build container.  The device's decode -> clip -> filter -> select -> sort -> NMS chain must reproduce them
    bit for bit (dh = dw = 0 in these cases: no exp() rounding involved)."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pc_reference.npz'))
    g = lambda k: d['c%d_in_%s' % (ci, k)]
    loc, score, anchor = g('loc'), g('score'), g('anchor')
    A = anchor.shape[0]
    scores2 = np.stack([np.zeros(A, np.float32), score], 1)[None]
    n_pre, n_post = int(g('n_pre')), int(g('n_post'))
    o = ops.rpn_proposals(torch.from_numpy(loc[None]).to(DEV), torch.from_numpy(scores2).to(DEV), torch.from_numpy(anchor).to(DEV),
                          tuple(int(v) for v in g('img')), float(g('min_size')) * float(g('scale')), n_pre, n_post, 0.7)
    want = d['c%d_out_roi' % ci]
    nk = int(o['n_rois'][0].item())
    assert nk == len(want)
    np.testing.assert_array_equal(o['rois'].cpu().numpy()[:nk], want)
    np.testing.assert_array_equal(o['levels'].cpu().numpy()[:nk], ob.map_rois_to_fpn_levels(want))


def test_proposals_random_scales_decode_tolerance():
    """exp() differs by <= 2 ulp between NumPy and the device: boxes compared with tolerance."""
    N = 1
    feat = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]
    anchors, locs, scores = _rpn_case(2, N, feat, exact=False)
    img_size = (128, 128)
    o = ops.rpn_proposals(torch.from_numpy(locs).to(DEV), torch.from_numpy(scores).to(DEV), torch.from_numpy(anchors).to(DEV),
                          img_size, 16.0, 3000, 300, 0.7, debug=True)
    want, dbg = ProposalCreator(n_train_pre_nms=3000, n_train_post_nms=300)(locs[0], scores[0, :, 1], anchors, img_size,
                                                                            return_debug=True)
    nk = int(o['n_rois'][0].item())
    got = o['rois'].cpu().numpy()[:nk]
    m = min(nk, len(want), 50)          # the head of the list is stable under ulp-level box differences
    np.testing.assert_allclose(got[:m], want[:m], rtol=1e-5, atol=1e-4)
    assert abs(nk - len(want)) <= 3


def test_pack_unpack_roundtrip():
    rs = np.random.RandomState(5)
    N, H, W, Cp, A = 2, 5, 7, 32, 3
    head = rs.standard_normal((N, H, W, Cp)).astype(np.float32)
    Atot = H * W * A + 11
    locs = torch.zeros((N, Atot, 4), device=DEV)
    scores = torch.zeros((N, Atot, 2), device=DEV)
    ops.rpn_pack(torch.from_numpy(head).to(DEV), A, locs, scores, 11)
    # reference semantics: NCHW conv output -> transpose(0,2,3,1).reshape(n,-1,4)  (multilevel_rpn.py:133-141)
    want_l = head[..., :12].reshape(N, H * W * A, 4)
    want_s = head[..., 12:18].reshape(N, H * W * A, 2)
    np.testing.assert_array_equal(locs.cpu().numpy()[:, 11:], want_l)
    np.testing.assert_array_equal(scores.cpu().numpy()[:, 11:], want_s)
    gh = ops.rpn_unpack_grad(locs, scores, (N, H, W, Cp), A, 11).cpu().numpy()
    np.testing.assert_array_equal(gh[..., :18], head[..., :18])
    assert np.all(gh[..., 18:] == 0)
