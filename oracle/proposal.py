"""Oracle: RPN proposal generation (TEST INFRASTRUCTURE - see oracle/__init__.py).

Restates ChainerCV ``ProposalCreator.__call__`` (third-party, not on disk) as called at
model/rpn/multilevel_region_proposal_network.py:156-161.  The in-tree mirror of that
algorithm is utils/proposal_creator.py:108-169 (dead code in the reference but the only
on-disk statement of the steps) -- followed line by line below, minus ``level_indices``.
Pinning: the CONTROL FLOW is pinned by tests/golden/pc_reference.npz (that in-tree class
executed in the build container, tests/test_oracle_pins.py); the third-party arithmetic it
calls (``loc2bbox``, ``non_maximum_suppression``) is this oracle's own restatement -
"parity unpinned" for those two.
"""
import numpy as np

from .boxes import loc2bbox, nms, argsort_desc_pinned, map_rois_to_fpn_levels

F = np.float32


class ProposalCreator(object):
    def __init__(self, nms_thresh=0.7, n_train_pre_nms=12000, n_train_post_nms=2000,
                 n_test_pre_nms=6000, n_test_post_nms=300, min_size=16):
        # defaults: utils/proposal_creator.py:50-58
        self.nms_thresh = nms_thresh
        self.n_train_pre_nms = n_train_pre_nms
        self.n_train_post_nms = n_train_post_nms
        self.n_test_pre_nms = n_test_pre_nms
        self.n_test_post_nms = n_test_post_nms
        self.min_size = min_size

    def __call__(self, loc, score, anchor, img_size, scale=1., train=True, return_debug=False):
        n_pre = self.n_train_pre_nms if train else self.n_test_pre_nms      # :108-113
        n_post = self.n_train_post_nms if train else self.n_test_post_nms
        roi = loc2bbox(anchor, loc)                                          # :125
        roi[:, 0::2] = np.clip(roi[:, 0::2], 0, img_size[0])                 # :128-131
        roi[:, 1::2] = np.clip(roi[:, 1::2], 0, img_size[1])
        min_size = F(self.min_size * scale)                                  # :134
        hs = roi[:, 2] - roi[:, 0]
        ws = roi[:, 3] - roi[:, 1]
        keep = np.where((hs >= min_size) & (ws >= min_size))[0]              # :137
        roi = roi[keep, :]
        score = np.asarray(score, F)[keep]
        order = argsort_desc_pinned(score.ravel())                           # :144 (tie pin)
        if n_pre > 0:
            order = order[:n_pre]
        roi = roi[order, :]
        keep2 = nms(roi, self.nms_thresh)                                    # :153-161
        if n_post > 0:
            keep2 = keep2[:n_post]
        out = roi[keep2]
        if return_debug:
            return out, dict(anchor_index=keep[order][keep2], pre_nms_index=keep[order],
                             nms_keep=keep2)
        return out


def rpn_proposals(locs, fg_scores, anchors, img_size, scale=1., train=True, creator=None):
    """model/rpn/multilevel_region_proposal_network.py:154-166: per-image proposals,
    concatenation, batch indices and FPN levels.  locs (N,A,4), fg_scores (N,A)."""
    creator = creator or ProposalCreator()
    rois, idx = [], []
    for i in range(locs.shape[0]):
        r = creator(locs[i], fg_scores[i], anchors, img_size, scale=scale, train=train)
        rois.append(r)
        idx.append(i * np.ones((len(r),), np.int32))
    rois = np.concatenate(rois, axis=0)
    return rois, np.concatenate(idx, axis=0), map_rois_to_fpn_levels(rois)
