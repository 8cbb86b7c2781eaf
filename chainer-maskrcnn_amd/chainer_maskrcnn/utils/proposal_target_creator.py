"""ProposalTargetCreator on the device.

Mirror of chainer_maskrcnn/utils/proposal_target_creator.py:12-137 (constructor :12-24, __call__
:26-137): same arguments and outputs; the NumPy/OpenCV body (5 D2H + 5 H2D copies, <=64 cv2.resize
calls) is replaced by ``mrcnn_proposal_target_f32`` + ``mrcnn_mask_target_u8`` /
``mrcnn_keypoint_target_f32``.  Random subsets: smallest-key selection with keys from
``mrcnn_random_keys_u32`` (``seed`` attribute, advanced per call) or caller-provided ``keys``;
sampled rows come out positives first in ascending candidate order (the reference: random order).
Reference-order mode (``random_state=`` a ``numpy.random.RandomState`` or the ``numpy.random`` module): the host
draws ``choice(n_candidates, size, replace=False)`` for the foreground and then the background set exactly like
:63-78 (one host sync for the two candidate counts) and the device emits the rows in that draw order - outputs are
then row-for-row those of the reference under the same NumPy seed; ``inplace_kp_quirk=True`` additionally reproduces
the reference's in-place mutation of the gt keypoints.
"""
import numpy as np
import torch

from chainer_maskrcnn._hip import ops


class ProposalTargetCreator(object):
    def __init__(self, sizes=[16], n_sample=256, pos_ratio=0.25, pos_iou_thresh=0.5, neg_iou_thresh_hi=0.5,
                 neg_iou_thresh_lo=0.0):
        self.sizes = sizes                      # stored, never used - as in the reference (:19)
        self.n_sample = n_sample
        self.pos_ratio = pos_ratio
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh_hi = neg_iou_thresh_hi
        self.neg_iou_thresh_lo = neg_iou_thresh_lo
        self.seed = 0
        self._state = None           # device-resident seed (created lazily; see set_seed)

    def set_seed(self, seed, device=None):
        self.seed = seed
        self._state = None if device is None else ops.seed_state(seed, device)

    def _keys(self, shape, device):
        if self._state is None or self._state.device != device:
            self._state = ops.seed_state(self.seed, device)
        return ops.random_keys_dev(shape, self._state)

    @property
    def pos_cap(self):
        return int(np.round(self.n_sample * self.pos_ratio))

    def sample_batch(self, rois, roi_levels, n_rois, gt_boxes, gt_labels, n_gt, masks=None, keypoints=None,
                     loc_normalize_mean=(0., 0., 0., 0.), loc_normalize_std=(0.1, 0.1, 0.2, 0.2), mask_size=14,
                     keys=None, mask_rows='positives', random_state=None, inplace_kp_quirk=False):
        """Batched, sync-free form used by the train chain.  rois (N*roi_cap,4) padded, gt_boxes (N,G,4),
        gt_labels (N,G) i32, masks (N,G,H,W) u8 or keypoints (N,G,K,3) f32.  Returns the dict of
        ops.proposal_target plus 'gt_roi_mask' ((N*rows, S, S) or (N*rows, K) int32, -1 = unused) where
        rows = pos_cap ('positives') or n_sample ('all')."""
        N, G = gt_labels.shape
        roi_cap = rois.shape[0] // N
        if keys is None:
            keys = self._keys((N, roi_cap + G), rois.device)
        args = (rois, roi_levels, n_rois, gt_boxes, gt_labels, n_gt, keys, self.n_sample, self.pos_ratio,
                self.pos_iou_thresh, self.neg_iou_thresh_hi, self.neg_iou_thresh_lo, loc_normalize_mean, loc_normalize_std)
        o = ops.proposal_target(*args)
        if random_state is not None:
            # reference draw order (:63-78): pos choice first, then neg choice, image by image; the candidate counts come
            # from a first pass of the kernel (host sync - this mode exists for parity with the reference's RNG stream)
            n_cand = o['n_cand'].cpu().numpy()
            order = np.zeros((2, N, self.n_sample), np.int32)
            for i in range(N):
                n_p, n_n = int(n_cand[i, 0]), int(n_cand[i, 1])
                k_p = int(min(self.pos_cap, n_p))
                if n_p > 0:
                    order[0, i, :k_p] = random_state.choice(n_p, size=k_p, replace=False)
                k_n = int(min(self.n_sample - k_p, n_n))
                if n_n > 0:
                    order[1, i, :k_n] = random_state.choice(n_n, size=k_n, replace=False)
            od = torch.from_numpy(order).to(rois.device)
            o = ops.proposal_target(*args, pos_order=od[0].contiguous(), neg_order=od[1].contiguous())
        rows = self.pos_cap if mask_rows == 'positives' else self.n_sample
        if masks is not None:
            o['gt_roi_mask'] = ops.mask_target(masks, o['sample_roi'], o['gt_assign'], o['n_pos'], self.n_sample, rows,
                                               mask_size)
        elif keypoints is not None:
            o['gt_roi_mask'] = ops.keypoint_target(keypoints, o['sample_roi'], o['gt_assign'], o['n_pos'], self.n_sample,
                                                   rows, mask_size, inplace_quirk=inplace_kp_quirk)
        o['mask_rows'] = rows
        return o

    def __call__(self, roi, bbox, label, mask, levels, loc_normalize_mean=(0., 0., 0., 0.),
                 loc_normalize_std=(0.1, 0.1, 0.2, 0.2), mask_size=14, binary_mask=True, keys=None, random_state=None,
                 inplace_kp_quirk=False):
        """Reference signature (single image): returns (sample_roi, sample_levels, gt_roi_loc, gt_roi_label,
        gt_roi_mask) with exact sizes (one host sync for the counts).  random_state: reference-order mode."""
        dev = roi.device
        R, G = roi.shape[0], bbox.shape[0]
        i32 = torch.int32
        o = self.sample_batch(roi.contiguous(), levels.to(torch.float32).contiguous(),
                              torch.tensor([R], dtype=i32, device=dev), bbox[None].contiguous(),
                              label[None].to(i32).contiguous(), torch.tensor([G], dtype=i32, device=dev),
                              masks=mask[None].contiguous() if binary_mask else None,
                              keypoints=None if binary_mask else mask[None].contiguous(),
                              loc_normalize_mean=loc_normalize_mean, loc_normalize_std=loc_normalize_std,
                              mask_size=mask_size, keys=keys, random_state=random_state, inplace_kp_quirk=inplace_kp_quirk)
        S, n_pos = int(o['n_sampled'][0].item()), int(o['n_pos'][0].item())
        return (o['sample_roi'][:S], o['sample_levels'][:S].to(torch.float32), o['gt_roi_loc'][:S],
                o['gt_roi_label'][:S], o['gt_roi_mask'][:n_pos])
