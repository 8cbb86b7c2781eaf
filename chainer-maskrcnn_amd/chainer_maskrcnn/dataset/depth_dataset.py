"""`--dataset depth` of train_keypoints.py (:86, :103-109): depth frames with 20 body keypoints stored one example per .npz
(arrays `depth` (H, W) and `keypoints` (K, 2|3) as (x, y[, confidence])), listed one relative path per line in a text file.

Counterpart of the reference's chainer_maskrcnn/dataset/depth_dataset.py:7-61 and utils/depth_transformer.py:4-10, NumPy only:
an example is (img (3, H, W) float32, bbox (1, 4) float32, keypoints (1, K, 3)) in the layout COCOKeypointsLoader yields, so the same
KeypointTransform / collate / BatchLoader path serves both datasets.
"""
import os

import numpy as np


class DepthDataset(object):
    n_keypoints = 20

    def __init__(self, path, root='.'):
        with open(path, 'r') as f:
            self.data = [line.strip() for line in f.readlines()]
        self.root = root

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        return self.get_example(i)

    def get_example(self, index):
        if not 0 <= index < len(self.data):
            raise IndexError('index is out of bounds.')
        with np.load(os.path.join(self.root, self.data[index])) as f:
            depth, kp = f['depth'], np.array(f['keypoints'], dtype=np.float64)
        hi = np.array(depth.shape, np.float64) - 1                 # (h - 1, w - 1): the clip bounds of depth_dataset.py:26-29, in that order
        kp[:, :2] = np.clip(kp[:, :2], 0, hi)
        if kp.shape[1] == 2:                                       # no confidence column: every joint labelled and visible
            kp = np.concatenate([kp, np.full((len(kp), 1), 2.0)], axis=1)
        else:
            kp[:, 2] = (kp[:, 2] > 0.2) * 2
        assert kp.shape[1] == 3
        lo_corner = np.clip(kp[:, :2].min(axis=0) - [10, 10], 0, hi)          # :43-46 (margins 10, 10 / 0, 10)
        hi_corner = np.clip(kp[:, :2].max(axis=0) + [0, 10], 0, hi)
        bbox = np.concatenate([lo_corner, hi_corner]).reshape(1, 4)
        kp[:, :2] = kp[:, [1, 0]]                                  # :49: stored (y, x) so that the keypoint Transform's swap lands right
        img = (depth.astype(np.float32) - 1000) / 3000 * 255       # :57: roughly [0, 255]; prepare() divides by 255 later
        return np.stack([img, img, img]), bbox, kp[None]


class DepthTransformer(object):
    """utils/depth_transformer.py:4-10: one random offset in [-15, 15) added to the whole frame.  rs: a RandomState (the reference
    draws from the global np.random)."""

    def __init__(self, rs=None):
        self.rs = rs if rs is not None else np.random

    def __call__(self, in_data):
        x, bbox, keypoint = in_data
        x = x + (self.rs.rand(1).astype(np.float32) - 0.5) * 30
        return x, bbox, keypoint
