"""Shared helpers for tests (synthetic inputs of BASELINE.md section 3)."""
import numpy as np


from chainer_maskrcnn.utils.synthetic import config2_inputs  # noqa: F401  (BASELINE.json configs[1] inputs live in the package)


def rand_rois_xy(rs, R, N, H, W, scale, lo=2.0):
    h = np.exp(rs.uniform(np.log(lo), np.log(H / scale * 1.2), R))
    w = np.exp(rs.uniform(np.log(lo), np.log(W / scale * 1.2), R))
    cy = rs.uniform(-0.1 * H / scale, 1.1 * H / scale, R)
    cx = rs.uniform(-0.1 * W / scale, 1.1 * W / scale, R)
    idx = rs.randint(0, N, R)
    return np.stack([idx, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
