"""Shared helpers for tests (synthetic inputs of BASELINE.md section 3)."""
import numpy as np


def config2_inputs(R=512, C=256, H=200, W=272, PH=7, PW=7):
    """BASELINE.json configs[1]: ROIAlign microbench inputs."""
    x = np.random.RandomState(0).standard_normal((1, C, H, W)).astype(np.float32)
    rs = np.random.RandomState(1)
    h = np.exp(rs.uniform(np.log(16), np.log(448), R))
    w = np.exp(rs.uniform(np.log(16), np.log(448), R))
    cy = rs.uniform(0, 4 * H, R)
    cx = rs.uniform(0, 4 * W, R)
    yx = np.stack([np.zeros(R), np.clip(cy - h / 2, 0, 4 * H), np.clip(cx - w / 2, 0, 4 * W),
                   np.clip(cy + h / 2, 0, 4 * H), np.clip(cx + w / 2, 0, 4 * W)], 1).astype(np.float32)
    gy = np.random.RandomState(2).standard_normal((R, C, PH, PW)).astype(np.float32)
    return x, yx, gy          # rois in the reference's (idx,y1,x1,y2,x2) order


def rand_rois_xy(rs, R, N, H, W, scale, lo=2.0):
    h = np.exp(rs.uniform(np.log(lo), np.log(H / scale * 1.2), R))
    w = np.exp(rs.uniform(np.log(lo), np.log(W / scale * 1.2), R))
    cy = rs.uniform(-0.1 * H / scale, 1.1 * H / scale, R)
    cx = rs.uniform(-0.1 * W / scale, 1.1 * W / scale, R)
    idx = rs.randint(0, N, R)
    return np.stack([idx, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
