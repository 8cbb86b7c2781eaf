"""Synthetic COCO-shaped batches (SURVEY.md section 8d / BASELINE.md section 3): the tensor contract of
``Transform`` + ``concat_examples`` (train.py:21-37) without a dataset.

image  RandomState(seed).rand(N,3,H,W) float32 in [0,1] (like ``prepare``: img/255, maskrcnn.py:274)
bbox   G boxes per image, sides log-uniform in [32, 512] px clipped to the image, (y1,x1,y2,x2)
label  randint(0, n_fg_class)
mask   filled axis-aligned ellipse inside each box, uint8 {0,1} at image resolution
keypoints (optional) K points uniform inside the box, visibility 2: (y, x, v)
"""
import numpy as np


def make_batch(seed, N, H, W, G=8, n_fg_class=80, n_keypoints=None):
    rs = np.random.RandomState(seed)
    imgs = rs.rand(N, 3, H, W).astype(np.float32)
    bbox = np.zeros((N, G, 4), np.float32)
    label = rs.randint(0, n_fg_class, (N, G)).astype(np.int32)
    masks = np.zeros((N, G, H, W), np.uint8)
    kps = np.zeros((N, G, n_keypoints or 1, 3), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(N):
        for g in range(G):
            h = min(np.exp(rs.uniform(np.log(32), np.log(512))), H - 1)
            w = min(np.exp(rs.uniform(np.log(32), np.log(512))), W - 1)
            y0 = rs.uniform(0, H - h)
            x0 = rs.uniform(0, W - w)
            bbox[i, g] = (y0, x0, y0 + h, x0 + w)
            cy, cx = y0 + h / 2, x0 + w / 2
            masks[i, g] = (((yy + 0.5 - cy) / (h / 2)) ** 2 + ((xx + 0.5 - cx) / (w / 2)) ** 2 <= 1.0)
            if n_keypoints:
                kps[i, g, :, 0] = rs.uniform(y0, y0 + h, n_keypoints)
                kps[i, g, :, 1] = rs.uniform(x0, x0 + w, n_keypoints)
                kps[i, g, :, 2] = 2
    out = dict(imgs=imgs, bboxes=bbox, labels=label, masks=masks)
    if n_keypoints:
        out['keypoints'] = kps
    return out


def config2_inputs(R=512, C=256, H=200, W=272, PH=7, PW=7):
    """BASELINE.json configs[1]: the ROIAlign microbench inputs: a (1,C,H,W) map at stride 4, R RoIs with
    log-uniform sides in [16, 448] px at uniform centres, clipped to the image, in the reference's (idx,y1,x1,y2,x2) order,
    and an upstream gradient for the pooled output."""
    x = np.random.RandomState(0).standard_normal((1, C, H, W)).astype(np.float32)
    rs = np.random.RandomState(1)
    h = np.exp(rs.uniform(np.log(16), np.log(448), R))
    w = np.exp(rs.uniform(np.log(16), np.log(448), R))
    cy = rs.uniform(0, 4 * H, R)
    cx = rs.uniform(0, 4 * W, R)
    yx = np.stack([np.zeros(R), np.clip(cy - h / 2, 0, 4 * H), np.clip(cx - w / 2, 0, 4 * W),
                   np.clip(cy + h / 2, 0, 4 * H), np.clip(cx + w / 2, 0, 4 * W)], 1).astype(np.float32)
    gy = np.random.RandomState(2).standard_normal((R, C, PH, PW)).astype(np.float32)
    return x, yx, gy
