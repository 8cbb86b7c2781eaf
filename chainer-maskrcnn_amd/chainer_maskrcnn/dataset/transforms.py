"""Training transforms of the reference (train.py:21-37 ``Transform``, train_keypoints.py:51-68) on the host.

The reference resizes through ChainerCV / OpenCV (``chainercv.transforms.resize`` = ``cv2.resize`` INTER_LINEAR on
float32 when cv2 is importable - train.py:8 imports it - and ``cv2.resize(..., INTER_NEAREST)`` for the masks).
Neither package exists on the target machine, so the two interpolations are restated here in NumPy with OpenCV's
coordinate rules (parity unpinned; the float path is checked against the independent oracle restatement):
  INTER_LINEAR  fx = (dx + 0.5) * (src/dst) - 0.5, sx = floor(fx), clamped to the edge; horizontal pass then vertical
                pass, float32 coefficients
  INTER_NEAREST sx = min(floor(dx * (src/dst)), src - 1)
"""
import numpy as np

F = np.float32


def _linear_taps(dst, src):
    scale = 1.0 / (float(dst) / float(src))            # cv2: inv_scale = dsize/ssize (double); scale = 1/inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(F)).astype(F)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0, src - 1
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, (F(1) - f).astype(F), f


def resize_linear(img, out_hw):
    """img (C,H,W) float32 -> (C,oh,ow) float32, cv2.resize(..., INTER_LINEAR) per channel."""
    img = np.asarray(img, F)
    C, H, W = img.shape
    oh, ow = out_hw
    if (oh, ow) == (H, W):
        return img.copy()
    x0, x1, a0, a1 = _linear_taps(ow, W)
    y0, y1, b0, b1 = _linear_taps(oh, H)
    rows = img[:, :, x0] * a0 + img[:, :, x1] * a1                  # horizontal pass on every source row
    return (rows[:, y0, :] * b0[:, None] + rows[:, y1, :] * b1[:, None]).astype(F)


def resize_nearest(mask, out_hw):
    """mask (H,W) any dtype -> (oh,ow), cv2.resize(mask, (ow,oh), interpolation=cv2.INTER_NEAREST)."""
    H, W = mask.shape
    oh, ow = out_hw
    sx = np.minimum(np.floor(np.arange(ow) * (1.0 / (float(ow) / W))).astype(np.int64), W - 1)
    sy = np.minimum(np.floor(np.arange(oh) * (1.0 / (float(oh) / H))).astype(np.int64), H - 1)
    return np.ascontiguousarray(mask[sy][:, sx])


def resize_bbox(bbox, in_size, out_size):
    """chainercv.transforms.resize_bbox: (y1,x1,y2,x2) scaled by out/in per axis."""
    bbox = np.array(bbox, dtype=np.float32).reshape(-1, 4)
    ys, xs = float(out_size[0]) / in_size[0], float(out_size[1]) / in_size[1]
    bbox[:, 0] *= ys
    bbox[:, 2] *= ys
    bbox[:, 1] *= xs
    bbox[:, 3] *= xs
    return bbox


def prepare(img, min_size=600, max_size=1000):
    """MaskRCNN.prepare on the host (maskrcnn.py:261-276): short side -> min_size unless the long side would exceed
    max_size; values scaled to [0,1]; no mean subtraction."""
    _, H, W = img.shape
    scale = min_size / min(H, W)
    if scale * max(H, W) > max_size:
        scale = max_size / max(H, W)
    return resize_linear(img, (int(H * scale), int(W * scale))) / F(255)


class Transform(object):
    """train.py:21-37.  in: (img, bbox, label, masks list) -> (img, bbox, label, masks (G,oH,oW) uint8, scale)."""

    def __init__(self, faster_rcnn):
        self.min_size, self.max_size = faster_rcnn.min_size, faster_rcnn.max_size

    def __call__(self, in_data):
        img, bbox, label, label_img = in_data
        _, H, W = img.shape
        img = prepare(img, self.min_size, self.max_size)
        _, o_H, o_W = img.shape
        scale = o_H / H
        bbox = resize_bbox(bbox, (H, W), (o_H, o_W))
        bbox[:, 2:] = np.maximum(bbox[:, 2:], bbox[:, 2:] + 1)           # == += 1 (train.py:32)
        masks = [resize_nearest(np.asarray(im), (o_H, o_W)) for im in label_img]
        masks = np.stack(masks).astype(np.uint8) if masks else np.zeros((0, o_H, o_W), np.uint8)
        return img, bbox, np.asarray(label, np.int32), masks, scale


class KeypointTransform(object):
    """train_keypoints.py:51-68.  in: (img, bbox, keypoints (G,17,(x,y,v))) -> (img, bbox, label = 0, kp (G,17,(y,x,v)), scale)."""

    def __init__(self, faster_rcnn):
        self.min_size, self.max_size = faster_rcnn.min_size, faster_rcnn.max_size

    def __call__(self, in_data):
        img, bbox, keypoints = in_data
        _, H, W = img.shape
        img = prepare(img, self.min_size, self.max_size)
        _, o_H, o_W = img.shape
        scale = o_H / H
        bbox = resize_bbox(bbox, (H, W), (o_H, o_W))
        label = np.zeros(bbox.shape[0], dtype=np.int32)
        keypoints = keypoints.astype(np.float32)
        kp = keypoints[:, :, [1, 0]]
        kp = np.concatenate([kp * scale, keypoints[:, :, 2, None]], axis=2)
        return img, bbox, label, kp, scale


class RawTransform(object):
    """Host half of the device-side Transform: everything except the two resizes, which run on the GPU
    (mrcnn_image_resize_u8_f32 / mrcnn_mask_resize_nearest_u8) on the raw uint8 data.  Returns
    (img_u8 (H,W,3), bbox, label, masks_u8 (G,H,W) | keypoints, scale, (oH,oW))."""

    def __init__(self, faster_rcnn, keypoints=False):
        self.min_size, self.max_size, self.keypoints = faster_rcnn.min_size, faster_rcnn.max_size, keypoints

    def out_size(self, H, W):
        scale = self.min_size / min(H, W)
        if scale * max(H, W) > self.max_size:
            scale = self.max_size / max(H, W)
        return int(H * scale), int(W * scale)

    def __call__(self, in_data):
        img = in_data[0]
        _, H, W = img.shape
        o_H, o_W = self.out_size(H, W)
        scale = o_H / H
        img_u8 = np.ascontiguousarray(img.transpose(1, 2, 0)).astype(np.uint8)      # decoded JPEGs are integer-valued
        bbox = resize_bbox(in_data[1], (H, W), (o_H, o_W))
        if self.keypoints:
            keypoints = in_data[2].astype(np.float32)
            kp = np.concatenate([keypoints[:, :, [1, 0]] * scale, keypoints[:, :, 2, None]], axis=2)
            return img_u8, bbox, np.zeros(bbox.shape[0], dtype=np.int32), kp, scale, (o_H, o_W)
        bbox[:, 2:] = np.maximum(bbox[:, 2:], bbox[:, 2:] + 1)
        masks = np.stack([np.asarray(m, np.uint8) for m in in_data[3]]) if len(in_data[3]) else np.zeros((0, H, W), np.uint8)
        return img_u8, bbox, np.asarray(in_data[2], np.int32), masks, scale, (o_H, o_W)
