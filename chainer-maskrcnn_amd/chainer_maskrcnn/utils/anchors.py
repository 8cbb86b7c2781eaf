"""Anchor enumeration (host, computed once per image size and cached on the device).

Follows ChainerCV ``generate_anchor_base`` / ``_enumerate_shifted_anchor`` as called at
chainer_maskrcnn/model/rpn/multilevel_region_proposal_network.py:70-71,128-129:
ratio-major anchor bases of one scale per level centred at (8, 8); shifts row-major (y outer, x
inner); anchors position-major, base-minor; float32.
"""
import numpy as np


def generate_anchor_base(base_size=16, ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32)):
    py = px = base_size / 2.
    out = np.zeros((len(ratios) * len(anchor_scales), 4), dtype=np.float32)
    for i, r in enumerate(ratios):
        for j, s in enumerate(anchor_scales):
            h = base_size * s * np.sqrt(r)
            w = base_size * s * np.sqrt(1. / r)
            out[i * len(anchor_scales) + j] = (py - h / 2., px - w / 2., py + h / 2., px + w / 2.)
    return out


def enumerate_shifted_anchor(anchor_base, feat_stride, height, width):
    sy = np.arange(0, height * feat_stride, feat_stride)
    sx = np.arange(0, width * feat_stride, feat_stride)
    sx, sy = np.meshgrid(sx, sy)
    shift = np.stack((sy.ravel(), sx.ravel(), sy.ravel(), sx.ravel()), axis=1)
    A, K = anchor_base.shape[0], shift.shape[0]
    anchor = anchor_base.reshape((1, A, 4)) + shift.reshape((1, K, 4)).transpose((1, 0, 2))
    return anchor.reshape((K * A, 4)).astype(np.float32)
