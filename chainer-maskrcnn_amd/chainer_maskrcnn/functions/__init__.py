"""Operator surface of the reference's ``chainer_maskrcnn.functions`` package plus the two Chainer loss functions a
user-written ``mask_loss_fun`` calls (``chainer.functions.sigmoid_cross_entropy`` / ``softmax_cross_entropy`` in
train.py:57 and train_keypoints.py:27), HIP-backed (see ``loss.py``)."""
from .loss import sigmoid_cross_entropy, softmax_cross_entropy  # noqa: F401
