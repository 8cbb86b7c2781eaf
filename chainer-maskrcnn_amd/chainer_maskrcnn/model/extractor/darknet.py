"""Tiny Darknet-style extractor on gfx950 kernels - legacy variant.

Mirror of chainer_maskrcnn/model/extractor/darknet.py:6-60: five ``ConvBatch`` units (3x3 convolution with bias ->
training-mode BatchNorm -> activation; 16, 32, 64, 128, 256 channels) with a 2x2 / stride-2 max pooling (cover_all) after
each of the first four; one output level of stride 16 (class attributes :21-29).  Forward only.
"""
from chainer_maskrcnn.nn.core import Conv, BatchNorm, ParamStore
from chainer_maskrcnn._hip import ops


class ConvBatch(object):
    def __init__(self, ps, name, in_channels, out_channels, ksize, stride, pad, activation='relu'):
        if activation not in ('relu', None):
            raise ValueError('ConvBatch: activation must be "relu" or None on this path')
        self.c = Conv(ps, name + '/c', in_channels, out_channels, ksize, stride, pad, bias=True, in_backbone=True)
        self.bn = BatchNorm(ps, name + '/bn', self.c.cout_p)
        self.activation = activation

    def __call__(self, x):
        h, _ = self.c.fwd(x)
        y, _ = self.bn.fwd(h, relu=self.activation == 'relu')
        return y


class Darknet(object):
    feat_strides = [16]
    spatial_scales = list(map(lambda x: 1. / x, feat_strides))
    anchor_base = 16
    anchor_sizes = [64]
    anchor_scales = list(map(lambda x: x / 16., anchor_sizes))

    def __init__(self, activation='relu', ps=None, prefix='extractor'):
        self.ps = ps if ps is not None else ParamStore()
        cin = 3
        self.convs = []
        for i, cout in enumerate((16, 32, 64, 128, 256)):
            self.convs.append(ConvBatch(self.ps, '%s/conv%d' % (prefix, i + 1), cin, cout, 3, 1, 1, activation))
            cin = cout
        self.out_channels = 256
        self.anchor_scales = list(map(lambda x: x / float(self.anchor_base), self.anchor_sizes))

    def __call__(self, x):
        """x (N,H,W,4) NHWC image -> (h,) with h (N, ceil(H/16), ceil(W/16), 256)."""
        h = x
        for i, cb in enumerate(self.convs):
            h = cb(h)
            if i < 4:
                h = ops.maxpool2x2_fwd(h)
        return h,
