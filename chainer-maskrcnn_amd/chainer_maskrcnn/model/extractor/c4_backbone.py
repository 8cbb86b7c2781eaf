"""ResNet-50 C4 backbone (conv1 .. res4) on gfx950 kernels - legacy variant.

Mirror of chainer_maskrcnn/model/extractor/c4_backbone.py:7-26: Chainer's ``ResNet50Layers`` without res5 / fc6, the
ORIGINAL stem pooling ``F.max_pooling_2d(ksize=3, stride=2)`` (cover_all; the FPN extractor uses 2x2 instead), BatchNorm
layers with ``disable_update()`` (:13-15: gamma / beta are not trained; batch statistics are still used in training mode),
returns the one-element tuple ``(res4,)`` (:26).  Forward only: the reference's train path for this backbone (the legacy
MaskRCNNTrainChain) is broken upstream.
"""
from chainer_maskrcnn.nn.core import Conv, BatchNorm, Bottleneck, ParamStore
from chainer_maskrcnn._hip import ops


class C4Backbone(object):
    feat_strides = [16]
    spatial_scales = [1. / 16]
    STAGES = (('res2', 3, 64, 64, 256, 1), ('res3', 4, 256, 128, 512, 2), ('res4', 6, 512, 256, 1024, 2))

    def __init__(self, pretrained_model=None, ps=None, prefix='extractor', stages=None, width_div=1):
        """``pretrained_model``: accepted like the reference's ('auto' downloads there; weights are loaded here through
        utils/chainer_npz.load_npz).  ``stages`` / ``width_div`` shrink the network for tests."""
        self.ps = ps if ps is not None else ParamStore()
        self._own_ps = ps is None
        p = prefix + '/'
        d = width_div
        self.conv1 = Conv(self.ps, p + 'conv1', 3, 64 // d, 7, 2, 3, bias=True, in_backbone=True)
        self.bn1 = BatchNorm(self.ps, p + 'bn1', 64 // d)
        self.stages = []
        for si, (name, n, cin, mid, cout, stride) in enumerate(self.STAGES):
            n = n if stages is None else stages[si]
            blocks = [Bottleneck(self.ps, p + '%s/a' % name, cin // d, mid // d, cout // d, stride, True)]
            for i in range(1, n):
                blocks.append(Bottleneck(self.ps, p + '%s/b%d' % (name, i), cout // d, mid // d, cout // d, 1, False))
            self.stages.append(blocks)
        self.out_channels = 1024 // d
        self.frozen_bn = True           # disable_update(): the optimizer must skip gamma / beta of this backbone

    def __call__(self, x):
        """x (N,H,W,4) NHWC image (zero 4th channel) -> (res4,) with res4 (N, H/16, W/16, 1024) NHWC."""
        h, _ = self.conv1.fwd(x)
        h, _ = self.bn1.fwd(h, relu=True)
        h = ops.maxpool3x3s2_fwd(h)
        for blocks in self.stages:
            for b in blocks:
                h, _ = b.fwd(h)
        return h,
