"""ResNet-50 + FPN feature extractor on gfx950 kernels.

Mirror of chainer_maskrcnn/model/extractor/feature_pyramid_network.py:9-71 (class attributes :9-16,
layers :22-40, forward :46-71) with the bottom-up network of Chainer's ``ResNet50Layers`` (SURVEY.md
Appendix A-8) spelled out.  Quirks kept on purpose (SURVEY.md Appendix B-1,2,4): 2x2/2 cover_all
max-pool after conv1, BatchNorm in training mode, no 3x3 smoothing on P5, P6 = 1x1 stride-2 conv of P5.
Tensors are NHWC; the image enters as (N,H,W,4) with a zero 4th channel.
"""
from chainer_maskrcnn.nn.core import Conv, BatchNorm, Bottleneck, ParamStore
from chainer_maskrcnn._hip import ops, nn as hnn

# The ReLU backward of a bottleneck's output is applied by the kernels that WRITE that output's gradient (data-gradient epilogues,
# lattice scatter) instead of by the block's own bn3 backward: bn3's two backward kernels read two streams instead of three and the
# separate shortcut gradient is not written (same numbers bit for bit; False = the round-2 data flow, kept for A/B).
MASK_IN_PRODUCER = True
# Stream placement of the lateral 1x1 convolutions (OFF by default - built when the round's GPU minutes were spent,
# bit-identity and an A/B are in tests/test_step_gpu.py / tools/ab_step.py, the full suite has not run with it):
#   forward:  lat_p2 / lat_p3 / lat_p4 need c2 / c3 / c4 only - enqueued on the (idle) weight-gradient stream as soon as their ResNet stage
#             is done, they run beside the later stages instead of in the top-down chain;
#   backward: their data gradients g_c2 .. g_c4 are not needed until the matching ResNet stage - computed on a stream of their own, off the
#             chain conv_p2 -> conv_p3 -> conv_p4 -> toplayer -> res5.
# Same kernels, same operands, same summation order: the same bits.
LATERALS_OFF_THE_CHAIN = False
_lat_streams = {}


def _lateral_stream(device):
    import torch
    key = (device.type, device.index)
    if key not in _lat_streams:
        _lat_streams[key] = torch.cuda.Stream(device=device)
    return _lat_streams[key]


class FeaturePyramidNetwork(object):
    feat_strides = [4, 8, 16, 32, 64]
    spatial_scales = list(map(lambda x: 1. / x, feat_strides))
    anchor_base = 16
    anchor_sizes = [32, 64, 128, 256, 512]
    anchor_scales = list(map(lambda x: x / 16., anchor_sizes))

    STAGES = (('res2', 3, 64, 64, 256, 1), ('res3', 4, 256, 128, 512, 2),
              ('res4', 6, 512, 256, 1024, 2), ('res5', 3, 1024, 512, 2048, 2))

    def __init__(self, ps=None, prefix='extractor', stages=None, width_div=1):
        """``stages`` / ``width_div`` shrink the network for tests (blocks per stage, channel divisor);
        the defaults are the reference's ResNet-50."""
        self.ps = ps if ps is not None else ParamStore()
        p = prefix + '/'
        d = width_div
        self.conv1 = Conv(self.ps, p + 'resnet/conv1', 3, 64 // d, 7, 2, 3, bias=True, in_backbone=True)
        self.bn1 = BatchNorm(self.ps, p + 'resnet/bn1', 64 // d)
        self.stages = []
        for si, (name, n, cin, mid, cout, stride) in enumerate(self.STAGES):
            n = n if stages is None else stages[si]
            cin, mid, cout = cin // d, mid // d, cout // d
            blocks = [Bottleneck(self.ps, p + 'resnet/%s/a' % name, cin, mid, cout, stride, True)]
            for i in range(1, n):
                blocks.append(Bottleneck(self.ps, p + 'resnet/%s/b%d' % (name, i), cout, mid, cout, 1, False))
            self.stages.append(blocks)
        fc = 256 // d
        self.out_channels = fc
        self.toplayer = Conv(self.ps, p + 'toplayer', 2048 // d, fc, 1)
        self.conv_p4 = Conv(self.ps, p + 'conv_p4', fc, fc, 3, 1, 1, fwd_tile=0)
        self.conv_p3 = Conv(self.ps, p + 'conv_p3', fc, fc, 3, 1, 1, fwd_tile=0)
        self.conv_p2 = Conv(self.ps, p + 'conv_p2', fc, fc, 3, 1, 1, fwd_tile=0)
        self.conv_p6 = Conv(self.ps, p + 'conv_p6', fc, fc, 1, 2, 0)
        self.lat_p4 = Conv(self.ps, p + 'lat_p4', 1024 // d, fc, 1)
        self.lat_p3 = Conv(self.ps, p + 'lat_p3', 512 // d, fc, 1)
        self.lat_p2 = Conv(self.ps, p + 'lat_p2', 256 // d, fc, 1)
        self.anchor_scales = list(map(lambda x: x / float(self.anchor_base), self.anchor_sizes))

    def __call__(self, x):
        """x (N,H,W,4) NHWC.  Returns (p2, p3, p4, p5, p6) NHWC and keeps the tape for backward()."""
        t = {}
        h, t['conv1'] = self.conv1.fwd(x)
        h, t['bn1'] = self.bn1.fwd(h, relu=True)
        t['pool_in'] = h
        h = ops.maxpool2x2_fwd(h)
        cs = []
        t['blocks'] = []
        off_chain = LATERALS_OFF_THE_CHAIN and x.is_cuda and hnn.PROFILE is None
        lat = {}
        if off_chain:
            import torch
            main, side = torch.cuda.current_stream(x.device), hnn.side_stream(x.device)
        for si, blocks in enumerate(self.stages):
            for b in blocks:
                h, ctx = b.fwd(h)
                t['blocks'].append((b, ctx))
            cs.append(h)
            if off_chain and si < 3:        # the stage's lateral, beside the stages that follow
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    lat[si] = (self.lat_p2, self.lat_p3, self.lat_p4)[si].fwd(h)
        c2, c3, c4, c5 = cs
        if off_chain:
            main.wait_stream(side)
            for l_, _ in lat.values():
                l_.record_stream(main)      # allocated on the side stream, consumed (and released) on the main stream
        p5, t['top'] = self.toplayer.fwd(c5)
        l4, t['lat4'] = lat[2] if off_chain else self.lat_p4.fwd(c4)
        m4 = ops.upsample2x_add_fwd(p5, l4)
        p4, t['p4'] = self.conv_p4.fwd(m4)
        l3, t['lat3'] = lat[1] if off_chain else self.lat_p3.fwd(c3)
        m3 = ops.upsample2x_add_fwd(p4, l3)
        p3, t['p3'] = self.conv_p3.fwd(m3)
        l2, t['lat2'] = lat[0] if off_chain else self.lat_p2.fwd(c2)
        m2 = ops.upsample2x_add_fwd(p3, l2)
        p2, t['p2'] = self.conv_p2.fwd(m2)
        p6, t['p6'] = self.conv_p6.fwd(p5)
        self.tape = t
        return p2, p3, p4, p5, p6

    def backward(self, grads, progress=None):
        """grads: [g_p2, g_p3, g_p4, g_p5, g_p6]; consumed (accumulated into in place).  ``progress(prefix)`` is
        called when every parameter registered at or after ``prefix`` has its final gradient (data-parallel overlap)."""
        t = self.tape
        g_p2, g_p3, g_p4, g_p5, g_p6 = grads
        self.conv_p6.bwd(t['p6'], g_p6, gx_acc=g_p5)
        g_m2 = self.conv_p2.bwd(t['p2'], g_p2)
        ops.upsample2x_bwd(g_m2, gtop=g_p3)
        # c2..c5 are ReLU outputs (the last block of a stage): every contribution to their gradient is masked where it is written
        # (mask_gx: data-gradient epilogue / lattice scatter), so the blocks' bn3 backward gets its gy with the mask applied
        off_chain = LATERALS_OFF_THE_CHAIN and g_p2.is_cuda and hnn.PROFILE is None
        if off_chain:
            import torch
            main, ls = torch.cuda.current_stream(g_p2.device), _lateral_stream(g_p2.device)

        def lateral_bwd(conv, ctx, g_m):
            if not off_chain:
                return conv.bwd(ctx, g_m, mask_gx=MASK_IN_PRODUCER)
            ls.wait_stream(main)                # g_m is complete on the main stream
            g_m.record_stream(ls)
            with torch.cuda.stream(ls):         # (the filter gradient goes from here to the weight-gradient stream as everywhere)
                g_c = conv.bwd(ctx, g_m, mask_gx=MASK_IN_PRODUCER)
            g_c.record_stream(main)             # accumulated into by the next stage's first block, on the main stream
            return g_c
        g_c2 = lateral_bwd(self.lat_p2, t['lat2'], g_m2)
        g_m3 = self.conv_p3.bwd(t['p3'], g_p3)
        ops.upsample2x_bwd(g_m3, gtop=g_p4)
        g_c3 = lateral_bwd(self.lat_p3, t['lat3'], g_m3)
        g_m4 = self.conv_p4.bwd(t['p4'], g_p4)
        ops.upsample2x_bwd(g_m4, gtop=g_p5)
        g_c4 = lateral_bwd(self.lat_p4, t['lat4'], g_m4)
        g_c5 = self.toplayer.bwd(t['top'], g_p5, mask_gx=MASK_IN_PRODUCER)
        if progress:
            progress(self.toplayer.name)        # the FPN layers are registered after the ResNet
        # The gradient of c2..c4 is (lateral gradient) + (input gradient of the next stage): the next
        # stage's first block accumulates its input gradient into the lateral one (gx_acc).
        acc_for = {id(self.stages[k][0]): g for k, g in ((1, g_c2), (2, g_c3), (3, g_c4))}
        g = g_c5
        first = t['blocks'][0][0]
        for b, ctx in reversed(t['blocks']):
            # every block's output gradient arrives masked (from the block after it, or from toplayer / the laterals above); its own
            # input gradient is masked for the block before it - except the first block, whose input is the max-pool output
            if off_chain and id(b) in acc_for:
                main.wait_stream(ls)            # the lateral gradient this block accumulates into
            g = b.bwd(ctx, g, gx_acc=acc_for.get(id(b)), gy_masked=MASK_IN_PRODUCER, mask_gx=MASK_IN_PRODUCER and b is not first)
            if progress:
                progress(b.conv1.name)
        g = ops.maxpool2x2_bwd(t['pool_in'], g)
        g, _ = self.bn1.bwd(t['bn1'], g)
        self.conv1.bwd(t['conv1'], g, need_gx=False)
        self.tape = None
