"""Seeded weights of the default model (ResNet-50 FPN, n_fg_class foreground classes) keyed and laid out like a Chainer
snapshot of ``model.faster_rcnn`` (train.py:134-137): Convolution2D W (Cout, Cin, KH, KW), Linear W (out, in),
Deconvolution2D W (Cin, Cout, KH, KW), BatchNormalization gamma / beta.  NumPy RandomState only, so the generator
(tests/golden/make_step_reference.py, build container) and the tests (anywhere) derive the same arrays from the seed and
the fixture does not have to carry 44 M numbers.  Test infrastructure."""
import numpy as np


def layer_list(n_fg_class=80, n_keypoints=None, n_mask_convs=8):
    """[(key prefix, kind, shape of W, has bias)] in a fixed order."""
    out = []
    conv = lambda name, cout, cin, k, bias=True: out.append((name, 'conv', (cout, cin, k, k), bias))
    bn = lambda name, c: out.append((name, 'bn', (c,), False))
    e = 'extractor/resnet/'
    conv(e + 'conv1', 64, 3, 7)
    bn(e + 'bn1', 64)
    cin = 64
    for stage, n, mid in (('res2', 3, 64), ('res3', 4, 128), ('res4', 6, 256), ('res5', 3, 512)):
        cout = 4 * mid
        for i in range(n):
            p = e + '%s/%s/' % (stage, 'a' if i == 0 else 'b%d' % i)
            conv(p + 'conv1', mid, cin if i == 0 else cout, 1, False); bn(p + 'bn1', mid)
            conv(p + 'conv2', mid, mid, 3, False); bn(p + 'bn2', mid)
            conv(p + 'conv3', cout, mid, 1, False); bn(p + 'bn3', cout)
            if i == 0:
                conv(p + 'conv4', cout, cin, 1, False); bn(p + 'bn4', cout)
        cin = cout
    conv('extractor/toplayer', 256, 2048, 1)
    conv('extractor/conv_p4', 256, 256, 3); conv('extractor/conv_p3', 256, 256, 3); conv('extractor/conv_p2', 256, 256, 3)
    conv('extractor/conv_p6', 256, 256, 1)
    conv('extractor/lat_p4', 256, 1024, 1); conv('extractor/lat_p3', 256, 512, 1); conv('extractor/lat_p2', 256, 256, 1)
    conv('rpn/conv', 256, 256, 3); conv('rpn/score', 6, 256, 1); conv('rpn/loc', 12, 256, 1)
    conv('head/conv1', 256, 256, 3)
    out.append(('head/fc1', 'linear', (1024, 256 * 7 * 7), True))
    out.append(('head/fc2', 'linear', (1024, 1024), True))
    out.append(('head/cls_loc', 'linear', (4, 1024), True))
    out.append(('head/score', 'linear', (n_fg_class + 1, 1024), True))
    if n_keypoints is None:         # FPNRoIMaskHead: mask1..mask4, one mask per foreground class
        for i in range(1, 5):
            conv('head/mask%d' % i, 256, 256, 3)
    else:                           # FPNRoIKeypointHead: a ChainList of n_mask_convs convolutions, one heat map per keypoint
        for i in range(n_mask_convs):
            conv('head/mask_convs/%d' % i, 256, 256, 3)
    out.append(('head/deconv1', 'deconv', (256, 256, 2, 2), True))
    conv('head/conv2', n_fg_class if n_keypoints is None else n_keypoints, 256, 1)
    return out


def chainer_weights(seed, n_fg_class=80, n_keypoints=None, n_mask_convs=8):
    rs = np.random.RandomState(seed)
    d = {}
    for name, kind, shape, bias in layer_list(n_fg_class, n_keypoints, n_mask_convs):
        if kind == 'bn':
            d[name + '/gamma'] = (1.0 + 0.2 * rs.standard_normal(shape)).astype(np.float32)
            d[name + '/beta'] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
            continue
        fan_in = int(np.prod(shape[1:])) if kind != 'deconv' else shape[0]
        std = np.sqrt(1.5 / fan_in)
        if name in ('rpn/score', 'rpn/loc', 'head/cls_loc', 'head/score'):
            std = 0.03          # small decoders: proposals stay near their anchors, box deltas near zero
        d[name + '/W'] = (std * rs.standard_normal(shape)).astype(np.float32)
        if bias:
            nb = shape[1] if kind == 'deconv' else shape[0]
            d[name + '/b'] = (0.05 * rs.standard_normal((nb,))).astype(np.float32)
    return d
