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


# ---- legacy variants: C4Backbone, Darknet, LightRoIMaskHead, ResnetRoIMaskHead ---------------------
def _bottleneck(out, p, cin, mid, cout, project):
    conv = lambda name, co, ci, k: out.append((name, 'conv', (co, ci, k, k), False))
    bn = lambda name, c: out.append((name, 'bn', (c,), False))
    conv(p + 'conv1', mid, cin, 1); bn(p + 'bn1', mid)
    conv(p + 'conv2', mid, mid, 3); bn(p + 'bn2', mid)
    conv(p + 'conv3', cout, mid, 1); bn(p + 'bn3', cout)
    if project:
        conv(p + 'conv4', cout, cin, 1); bn(p + 'bn4', cout)


def legacy_layer_list(kind, n_class=6, in_channels=256):
    """[(snapshot key prefix relative to the link, kind, Chainer shape of W, has bias)] of one legacy component, the keys as
    ``chainer.serializers.save_npz`` would write them for that link: 'c4' = C4Backbone (ResNet50Layers conv1 .. res4),
    'darknet' = Darknet (five ConvBatch), 'light' = LightRoIMaskHead on an ``in_channels`` feature map, 'res5' =
    ResnetRoIMaskHead (res5 block + conv1 + deconv1 / conv2 + cls_loc / score)."""
    out = []
    if kind == 'c4':
        out.append(('conv1', 'conv', (64, 3, 7, 7), True)); out.append(('bn1', 'bn', (64,), False))
        cin = 64
        for stage, n, mid in (('res2', 3, 64), ('res3', 4, 128), ('res4', 6, 256)):
            cout = 4 * mid
            for i in range(n):
                _bottleneck(out, '%s/%s/' % (stage, 'a' if i == 0 else 'b%d' % i), cin if i == 0 else cout, mid, cout, i == 0)
            cin = cout
    elif kind == 'darknet':
        cin = 3
        for i, cout in enumerate((16, 32, 64, 128, 256)):
            out.append(('conv%d/c' % (i + 1), 'conv', (cout, cin, 3, 3), True)); out.append(('conv%d/bn' % (i + 1), 'bn', (cout,), False))
            cin = cout
    elif kind == 'light':
        k, cm, co = 15, 256, 490
        out.append(('conv_ul', 'conv', (cm, in_channels, k, 1), True)); out.append(('conv_bl', 'conv', (co, cm, 1, k), True))
        out.append(('conv_ur', 'conv', (cm, in_channels, 1, k), True)); out.append(('conv_br', 'conv', (co, cm, k, 1), True))
        out.append(('fc', 'linear', (2048, co * 7 * 7), True))
        out.append(('cls_loc', 'linear', (4, 2048), True)); out.append(('score', 'linear', (n_class, 2048), True))
        out.append(('conv2', 'conv', (256, co, 3, 3), True)); out.append(('conv3_', 'conv', (256, 256, 3, 3), True))
        out.append(('conv4', 'conv', (256, 256, 3, 3), True))
        out.append(('deconv1_', 'deconv', (co, n_class - 1, 2, 2), True))
    elif kind == 'res5':
        _bottleneck(out, 'res5/a/', 1024, 512, 2048, True)
        _bottleneck(out, 'res5/b1/', 2048, 512, 2048, False)
        _bottleneck(out, 'res5/b2/', 2048, 512, 2048, False)
        out.append(('conv1', 'conv', (2048, 2048, 3, 3), True))
        out.append(('deconv1', 'deconv', (2048, 256, 2, 2), True))
        out.append(('conv2', 'conv', (n_class - 1, 256, 3, 3), True))
        out.append(('cls_loc', 'linear', (n_class * 4, 2048), True)); out.append(('score', 'linear', (n_class, 2048), True))
    else:
        raise ValueError(kind)
    return out


def legacy_chainer_weights(kind, seed, **kw):
    """Seeded arrays of one legacy component in Chainer's layouts (see chainer_weights)."""
    rs = np.random.RandomState(seed)
    d = {}
    for name, lk, shape, bias in legacy_layer_list(kind, **kw):
        if lk == 'bn':
            d[name + '/gamma'] = (1.0 + 0.2 * rs.standard_normal(shape)).astype(np.float32)
            d[name + '/beta'] = (0.1 * rs.standard_normal(shape)).astype(np.float32)
            continue
        fan_in = int(np.prod(shape[1:])) if lk != 'deconv' else shape[0]
        std = np.sqrt(1.5 / fan_in)
        if name in ('cls_loc', 'score'):
            std = 0.03
        d[name + '/W'] = (std * rs.standard_normal(shape)).astype(np.float32)
        if bias:
            nb = shape[1] if lk == 'deconv' else shape[0]
            d[name + '/b'] = (0.05 * rs.standard_normal((nb,))).astype(np.float32)
    return d


def legacy_native(kind, arrays, shapes, prefix, **kw):
    """The same arrays in the product's storage convention, keyed by ParamStore name (``prefix`` + key): convolution W
    (Cout_p, KH, KW, Cin_p) zero-padded; Linear W (out_p, 1, 1, in_p) - LightRoIMaskHead's ``fc`` with its input re-ordered
    from Chainer's (c, h, w) flattening to (h, w, c_p); a 2x2 / 2 Deconvolution2D W (Cin, Cout, 2, 2) as the 1x1 convolution
    to ((a*2+b)*Cout_p + o) output channels the product runs (+ its bias under '<name>/b'); padded BatchNorm channels keep
    gamma 1 / beta 0.  ``shapes``: ParamStore name -> padded shape (``ParamStore.offsets`` of the product layer)."""
    out = {}
    for name, lk, shape, bias in legacy_layer_list(kind, **kw):
        full = prefix + name
        if lk == 'bn':
            for k, fill in (('gamma', 1.0), ('beta', 0.0)):
                t = np.full(shapes[full + '/' + k], fill, np.float32)
                t[:shape[0]] = arrays[name + '/' + k]
                out[full + '/' + k] = t
            continue
        w = arrays[name + '/W']
        tgt = np.zeros(shapes[full + '/W'], np.float32)
        if lk == 'conv':
            co, ci, kh, kw_ = w.shape
            tgt[:co, :, :, :ci] = w.transpose(0, 2, 3, 1)
        elif lk == 'linear' and kind == 'light' and name == 'fc':
            co, s = 490, 7
            cp = tgt.shape[3] // (s * s)
            v = np.zeros((w.shape[0], s, s, cp), np.float32)
            v[..., :co] = w.reshape(-1, co, s, s).transpose(0, 2, 3, 1)
            tgt[:w.shape[0], 0, 0, :] = v.reshape(w.shape[0], -1)
        elif lk == 'linear':
            tgt[:w.shape[0], 0, 0, :w.shape[1]] = w
        else:                                       # deconv (ci, o, a, b) -> ((a*2+b)*Cout_p + o, 1, 1, ci)
            ci, co = w.shape[:2]
            cop = tgt.shape[0] // 4
            for a in range(2):
                for b in range(2):
                    tgt[(a * 2 + b) * cop:(a * 2 + b) * cop + co, 0, 0, :ci] = w[:, :, a, b].T
        out[full + '/W'] = tgt
        if bias:
            t = np.zeros(shapes[full + '/b'], np.float32)
            t[:arrays[name + '/b'].shape[0]] = arrays[name + '/b']
            out[full + '/b'] = t
    return out
