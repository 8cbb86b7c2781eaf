"""Stand-ins for the THIRD-PARTY packages of the reference (Chainer, ChainerCV) - just enough of their documented semantics
 for the reference's OWN model and training-chain code to execute in the build container:

    FeaturePyramidNetwork.__call__, MultilevelRegionProposalNetwork.__call__, FPNRoIMaskHead.__call__, MaskRCNN.__init__,
    FPNMaskRCNNTrainChain.__call__, ProposalTargetCreator.__call__, calc_mask_loss        (all under /root/reference)

run on these primitives, in float64 NumPy (torch-CPU float64 for the convolutions), forward only.  TEST INFRASTRUCTURE, used
by tests/golden/make_reference_vectors.py alone; nothing here is product code and nothing here is taken from the reference -
every function restates the public behaviour of the Chainer / ChainerCV API it is named after:

    chainer.Chain / ChainList / init_scope / Variable(.array, .data) / cuda / config / using_config / reporter
    chainer.links: Convolution2D, Linear, Deconvolution2D, BatchNormalization (training mode: batch statistics, eps 2e-5)
    chainer.links.model.vision.resnet: ResNet50Layers, BuildingBlock (BottleneckA / BottleneckB, stride on the first 1x1)
    chainer.functions: relu, max_pooling_2d (cover_all), unpooling_2d, resize_images, concat, softmax, sigmoid,
      softmax_cross_entropy, sigmoid_cross_entropy
    chainercv: FasterRCNN, FasterRCNNTrainChain, _fast_rcnn_loc_loss, AnchorTargetCreator, ProposalCreator, anchors
      (the last three are this repo's oracle restatements, as everywhere in this generator)

What a fixture made with them pins is therefore the reference's WIRING - which layer feeds which, where the ReLUs and the
pooling sit, how RoIs and targets flow, which rows enter which loss - not the third-party arithmetic."""
import contextlib
import types

import numpy as np
import torch
import torch.nn.functional as TF

D = np.float64


class Var(np.ndarray):
    """chainer.Variable: here an ndarray that also answers .array / .data."""

    def __new__(cls, data):          # chainer.Variable(array)
        return np.ascontiguousarray(np.asarray(data, D)).view(cls)

    @property
    def array(self):
        return np.asarray(self)

    @property
    def data(self):
        return np.asarray(self)


def V(x):
    return np.ascontiguousarray(np.asarray(x, D)).view(Var)


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, D)))


# ---- chainer core -------------------------------------------------------------------------------------------------------
class Chain(object):
    xp = np

    def __init__(self, *a, **k):
        pass

    @contextlib.contextmanager
    def init_scope(self):
        yield


class ChainList(Chain):
    """chainer.ChainList: children are registered under their index ('0', '1', ...: the snapshot keys)."""

    def __init__(self, *links):
        self._children = []
        for l in links:
            self.add_link(l)

    def add_link(self, link):
        setattr(self, str(len(self._children)), link)
        self._children.append(link)

    def children(self):
        return iter(self._children)

    def __iter__(self):
        return iter(self._children)

    def __len__(self):
        return len(self._children)


config = types.SimpleNamespace(train=True, enable_backprop=True)


@contextlib.contextmanager
def using_config(name, value):
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)

REPORTED = {}


def report(values, observer=None):
    REPORTED.clear()
    REPORTED.update({k: float(np.asarray(v)) for k, v in values.items()})


class _Normal(object):          # chainer.initializers.Normal: weights are assigned from outside, initialisers are inert
    def __init__(self, *a, **k):
        pass


# ---- links --------------------------------------------------------------------------------------------------------------
class Convolution2D(Chain):
    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, initialW=None, initial_bias=None, **kw):
        self.stride, self.pad, self.nobias = stride, pad, nobias
        self.W = self.b = None

    def __call__(self, x):
        return V(TF.conv2d(_t(x), _t(self.W), None if self.b is None else _t(self.b), self.stride, self.pad).numpy())


class Deconvolution2D(Chain):
    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, initialW=None, **kw):
        self.stride, self.pad = stride, pad
        self.W = self.b = None          # W: (in, out, kh, kw)

    def __call__(self, x):
        return V(TF.conv_transpose2d(_t(x), _t(self.W), None if self.b is None else _t(self.b), self.stride, self.pad).numpy())


class Linear(Chain):
    def __init__(self, in_size, out_size=None, nobias=False, initialW=None, **kw):
        self.W = self.b = None          # W: (out, in); the input is flattened from axis 1 (C, H, W order)

    def __call__(self, x):
        x = np.asarray(x, D)
        return V(x.reshape(x.shape[0], -1) @ np.asarray(self.W, D).T + np.asarray(self.b, D))


class BatchNormalization(Chain):
    def __init__(self, size, **kw):
        self.gamma = self.beta = self.avg_mean = self.avg_var = None
        self.eps = 2e-5
        self.last_mean = self.last_var = None
        self.update_enabled = True

    def disable_update(self):           # chainer.Link.disable_update: the optimizer skips this link's parameters (forward unchanged)
        self.update_enabled = False

    def __call__(self, x):
        x = np.asarray(x, D)
        if config.train:            # batch statistics over (N, H, W), biased variance (kept: a generator may adopt them as the
            m = x.mean(axis=(0, 2, 3), keepdims=True)          # running statistics of a later inference-mode call)
            v = x.var(axis=(0, 2, 3), keepdims=True)
            self.last_mean, self.last_var = m.reshape(-1), v.reshape(-1)
        else:                       # inference: the running statistics
            m = np.asarray(self.avg_mean, D)[None, :, None, None]
            v = np.asarray(self.avg_var, D)[None, :, None, None]
        g, b = np.asarray(self.gamma, D)[None, :, None, None], np.asarray(self.beta, D)[None, :, None, None]
        return V(g * (x - m) / np.sqrt(v + self.eps) + b)


def relu(x):
    return V(np.maximum(np.asarray(x, D), 0.0))


class BottleneckA(Chain):
    def __init__(self, in_channels, mid_channels, out_channels, stride=2):
        self.conv1 = Convolution2D(in_channels, mid_channels, 1, stride, 0, nobias=True)
        self.bn1 = BatchNormalization(mid_channels)
        self.conv2 = Convolution2D(mid_channels, mid_channels, 3, 1, 1, nobias=True)
        self.bn2 = BatchNormalization(mid_channels)
        self.conv3 = Convolution2D(mid_channels, out_channels, 1, 1, 0, nobias=True)
        self.bn3 = BatchNormalization(out_channels)
        self.conv4 = Convolution2D(in_channels, out_channels, 1, stride, 0, nobias=True)
        self.bn4 = BatchNormalization(out_channels)

    def __call__(self, x):
        h1 = relu(self.bn1(self.conv1(x)))
        h1 = relu(self.bn2(self.conv2(h1)))
        h1 = self.bn3(self.conv3(h1))
        h2 = self.bn4(self.conv4(x))
        return relu(h1 + h2)


class BottleneckB(Chain):
    def __init__(self, in_channels, mid_channels):
        self.conv1 = Convolution2D(in_channels, mid_channels, 1, 1, 0, nobias=True)
        self.bn1 = BatchNormalization(mid_channels)
        self.conv2 = Convolution2D(mid_channels, mid_channels, 3, 1, 1, nobias=True)
        self.bn2 = BatchNormalization(mid_channels)
        self.conv3 = Convolution2D(mid_channels, in_channels, 1, 1, 0, nobias=True)
        self.bn3 = BatchNormalization(in_channels)

    def __call__(self, x):
        h = relu(self.bn1(self.conv1(x)))
        h = relu(self.bn2(self.conv2(h)))
        h = self.bn3(self.conv3(h))
        return relu(h + x)


class BuildingBlock(Chain):
    def __init__(self, n_layer, in_channels, mid_channels, out_channels, stride):
        self.a = BottleneckA(in_channels, mid_channels, out_channels, stride)
        self._names = ['a']
        for i in range(n_layer - 1):
            setattr(self, 'b%d' % (i + 1), BottleneckB(out_channels, mid_channels))
            self._names.append('b%d' % (i + 1))

    def __call__(self, x):
        for n in self._names:
            x = getattr(self, n)(x)
        return x


class ResNet50Layers(Chain):
    def __init__(self, pretrained_model='auto'):
        self.conv1 = Convolution2D(3, 64, 7, 2, 3)
        self.bn1 = BatchNormalization(64)
        self.res2 = BuildingBlock(3, 64, 64, 256, 1)
        self.res3 = BuildingBlock(4, 256, 128, 512, 2)
        self.res4 = BuildingBlock(6, 512, 256, 1024, 2)
        self.res5 = BuildingBlock(3, 1024, 512, 2048, 2)
        self.fc6 = Linear(2048, 1000)

    def links(self):
        """chainer.Chain.links(): this link and every descendant (used by C4Backbone to find its BatchNormalization links)."""
        seen, stack = set(), [self]
        while stack:
            l = stack.pop()
            if id(l) in seen:
                continue
            seen.add(id(l))
            yield l
            for v in vars(l).values():
                if isinstance(v, Chain):
                    stack.append(v)

    def __call__(self, x, layers=['prob'], **kwargs):
        """chainer ResNet50Layers.__call__: the entries of the ``functions`` OrderedDict are applied in order; the outputs
        named in ``layers`` are collected, and evaluation stops once all of them have been produced."""
        h, out, todo = x, {}, set(layers)
        for key, funcs in self.functions.items():
            if not todo:
                break
            for f in funcs:
                h = f(h)
            if key in todo:
                out[key] = h
                todo.discard(key)
        return out


def _global_average_pooling_2d(x):
    """chainer.links.model.vision.resnet._global_average_pooling_2d: average_pooling_2d over the whole map, reshaped (n, c)."""
    x = np.asarray(x, D)
    return V(x.mean(axis=(2, 3)))


# ---- functions ----------------------------------------------------------------------------------------------------------
def max_pooling_2d(x, ksize, stride=None, pad=0, cover_all=True):
    k = ksize if isinstance(ksize, int) else ksize[0]
    s = k if stride is None else (stride if isinstance(stride, int) else stride[0])
    return V(TF.max_pool2d(_t(x), k, s, pad, ceil_mode=bool(cover_all)).numpy())


def unpooling_2d(x, ksize, stride=None, pad=0, outsize=None, cover_all=True):
    x = np.asarray(x, D)
    y = np.repeat(np.repeat(x, ksize, axis=2), ksize, axis=3)
    return V(y[:, :, :outsize[0], :outsize[1]])


def resize_images(x, output_shape):
    """chainer.functions.resize_images: bilinear with the corner pixels aligned, u = linspace(0, H - 1, out_H)
    (SURVEY.md Appendix A-8)."""
    x = np.asarray(x, D)

    def axis(n, m):
        u = np.linspace(0, n - 1, m)
        u0 = np.clip(np.floor(u).astype(np.int64), 0, max(n - 2, 0))
        u1 = np.minimum(u0 + 1, n - 1)
        return u0, u1, u0 + 1 - u, u - u0
    y0, y1, wy0, wy1 = axis(x.shape[2], output_shape[0])
    x0, x1, wx0, wx1 = axis(x.shape[3], output_shape[1])
    rows = x[:, :, y0] * wy0[None, None, :, None] + x[:, :, y1] * wy1[None, None, :, None]
    return V(rows[:, :, :, x0] * wx0 + rows[:, :, :, x1] * wx1)


def concat(xs, axis=1):
    return V(np.concatenate([np.asarray(x, D) for x in xs], axis=axis))


def softmax(x, axis=1):
    x = np.asarray(x, D)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return V(e / e.sum(axis=axis, keepdims=True))


def sigmoid(x):
    return V(1.0 / (1.0 + np.exp(-np.asarray(x, D))))


def softmax_cross_entropy(x, t, normalize=True, ignore_label=-1):
    x, t = np.asarray(x, D), np.asarray(t)
    valid = t != ignore_label
    z = x - x.max(axis=1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    picked = logp[np.arange(len(t)), np.where(valid, t, 0)]
    return V(-(picked * valid).sum() / max(int(valid.sum()), 1))


def sigmoid_cross_entropy(x, t, normalize=True):
    x, t = np.asarray(x, D), np.asarray(t)
    valid = t != -1
    loss = -(valid * (x * (t - (x >= 0)) - np.log1p(np.exp(-np.abs(x)))))
    return V(loss.sum() / max(int(valid.sum()), 1))


# ---- chainercv ----------------------------------------------------------------------------------------------------------
class FasterRCNN(Chain):
    def __init__(self, extractor, rpn, head, mean, min_size=600, max_size=1000, loc_normalize_mean=(0., 0., 0., 0.),
                 loc_normalize_std=(0.1, 0.1, 0.2, 0.2)):
        self.extractor, self.rpn, self.head = extractor, rpn, head
        self.mean, self.min_size, self.max_size = mean, min_size, max_size
        self.loc_normalize_mean, self.loc_normalize_std = loc_normalize_mean, loc_normalize_std
        self.use_preset('visualize')

    @property
    def n_class(self):
        return self.head.n_class

    def use_preset(self, preset):
        if preset == 'visualize':
            self.nms_thresh, self.score_thresh = 0.3, 0.7
        elif preset == 'evaluate':
            self.nms_thresh, self.score_thresh = 0.3, 0.05
        else:
            raise ValueError('preset must be visualize or evaluate')


def _smooth_l1_loss(x, t, in_weight, sigma):
    sigma2 = sigma ** 2
    diff = in_weight * (np.asarray(x, D) - np.asarray(t, D))
    abs_diff = np.abs(diff)
    flag = (abs_diff < (1. / sigma2)).astype(D)
    y = flag * (sigma2 / 2.) * diff * diff + (1 - flag) * (abs_diff - 0.5 / sigma2)
    return y.sum()


def _fast_rcnn_loc_loss(pred_loc, gt_loc, gt_label, sigma):
    gt_label = np.asarray(gt_label)
    in_weight = np.zeros(np.asarray(gt_loc).shape, D)
    in_weight[gt_label > 0] = 1
    loc_loss = _smooth_l1_loss(pred_loc, gt_loc, in_weight, sigma)
    return V(loc_loss / np.sum(gt_label >= 0))


class FasterRCNNTrainChain(Chain):
    def __init__(self, faster_rcnn, rpn_sigma=3., roi_sigma=1., anchor_target_creator=None, proposal_target_creator=None):
        from oracle import targets as otargets
        self.faster_rcnn = faster_rcnn
        self.rpn_sigma, self.roi_sigma = rpn_sigma, roi_sigma
        self.anchor_target_creator = anchor_target_creator if anchor_target_creator is not None else otargets.AnchorTargetCreator()
        self.proposal_target_creator = proposal_target_creator
        self.loc_normalize_mean = faster_rcnn.loc_normalize_mean
        self.loc_normalize_std = faster_rcnn.loc_normalize_std


class ProposalCreator(object):
    """chainercv ProposalCreator as the RPN calls it (score, anchors of all levels): this repo's oracle."""

    def __init__(self, **kw):
        from oracle import proposal as oproposal
        self.pc = oproposal.ProposalCreator(**kw)

    def __call__(self, loc, score, anchor, img_size, scale=1.):
        return self.pc(np.asarray(loc, np.float32), np.asarray(score, np.float32), np.asarray(anchor, np.float32), img_size,
                       scale=scale, train=bool(config.train))
