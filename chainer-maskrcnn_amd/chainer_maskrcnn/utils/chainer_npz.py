"""Chainer-NPZ weight files <-> the flat parameter store.

The reference snapshots ``model.faster_rcnn`` with ``chainer.serializers.save_npz`` (train.py:134-137) and loads
with ``load_npz(..., strict=False)`` (train.py:99-101).  NPZ keys are Chainer link paths; arrays use Chainer's
layouts:  Convolution2D W (Cout, Cin, KH, KW), Linear W (out, in) with the input flattened (C, H, W)-major,
Deconvolution2D W (Cin, Cout, KH, KW), BatchNormalization gamma / beta / avg_mean / avg_var / N.

Storage here differs only mechanically - conv W (Cout_p, KH, KW, Cin_p) with zero-padded channels, NHWC
flattening for ``fc1``, ``rpn/loc`` + ``rpn/score`` fused into ``rpn/loc_score``, ``head/cls_loc`` + ``head/score``
fused into ``head/score_cls_loc``, the deconvolution as a 1x1 convolution to (a, b, Cout) - so the mapping below is
exact in both directions (round-trip tested on the CPU).  Everything is host-side NumPy: it runs once at load/save.
"""
import numpy as np


def _conv_to_native(w, cout_p, cin_p):
    cout, cin, kh, kw = w.shape
    out = np.zeros((cout_p, kh, kw, cin_p), np.float32)
    out[:cout, :, :, :cin] = w.transpose(0, 2, 3, 1)
    return out


def _conv_from_native(wn, cout, cin):
    return np.ascontiguousarray(wn[:cout, :, :, :cin].transpose(0, 3, 1, 2))


def _vec_to_native(v, n_p):
    out = np.zeros((n_p,), np.float32)
    out[:v.shape[0]] = v
    return out


class ChainerNpzMap(object):
    """Built from a MaskRCNN instance (needs the layer objects for logical / padded channel counts)."""

    def __init__(self, model):
        self.m = model
        self.ps = model.ps
        ext, rpn, head = model.extractor, model.rpn, model.head
        self.convs = {}         # chainer link path -> Conv layer stored under the same name
        self.bns = []
        self._add_conv(ext.conv1)
        self.bns.append(ext.bn1)
        for blocks in ext.stages:
            for b in blocks:
                for cv in (b.conv1, b.conv2, b.conv3) + ((b.conv4,) if b.project else ()):
                    self._add_conv(cv)
                self.bns += [b.bn1, b.bn2, b.bn3] + ([b.bn4] if b.project else [])
        for cv in (ext.toplayer, ext.conv_p4, ext.conv_p3, ext.conv_p2, ext.conv_p6, ext.lat_p4, ext.lat_p3, ext.lat_p2,
                   rpn.conv, head.conv1, head.fc2, head.conv2) + tuple(head.mask_convs):
            self._add_conv(cv)
        self.rpn, self.head = rpn, head

    def _add_conv(self, cv):
        self.convs[cv.name] = cv

    # ------------------------------------------------------------------------------------------
    def to_chainer(self):
        """dict of NumPy arrays keyed like a Chainer snapshot of ``model.faster_rcnn``."""
        g = lambda n: self.ps.p(n).detach().cpu().numpy()
        out = {}
        for name, cv in self.convs.items():
            wn = g(name + '/W')
            if name.endswith('/fc2'):
                out[name + '/W'] = np.ascontiguousarray(wn[:cv.cout, 0, 0, :cv.cin])
            else:
                out[name + '/W'] = _conv_from_native(wn, cv.cout, cv.cin)
            if cv.has_bias:
                out[name + '/b'] = g(name + '/b')[:cv.cout].copy()
        for bn in self.bns:
            for k in ('gamma', 'beta'):
                out['%s/%s' % (bn.name, k)] = g('%s/%s' % (bn.name, k)).copy()
            for k in ('avg_mean', 'avg_var'):
                out['%s/%s' % (bn.name, k)] = self.ps.buffers['%s/%s' % (bn.name, k)].cpu().numpy().copy()
            out[bn.name + '/N'] = np.array(0)
        A = self.rpn.n_anchor
        w = g(self.rpn.head.name + '/W')[:, 0, 0, :self.rpn.head.cin]
        b = g(self.rpn.head.name + '/b')
        out['rpn/loc/W'] = w[:4 * A][:, :, None, None].copy()
        out['rpn/loc/b'] = b[:4 * A].copy()
        out['rpn/score/W'] = w[4 * A:6 * A][:, :, None, None].copy()
        out['rpn/score/b'] = b[4 * A:6 * A].copy()
        h = self.head
        w = g(h.box_out.name + '/W')[:, 0, 0, :h.box_out.cin]
        b = g(h.box_out.name + '/b')
        out['head/score/W'] = w[:h.n_class].copy()
        out['head/score/b'] = b[:h.n_class].copy()
        out['head/cls_loc/W'] = w[h.LOC0:h.LOC0 + 4].copy()
        out['head/cls_loc/b'] = b[h.LOC0:h.LOC0 + 4].copy()
        # fc1: ours (out, 1, 1, (h, w, c)) -> Chainer Linear (out, (c, h, w))
        c, s = h.channels, h.roi_size_box
        w = g(h.fc1.name + '/W')[:h.fc1.cout, 0, 0, :]
        out['head/fc1/W'] = np.ascontiguousarray(w.reshape(-1, s, s, c).transpose(0, 3, 1, 2).reshape(w.shape[0], -1))
        out['head/fc1/b'] = g(h.fc1.name + '/b')[:h.fc1.cout].copy()
        # deconv1: ours ((a*2+b)*C + o, 1, 1, ci) -> Chainer Deconvolution2D (ci, o, a, b)
        w = g(h.deconv1.name + '/W')[:, 0, 0, :c].reshape(2, 2, c, c)
        out['head/deconv1/W'] = np.ascontiguousarray(w.transpose(3, 2, 0, 1))
        out['head/deconv1/b'] = g(h.deconv_b).copy()
        return out

    def from_chainer(self, arrays, strict=False):
        """Write the arrays of a Chainer snapshot into the parameter store.  Returns the list of keys loaded."""
        import torch
        loaded = []

        def put(name, arr):
            dst = self.ps.p(name) if name in self.ps.offsets else self.ps.buffers[name]
            dst.copy_(torch.from_numpy(np.ascontiguousarray(arr, np.float32)).reshape(dst.shape))

        def have(*keys):
            ok = all(k in arrays for k in keys)
            if not ok and strict:
                raise KeyError('missing %s' % (keys,))
            return ok

        for name, cv in self.convs.items():
            if name == self.head.fc1.name:
                continue
            if have(name + '/W'):
                w = np.asarray(arrays[name + '/W'])
                if w.ndim == 2:
                    w = w[:, :, None, None]
                put(name + '/W', _conv_to_native(w, cv.cout_p, cv.cin_p))
                loaded.append(name + '/W')
            if cv.has_bias and have(name + '/b'):
                put(name + '/b', _vec_to_native(np.asarray(arrays[name + '/b']), cv.cout_p))
                loaded.append(name + '/b')
        for bn in self.bns:
            for k in ('gamma', 'beta', 'avg_mean', 'avg_var'):
                key = '%s/%s' % (bn.name, k)
                if have(key):
                    put(key, arrays[key])
                    loaded.append(key)
        A = self.rpn.n_anchor
        if have('rpn/loc/W', 'rpn/score/W', 'rpn/loc/b', 'rpn/score/b'):
            hd = self.rpn.head
            w = np.zeros((hd.cout_p, 1, 1, hd.cin_p), np.float32)
            b = np.zeros((hd.cout_p,), np.float32)
            w[:4 * A, 0, 0, :hd.cin] = np.asarray(arrays['rpn/loc/W']).reshape(4 * A, -1)
            w[4 * A:6 * A, 0, 0, :hd.cin] = np.asarray(arrays['rpn/score/W']).reshape(2 * A, -1)
            b[:4 * A] = arrays['rpn/loc/b']
            b[4 * A:6 * A] = arrays['rpn/score/b']
            put(hd.name + '/W', w)
            put(hd.name + '/b', b)
            loaded += ['rpn/loc/W', 'rpn/score/W', 'rpn/loc/b', 'rpn/score/b']
        h = self.head
        if have('head/score/W', 'head/cls_loc/W', 'head/score/b', 'head/cls_loc/b'):
            bo = h.box_out
            w = np.zeros((bo.cout_p, 1, 1, bo.cin_p), np.float32)
            b = np.zeros((bo.cout_p,), np.float32)
            w[:h.n_class, 0, 0, :bo.cin] = arrays['head/score/W']
            w[h.LOC0:h.LOC0 + 4, 0, 0, :bo.cin] = arrays['head/cls_loc/W']
            b[:h.n_class] = arrays['head/score/b']
            b[h.LOC0:h.LOC0 + 4] = arrays['head/cls_loc/b']
            put(bo.name + '/W', w)
            put(bo.name + '/b', b)
            loaded += ['head/score/W', 'head/cls_loc/W', 'head/score/b', 'head/cls_loc/b']
        c, s = h.channels, h.roi_size_box
        if have('head/fc1/W', 'head/fc1/b'):
            w = np.asarray(arrays['head/fc1/W'])
            wn = np.zeros((h.fc1.cout_p, 1, 1, h.fc1.cin_p), np.float32)
            wn[:w.shape[0], 0, 0, :] = w.reshape(-1, c, s, s).transpose(0, 2, 3, 1).reshape(w.shape[0], -1)
            put(h.fc1.name + '/W', wn)
            put(h.fc1.name + '/b', _vec_to_native(np.asarray(arrays['head/fc1/b']), h.fc1.cout_p))
            loaded += ['head/fc1/W', 'head/fc1/b']
        if have('head/deconv1/W', 'head/deconv1/b'):
            w = np.asarray(arrays['head/deconv1/W'])            # (ci, o, a, b)
            wn = np.zeros((4 * c, 1, 1, h.deconv1.cin_p), np.float32)
            wn[:, 0, 0, :c] = w.transpose(2, 3, 1, 0).reshape(4 * c, c)
            put(h.deconv1.name + '/W', wn)
            put(h.deconv_b, arrays['head/deconv1/b'])
            loaded += ['head/deconv1/W', 'head/deconv1/b']
        return loaded


def save_npz(path, model):
    np.savez(path, **ChainerNpzMap(model).to_chainer())


def load_npz(path, model, strict=False):
    z = np.load(path)
    return ChainerNpzMap(model).from_chainer({k: z[k] for k in z.files}, strict=strict)


def load_resnet50_npz(path, model, strict=False):
    """ImageNet initialisation of the bottom-up pathway: a snapshot of ``chainer.links.ResNet50Layers`` (the file Chainer's
    ``ResNet50Layers('auto')`` loads for the reference's extractor, feature_pyramid_network.py:22 - keys ``conv1/W``,
    ``bn1/gamma``, ``res2/a/conv1/W`` ... ``res5/b2/bn3/avg_var``, ``fc6/W``) goes under ``extractor/resnet/``; ``fc6`` is
    dropped like the reference does (:23).  Returns the list of keys loaded (with the prefix)."""
    z = np.load(path)
    arrays = {'extractor/resnet/' + k: z[k] for k in z.files if not k.startswith('fc6/')}
    return ChainerNpzMap(model).from_chainer(arrays, strict=strict)
