"""Parameter store and layer objects with hand-written backward passes.

Role in the mirror of the reference: this is what Chainer's ``Link``/``Variable`` graph does for
chainer_maskrcnn/model/* - here as an explicit tape: ``y, ctx = layer.fwd(x)`` /
``gx = layer.bwd(ctx, gy)``.  All arithmetic is in csrc/*.hip; tensors are NHWC fp32 device
buffers.

MI355X-first choices
  * ONE flat fp32 buffer for all parameters, one for gradients, one for momentum: the optimiser is
    a single fused kernel and the data-parallel all-reduce runs over contiguous slices of the
    gradient buffer, launched bucket by bucket as backward produces them (parameters are registered
    in forward order, so backward fills the buffer from the end towards the start).
  * Fan-out points of the graph accumulate in place through the ``accumulate`` flags of the
    backward kernels instead of separate add passes.
"""
import math

import numpy as np
import torch

from chainer_maskrcnn._hip import nn as hnn
from chainer_maskrcnn._hip import ops


# chainer.config.train: BatchNorm uses batch statistics (True) or the running averages (False, MaskRCNN.predict).
TRAIN = True

# Weight-gradient GEMMs on a second stream (joined by join_side_stream() at the end of backward).
FILTER_GRAD_ON_SIDE_STREAM = True
# Winograd layers: one read of gy for the data gradient, the filter-gradient operand and the bias gradient.  Saves ~2 GB of
# reads per step but the filter gradient then starts after the data gradient (less overlap): same-box A/B 26.80 vs 26.68
# ms/step (tools/ab_step.py), so it is off.
WINOGRAD_SHARED_GY_TRANSFORM = False
# Per-layer Winograd tile of the FORWARD pass.  Measured on the full network (profiles/r02_winograd_layer_probe.txt): the
# F(4x4,3x3) forward transform costs gradient accuracy ONLY in the ResNet conv2 layers (their 1e-4 output error passes
# through training-mode BatchNorm and moves c4 / c5, hence the lateral / RPN gradients, by up to 2e-2); in the layers
# behind the backbone - FPN smoothing convolutions, the RPN convolution, the box / mask / keypoint head convolutions -
# every gradient stays at the float32 noise floor.  Those layers are built with fwd_tile=0 ("F(4x4) where the map is large
# enough"); while the process-wide forward setting is the shipped 2 their forward AND backward calls are bracketed with
# that value (the library reads the setting on the host at call time, and the filter-gradient pass reuses the forward's
# transformed input only when both used the same tile).  LAYER_TILE_HINTS = False: every layer follows the process-wide
# setting.  FWD_TILE_RULE: callable(layer name) -> tile or None, overrides the hints (measurement, tools/wino_error_probe.py).
LAYER_TILE_HINTS = True
FWD_TILE_RULE = None
# BatchNorm statistics from the producing convolution's GEMM epilogue (mrcnn_conv2d_fwd_bnstats_f32) where the launch allows
FUSE_BN_STATS = True
# Measurement (tests/tools/seed_table.py): with the float32-accurate emulation selected for the FORWARD pass, False keeps the forward
# pass of the layers behind the backbone (every convolution outside extractor/resnet: FPN, RPN, heads - no BatchNorm behind them) on the
# float32 MFMA.
FWD_EMULATION_BEHIND_BACKBONE = True
# ... and False here keeps the forward pass of the BACKBONE's convolutions (extractor/resnet) on the float32 MFMA.
FWD_EMULATION_IN_BACKBONE = True
# One foreign call per ResNet bottleneck and pass (csrc/blocks.hip: mrcnn_bottleneck_fwd_f32 / _bwd_f32 enqueue the block's whole chain of
# launches, same kernels / operands / order / streams as the per-layer path below - the same bits) instead of ~25 calls and ~20
# allocations: the host side of the step (train.py:117-132) was 16.4 ms of Python per 21.7 ms step (profiles/r04_host_time.txt).
# False = the per-layer path (also taken under bench instrumentation, evaluation mode and the measurement knobs it has no notion of).
COMPOSITE_BLOCKS = True


class _layer_tiles(object):
    def __init__(self, conv):
        self.t = FWD_TILE_RULE(conv.name) if FWD_TILE_RULE is not None else (conv.fwd_tile if LAYER_TILE_HINTS else None)
        self.keep = None
        self.behind = not conv.in_backbone                 # FPN laterals / smoothing, RPN, heads: no BatchNorm behind them

    def __enter__(self):
        self.keep_split = None
        if (not FWD_EMULATION_BEHIND_BACKBONE and self.behind) or (not FWD_EMULATION_IN_BACKBONE and not self.behind):
            sp = hnn.split_operands()
            if sp[0] == 3:
                self.keep_split = sp
                from chainer_maskrcnn._hip import lib, check
                check(lib().mrcnn_conv2d_set_split_operands(0, sp[1], sp[2]))
        if self.t is None:
            return
        cur = hnn.winograd_pass_tiles()
        if cur[0] == 2 and cur[0] != self.t:
            self.keep = cur
            hnn.set_winograd_pass_tiles(self.t, cur[1], cur[2])

    def __exit__(self, *a):
        if self.keep is not None:
            hnn.set_winograd_pass_tiles(*self.keep)
        if self.keep_split is not None:
            from chainer_maskrcnn._hip import lib, check
            check(lib().mrcnn_conv2d_set_split_operands(*self.keep_split))


def join_side_stream(device):
    torch.cuda.current_stream(device).wait_stream(hnn.side_stream(device))


def pad_to(n, m):
    return (n + m - 1) // m * m


class ParamStore(object):
    """Registers named parameters, then materialises them as views of flat device buffers."""

    ALIGN = 64          # floats (256 B): every view is 16-B aligned for the float4 kernels

    def __init__(self):
        self._specs = []        # (name, shape, init(np.RandomState) -> ndarray, trainable)
        self.offsets = {}
        self.params = self.grads = self.momentum = None
        self.size = 0
        self.buffers = {}       # non-trainable state (BN running statistics)
        self._views = {}        # id(flat buffer) -> (flat buffer, {name: view})

    def register(self, name, shape, init, trainable=True):
        if name in self.offsets:
            raise KeyError('duplicate parameter %s' % name)
        n = int(np.prod(shape))
        if trainable:
            self.offsets[name] = (self.size, tuple(shape))
            self.size += pad_to(n, self.ALIGN)
        self._specs.append((name, tuple(shape), init, trainable))

    def materialise(self, device, seed=1234):
        rs = np.random.RandomState(seed)
        host = np.zeros((self.size,), np.float32)
        for name, shape, init, trainable in self._specs:
            v = np.asarray(init(rs), np.float32).reshape(shape)
            if trainable:
                o = self.offsets[name][0]
                host[o:o + v.size] = v.ravel()
            else:
                self.buffers[name] = torch.from_numpy(v.copy()).to(device)
        self.params = torch.from_numpy(host).to(device)
        self.grads = torch.zeros_like(self.params)
        self.momentum = torch.zeros_like(self.params)
        self._views = {}            # the views of earlier flat buffers (and the buffers they keep alive) go with them
        return self

    def _view(self, flat, name):
        # (views are cached per flat buffer: ~250 parameter / gradient lookups per step were 1.6 ms of slicing)
        cache = self._views.get(id(flat))
        if cache is None or cache[0] is not flat:
            # a flat buffer this store no longer owns (rebound params / grads, a device move) must not stay alive through its views
            self._views = {k: c for k, c in self._views.items() if c[0] is self.params or c[0] is self.grads or c[0] is self.momentum}
            cache = self._views[id(flat)] = (flat, {})
        v = cache[1].get(name)
        if v is None:
            o, shape = self.offsets[name]
            v = cache[1][name] = flat[o:o + int(np.prod(shape))].view(shape)
        return v

    def p(self, name):
        return self._view(self.params, name)

    def g(self, name):
        return self._view(self.grads, name)

    def names(self):
        return list(self.offsets)

    def n_params(self):
        return sum(int(np.prod(s)) for _, s in self.offsets.values())


def lecun_normal(fan_in):
    return lambda shape: (lambda rs: rs.standard_normal(shape) * (1.0 / math.sqrt(fan_in)))


def normal(std):
    return lambda shape: (lambda rs: rs.standard_normal(shape) * std)


class Conv(object):
    """Convolution2D / Linear / (as 1x1 to 4*Cout) Deconvolution2D on NHWC tensors.

    Logical channel counts (cin, cout) are zero-padded to what the MFMA kernel wants (cin: multiple
    of 32, or 4 for the image layer; cout: multiple of 32).  ``cout_map`` optionally places logical
    output channels at chosen padded positions (fused heads).
    """

    def __init__(self, ps, name, cin, cout, k=1, stride=1, pad=0, bias=True, relu=False, init=None,
                 cin_p=None, cout_p=None, cout_index=None, fwd_tile=None, in_backbone=None):
        self.ps, self.name, self.fwd_tile = ps, name, fwd_tile
        # a convolution whose output feeds a training-mode BatchNorm (every backbone's: FPN ResNet, C4, Darknet, the res5 head) keeps its
        # FORWARD pass on the float32 MFMA under 'bf16x6_behind_backbone'; set by the extractor that builds it (default: by the FPN's names)
        self.in_backbone = ('/resnet/' in name) if in_backbone is None else bool(in_backbone)
        self.cin, self.cout, self.k, self.stride, self.pad, self.relu = cin, cout, k, stride, pad, relu
        self.cin_p = cin_p or (4 if cin <= 4 else pad_to(cin, 32))
        self.cout_p = cout_p or pad_to(cout, 32)
        self.has_bias = bias
        idx = np.arange(cout) if cout_index is None else np.asarray(cout_index)
        fan_in = cin * k * k
        gen = (init or lecun_normal(fan_in))((cout, k, k, cin))

        def w_init(rs):
            w = np.zeros((self.cout_p, k, k, self.cin_p), np.float32)
            w[idx, :, :, :cin] = gen(rs)
            return w
        ps.register(name + '/W', (self.cout_p, k, k, self.cin_p), w_init)
        if bias:
            ps.register(name + '/b', (self.cout_p,), lambda rs: np.zeros((self.cout_p,), np.float32))

    @property
    def W(self):
        return self.ps.p(self.name + '/W')

    @property
    def b(self):
        return self.ps.p(self.name + '/b') if self.has_bias else None

    def fwd(self, x, relu=None, bn_stats=False):
        """bn_stats: the output feeds a training-mode BatchNorm - where the geometry allows, the GEMM epilogue also leaves the
        statistics partials in ``self.last_bn_part`` (None otherwise) and BatchNorm.fwd skips its statistics pass."""
        relu = self.relu if relu is None else relu
        self.last_bn_part = None
        hnn.LOGICAL = (self.cin, self.cout)
        if bn_stats and FUSE_BN_STATS and TRAIN and not relu and not self.has_bias:
            with _layer_tiles(self):
                res = hnn.conv2d_fwd_bnstats_raw(x, self.W, self.stride, self.pad, keep_v=True)
            if res is not None:
                hnn.LOGICAL = None
                y, v, self.last_bn_part = res
                return y, (x, None, v)
        # training: layers on the Winograd path keep their transformed input for the filter-gradient pass
        with _layer_tiles(self):
            y, v = hnn.conv2d_fwd_raw(x, self.W, self.b, self.stride, self.pad, relu, keep_v=True) if TRAIN else \
                (hnn.conv2d_fwd_raw(x, self.W, self.b, self.stride, self.pad, relu), None)
        hnn.LOGICAL = None
        return y, (x, y if relu else None, v)

    def bwd(self, ctx, gy, need_gx=True, gx_acc=None, accumulate_params=False, gy_masked=False, mask_gx=False):
        """gy: gradient w.r.t. the (post-ReLU) output.  gx_acc: tensor to accumulate gx into.
        accumulate_params: add into the parameter gradients (layer applied several times).
        gy_masked: the producer of gy already applied this layer's ReLU mask (the layer above ran with mask_gx).
        mask_gx: this layer's input is itself a ReLU output - zero gx where x <= 0 in the data-gradient epilogue,
        which is the ReLU backward of the layer below (call that layer with gy_masked=True)."""
        with _layer_tiles(self):
            return self._bwd(ctx, gy, need_gx, gx_acc, accumulate_params, gy_masked, mask_gx)

    def _bwd(self, ctx, gy, need_gx, gx_acc, accumulate_params, gy_masked, mask_gx):
        x, y, v = ctx
        if y is not None and not gy_masked:
            gy = ops.relu_bwd(gy, y)
        gw = self.ps.g(self.name + '/W')
        gb = self.ps.g(self.name + '/b') if self.has_bias else None
        hnn.LOGICAL = (self.cin, self.cout)
        try:
            if WINOGRAD_SHARED_GY_TRANSFORM and need_gx and self.stride == 1 and hnn.winograd_w_bytes(tuple(x.shape), tuple(gw.shape), 1, self.pad) > 0:
                # Winograd layer: the data-gradient call reads gy once for its own GEMM, the filter-gradient GEMM's
                # operand and the bias gradient; the filter gradient then runs on the side stream from the two kept
                # transforms (v from forward, wt from here) beside the next layer's backward
                assert not (mask_gx and gx_acc is not None)
                gx, wt = hnn.conv2d_bwd_data_raw(gy, self.W, tuple(x.shape), 1, self.pad, out=gx_acc, relu_x=x if mask_gx else None,
                                                 emit_w=True, gb=gb, gb_accumulate=accumulate_params)
                main = torch.cuda.current_stream(gy.device)
                side = hnn.side_stream(gy.device) if (FILTER_GRAD_ON_SIDE_STREAM and hnn.PROFILE is None) else main
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    hnn.conv2d_bwd_filter_raw(x, gy, tuple(gw.shape), 1, self.pad, False, gw=gw, gb=None,
                                              accumulate=accumulate_params, wino_v=v, wino_w=wt)
                if side is not main:
                    for t_ in (gy, x, v, wt):
                        if t_ is not None:
                            t_.record_stream(side)
                return gx
            if FILTER_GRAD_ON_SIDE_STREAM and hnn.PROFILE is None:
                main = torch.cuda.current_stream(gy.device)
                side = hnn.side_stream(gy.device)
                side.wait_stream(main)                      # gy (and x) are complete on the main stream
                with torch.cuda.stream(side):
                    hnn.conv2d_bwd_filter_raw(x, gy, tuple(gw.shape), self.stride, self.pad, self.has_bias, gw=gw, gb=gb,
                                              accumulate=accumulate_params, wino_v=v)
                gy.record_stream(side)                      # allocator: do not recycle gy before the side kernel is done
                x.record_stream(side)
                if v is not None:
                    v.record_stream(side)
            else:
                hnn.conv2d_bwd_filter_raw(x, gy, tuple(gw.shape), self.stride, self.pad, self.has_bias, gw=gw, gb=gb,
                                          accumulate=accumulate_params, wino_v=v)
            if not need_gx:
                return None
            if self.stride == 1:
                # (the epilogue masks the TOTAL: (old + product) where relu_x > 0 - direct, split-K and tail-split launches alike)
                assert not (mask_gx and gx_acc is not None) or self.k == 1
                return hnn.conv2d_bwd_data_raw(gy, self.W, tuple(x.shape), 1, self.pad, out=gx_acc,
                                               relu_x=x if mask_gx else None)
            assert self.k == 1 and self.pad == 0, 'strided backward-data only for 1x1 convolutions'
            g_sub = self.bwd_data_sub(gy)
        finally:
            hnn.LOGICAL = None
        return ops.subsample_bwd(g_sub, tuple(x.shape), self.stride, gx=gx_acc, relu_x=x if mask_gx else None)

    def bwd_data_sub(self, gy, out=None):
        """Data gradient of a strided 1x1 convolution on the subsampled lattice (N,Ho,Wo,Cin)."""
        N, Ho, Wo, _ = gy.shape
        keep = hnn.LOGICAL
        hnn.LOGICAL = (self.cin, self.cout)
        g = hnn.conv2d_bwd_data_raw(gy, self.W, (N, Ho, Wo, self.cin_p), 1, 0, out=out)
        hnn.LOGICAL = keep
        return g


class BatchNorm(object):
    """Training-mode BatchNormalization (batch statistics; eps 2e-5, decay 0.9 as in Chainer)."""

    def __init__(self, ps, name, c):
        self.ps, self.name, self.c = ps, name, c
        ps.register(name + '/gamma', (c,), lambda rs: np.ones((c,), np.float32))
        ps.register(name + '/beta', (c,), lambda rs: np.zeros((c,), np.float32))
        ps.register(name + '/avg_mean', (c,), lambda rs: np.zeros((c,), np.float32), trainable=False)
        ps.register(name + '/avg_var', (c,), lambda rs: np.ones((c,), np.float32), trainable=False)

    def fwd(self, x, relu=False, residual=None, partials=None):
        gamma, beta = self.ps.p(self.name + '/gamma'), self.ps.p(self.name + '/beta')
        if not TRAIN:
            y = ops.bn_infer_fwd(x, gamma, beta, self.ps.buffers[self.name + '/avg_mean'],
                                 self.ps.buffers[self.name + '/avg_var'], residual, relu)
            return y, None
        if partials is not None and partials.shape[2] == x.shape[-1]:
            y, mean, invstd = ops.bn_train_fwd_stats(x, partials, gamma, beta, residual, relu, self.ps.buffers[self.name + '/avg_mean'],
                                                     self.ps.buffers[self.name + '/avg_var'])
        else:
            y, mean, invstd = ops.bn_train_fwd(x, gamma, beta, residual, relu, self.ps.buffers[self.name + '/avg_mean'],
                                               self.ps.buffers[self.name + '/avg_var'])
        # BN + ReLU without a residual: backward recomputes the ReLU mask from x (bitwise the forward's y), so y is not
        # kept for (or read by) the backward pass
        return y, (x, y if (residual is not None or not relu) else None, mean, invstd, relu)

    def bwd(self, ctx, gy, want_gres=False, gy_masked=False):
        """gy_masked: the producer of gy already zeroed it where this layer's ReLU output is <= 0 (its data-gradient epilogue
        ran with mask_gx on the tensor this layer produced): no mask stream is read here and gres would be gy itself."""
        x, y, mean, invstd, relu = ctx
        if gy_masked:
            assert relu and not want_gres
            relu, y = False, None
        from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr
        C = self.c
        P = x.numel() // C
        gx = torch.empty_like(x)
        gres = torch.empty_like(x) if want_gres else None
        ws = hnn.workspace(lib().mrcnn_bn_workspace_bytes(P, C), x.device)
        check(lib().mrcnn_bn_train_bwd_f32(ptr(gy), ptr(x), ptr(y), ptr(self.ps.p(self.name + '/gamma')),
                                           ptr(self.ps.p(self.name + '/beta')), ptr(mean),
                                           ptr(invstd), ptr(gx), ptr(gres), ptr(self.ps.g(self.name + '/gamma')),
                                           ptr(self.ps.g(self.name + '/beta')), P, C, int(relu), ptr(ws), ws.numel(),
                                           stream_ptr()))
        return gx, gres


class Bottleneck(object):
    """ResNet bottleneck (Chainer BottleneckA when ``project`` else BottleneckB):
    relu(bn3(conv3(relu(bn2(conv2 3x3(relu(bn1(conv1 1x1/s))))))) + shortcut), stride on the first 1x1."""

    def __init__(self, ps, name, cin, mid, cout, stride, project):
        self.project, self.stride = project, stride
        he = lambda fan_in: (lambda shape: (lambda rs: rs.standard_normal(shape) * math.sqrt(2.0 / fan_in)))
        self.conv1 = Conv(ps, name + '/conv1', cin, mid, 1, stride, 0, bias=False, init=he(cin), in_backbone=True)
        self.bn1 = BatchNorm(ps, name + '/bn1', mid)
        self.conv2 = Conv(ps, name + '/conv2', mid, mid, 3, 1, 1, bias=False, init=he(mid * 9), in_backbone=True)
        self.bn2 = BatchNorm(ps, name + '/bn2', mid)
        self.conv3 = Conv(ps, name + '/conv3', mid, cout, 1, 1, 0, bias=False, init=he(mid), in_backbone=True)
        self.bn3 = BatchNorm(ps, name + '/bn3', cout)
        if project:
            self.conv4 = Conv(ps, name + '/conv4', cin, cout, 1, stride, 0, bias=False, init=he(cin), in_backbone=True)
            self.bn4 = BatchNorm(ps, name + '/bn4', cout)

    def _composite_ok(self, x):
        return (COMPOSITE_BLOCKS and TRAIN and FUSE_BN_STATS and x.is_cuda and hnn.PROFILE is None and FWD_TILE_RULE is None
                and not WINOGRAD_SHARED_GY_TRANSFORM)

    def _descriptor(self, x):
        """mrcnn_bottleneck_t for this block on input x (cached: the parameter / gradient pointers are views of the flat buffers)."""
        ps = self.conv1.ps
        N, H, W, _ = x.shape
        fwd_split = -1 if FWD_EMULATION_IN_BACKBONE else 0
        # (every tensor whose raw pointer goes into the descriptor is part of the key: the BatchNorm running statistics live outside the
        # flat buffers and may be rebound on their own - ADVICE r5)
        bufs = tuple(ps.buffers[b.name + sfx].data_ptr() for b in ([self.bn1, self.bn2, self.bn3] + ([self.bn4] if self.project else []))
                     for sfx in ('/avg_mean', '/avg_var'))
        key = (N, H, W, ps.params.data_ptr(), ps.grads.data_ptr(), fwd_split, bufs)
        if getattr(self, '_desc_key', None) != key:
            from chainer_maskrcnn import _hip
            d = _hip.Bottleneck()
            d.N, d.H, d.W = N, H, W
            d.cin, d.mid, d.cout = self.conv1.cin_p, self.conv1.cout_p, self.conv3.cout_p
            d.stride, d.project, d.fwd_split = self.stride, int(self.project), fwd_split
            d.eps, d.decay = 2e-5, 0.9
            layers = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)] + ([(self.conv4, self.bn4)] if self.project else [])
            for i, (c, b) in enumerate(layers):
                d.w[i], d.gw[i] = c.W.data_ptr(), ps.g(c.name + '/W').data_ptr()
                d.gamma[i], d.beta[i] = ps.p(b.name + '/gamma').data_ptr(), ps.p(b.name + '/beta').data_ptr()
                d.ggamma[i], d.gbeta[i] = ps.g(b.name + '/gamma').data_ptr(), ps.g(b.name + '/beta').data_ptr()
                d.run_mean[i], d.run_var[i] = ps.buffers[b.name + '/avg_mean'].data_ptr(), ps.buffers[b.name + '/avg_var'].data_ptr()
            self._desc, self._desc_key = d, key
        return self._desc

    def _fwd_composite(self, x):
        import ctypes
        from chainer_maskrcnn import _hip
        lib = _hip.lib()
        if not x.is_contiguous():
            raise ValueError('non-contiguous tensor passed to a HIP op')
        d = self._descriptor(x)
        plan = _hip.BottleneckPlan()
        _hip.check(lib.mrcnn_bottleneck_fwd_plan(ctypes.byref(d), ctypes.byref(plan)))
        N, H, W, _ = x.shape
        s = self.stride
        y = torch.empty((N, (H - 1) // s + 1, (W - 1) // s + 1, d.cout), dtype=torch.float32, device=x.device)
        arena = torch.empty((plan.arena_bytes,), dtype=torch.uint8, device=x.device)
        ws = hnn.workspace(plan.ws_bytes, x.device)
        _hip.check(lib.mrcnn_bottleneck_fwd_f32(ctypes.byref(d), ctypes.byref(plan), x.data_ptr(), y.data_ptr(), arena.data_ptr(), arena.numel(),
                                                ws.data_ptr(), ws.numel(), _hip.stream_ptr()))
        return y, ('composite', x, y, arena, plan)

    def _bwd_composite(self, ctx, gy, gx_acc, gy_masked, mask_gx):
        import ctypes
        from chainer_maskrcnn import _hip
        lib = _hip.lib()
        _, x, y, fwd_arena, plan = ctx
        if not gy.is_contiguous() or (gx_acc is not None and not gx_acc.is_contiguous()):
            raise ValueError('non-contiguous tensor passed to a HIP op')
        d = self._descriptor(x)
        dev = x.device
        sizes = (ctypes.c_size_t * 3)()
        _hip.check(lib.mrcnn_bottleneck_bwd_sizes(ctypes.byref(d), sizes))
        main = torch.cuda.current_stream(dev)
        side = hnn.side_stream(dev) if FILTER_GRAD_ON_SIDE_STREAM else main
        arena = torch.empty((max(int(sizes[0]), 1),), dtype=torch.uint8, device=dev)
        ws_main, ws_side = hnn.workspace(sizes[1], dev, main), hnn.workspace(sizes[2], dev, side)
        g_r = None if (gy_masked or self.project) else torch.empty_like(y)      # (a projection block never writes the masked gradient)
        gx_new = torch.empty_like(x) if (self.project and gx_acc is None) else None
        _hip.check(lib.mrcnn_bottleneck_bwd_f32(ctypes.byref(d), ctypes.byref(plan), x.data_ptr(), y.data_ptr(), fwd_arena.data_ptr(), gy.data_ptr(),
                                                int(bool(gy_masked)), _hip.ptr(g_r), _hip.ptr(gx_acc), _hip.ptr(gx_new), int(bool(mask_gx)),
                                                arena.data_ptr(), arena.numel(), ws_main.data_ptr(), ws_main.numel(), ws_side.data_ptr(),
                                                ws_side.numel(), main.cuda_stream, side.cuda_stream))
        if side is not main:            # the filter gradients read these on the side stream: the allocator must not recycle them before
            for t_ in (x, fwd_arena, arena):
                t_.record_stream(side)
        if gx_acc is not None:
            return gx_acc
        if self.project:
            return gx_new
        return gy if gy_masked else g_r          # identity shortcut: accumulated into the shortcut gradient in place

    def fwd(self, x):
        if self._composite_ok(x):
            return self._fwd_composite(x)
        h1, c1 = self.conv1.fwd(x, bn_stats=True)
        a1, b1 = self.bn1.fwd(h1, relu=True, partials=self.conv1.last_bn_part)
        h2, c2 = self.conv2.fwd(a1, bn_stats=True)
        a2, b2 = self.bn2.fwd(h2, relu=True, partials=self.conv2.last_bn_part)
        h3, c3 = self.conv3.fwd(a2, bn_stats=True)
        p3 = self.conv3.last_bn_part
        if self.project:
            h4, c4 = self.conv4.fwd(x, bn_stats=True)
            r, b4 = self.bn4.fwd(h4, partials=self.conv4.last_bn_part)
        else:
            r, c4, b4 = x, None, None
        y, b3 = self.bn3.fwd(h3, relu=True, residual=r, partials=p3)
        return y, (c1, b1, c2, b2, c3, b3, c4, b4)

    def bwd(self, ctx, gy, gx_acc=None, gy_masked=False, mask_gx=False):
        """gy_masked: gy already carries this block's output ReLU mask (its producer ran with mask_gx on this block's output);
        bn3's backward then reads two streams instead of three and writes no separate shortcut gradient (it IS gy).
        mask_gx: the block's input is itself a ReLU output (the previous block's): the returned gradient - the sum of all its
        contributions - is zeroed where that input is <= 0, in the epilogue of the kernel that writes it last."""
        if ctx[0] == 'composite':
            return self._bwd_composite(ctx, gy, gx_acc, gy_masked, mask_gx)
        c1, b1, c2, b2, c3, b3, c4, b4 = ctx
        if gy_masked:
            g_h3, _ = self.bn3.bwd(b3, gy, gy_masked=True)
            g_r = gy
        else:
            g_h3, g_r = self.bn3.bwd(b3, gy, want_gres=True)
        g_a2 = self.conv3.bwd(c3, g_h3)
        g_h2, _ = self.bn2.bwd(b2, g_a2)
        g_a1 = self.conv2.bwd(c2, g_h2)
        g_h1, _ = self.bn1.bwd(b1, g_a1)
        if not self.project:
            if gx_acc is not None:
                g_r = ops.add(g_r, gx_acc, out=gx_acc)
            return self.conv1.bwd(c1, g_h1, gx_acc=g_r, mask_gx=mask_gx)
        g_h4, _ = self.bn4.bwd(b4, g_r)
        if self.stride == 1:
            gx = self.conv1.bwd(c1, g_h1, gx_acc=gx_acc)
            return self.conv4.bwd(c4, g_h4, gx_acc=gx, mask_gx=mask_gx)
        # both strided 1x1 convolutions read the same lattice: sum on the lattice, scatter once
        self.conv1.bwd(c1, g_h1, need_gx=False)
        self.conv4.bwd(c4, g_h4, need_gx=False)
        g_sub = self.conv1.bwd_data_sub(g_h1)
        self.conv4.bwd_data_sub(g_h4, out=g_sub)
        return ops.subsample_bwd(g_sub, tuple(c1[0].shape), self.stride, gx=gx_acc, relu_x=c1[0] if mask_gx else None)
