"""Device-side pieces a user-written ``mask_loss_fun`` is made of.

The reference hands ``FPNMaskRCNNTrainChain`` a plain Python function (train.py:50-58,98;
train_keypoints.py:21-27) whose body uses ``chainer.functions.sigmoid_cross_entropy`` /
``softmax_cross_entropy``, NumPy-style fancy indexing ``roi_cls_mask[xp.arange(R), gt_roi_label - 1]``,
slicing and ``reshape``.  This module supplies the same pieces on HIP kernels so such a body runs
unchanged on device tensors:

  * ``sigmoid_cross_entropy(x, t)`` / ``softmax_cross_entropy(x, t)`` - Chainer's defaults
    (normalize=True, ignore label -1, mean reduction) as ``torch.autograd.Function``s over
    ``mrcnn_sigmoid_ce_f32`` / ``mrcnn_softmax_ce_f32`` (forward value and gradient in one pass, the
    upstream gradient applied by ``mrcnn_scale_by_dev_f32``);
  * ``MaskLogits`` - the tensor type the chain passes as ``roi_cls_mask``: an (R,C,H,W) NCHW tensor whose
    ``x[xp.arange(R), idx]`` runs ``mrcnn_select_channel_f32`` (forward and backward); every other index
    expression, slices and reshapes are torch views;
  * ``xp`` - the array-module argument: ``xp.arange(n)`` returns a lazy index object (no kernel).

There is no CPU fallback: the functions raise for host tensors.
"""
import torch

from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr
from chainer_maskrcnn._hip.nn import workspace


def _loss_ws(dev):
    return workspace(lib().mrcnn_loss_workspace_bytes(), dev)


def _scale_(t, g):
    """t *= g (device scalar)."""
    g = g.to(torch.float32).contiguous()
    check(lib().mrcnn_scale_by_dev_f32(ptr(t), t.numel(), ptr(g), stream_ptr()))
    return t


def _plain(t):
    return t.as_subclass(torch.Tensor) if type(t) is not torch.Tensor else t


class _SigmoidCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        _hip.require_cuda(x, t)
        x = x.contiguous()
        t = t.to(torch.int32).contiguous()
        if x.shape != t.shape:
            raise ValueError('sigmoid_cross_entropy: x %s and t %s must have the same shape' % (tuple(x.shape), tuple(t.shape)))
        out = torch.empty((2,), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        ws = _loss_ws(x.device)
        check(lib().mrcnn_sigmoid_ce_f32(ptr(x), ptr(t), x.numel(), ptr(out), ptr(gx), ptr(ws), ws.numel(), stream_ptr()))
        ctx.gx = gx
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        gx, ctx.gx = ctx.gx, None
        return _scale_(gx, g), None


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        _hip.require_cuda(x, t)
        if x.dim() != 2 or t.dim() != 1 or t.shape[0] != x.shape[0]:
            raise ValueError('softmax_cross_entropy: expected x (M,K) and t (M,)')
        x = x.contiguous()
        t = t.to(torch.int32).contiguous()
        M, K = x.shape
        out = torch.empty((2,), dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        ws = _loss_ws(x.device)
        check(lib().mrcnn_softmax_ce_f32(ptr(x), 1, K, 0, 1, ptr(t), M, K, -1, ptr(out), ptr(gx), K, 0, 1, 0, ptr(ws),
                                         ws.numel(), stream_ptr()))
        ctx.gx = gx
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        gx, ctx.gx = ctx.gx, None
        return _scale_(gx, g), None


def sigmoid_cross_entropy(x, t):
    """chainer.functions.sigmoid_cross_entropy(x, t) (train.py:57-58): mean over the elements with t != -1 of the
    binary cross entropy with logits."""
    return _SigmoidCE.apply(_plain(x), _plain(t))


def softmax_cross_entropy(x, t):
    """chainer.functions.softmax_cross_entropy(x (M,K), t (M,)) (train_keypoints.py:27): rows with t == -1 are ignored,
    the sum is divided by the number of valid rows."""
    return _SoftmaxCE.apply(_plain(x), _plain(t))


class _SelectChannel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        _hip.require_cuda(x, idx)
        x = x.contiguous()
        idx = idx.to(torch.int32).contiguous()
        R, C = x.shape[:2]
        HW = x[0, 0].numel()
        y = torch.empty((R,) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        check(lib().mrcnn_select_channel_f32(ptr(x), ptr(idx), R, C, HW, ptr(y), 0, stream_ptr()))
        ctx.idx, ctx.shape = idx, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        R, C = ctx.shape[:2]
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=gy.device)
        check(lib().mrcnn_select_channel_f32(ptr(gy.contiguous()), ptr(ctx.idx), R, C, gy[0].numel(), ptr(gx), 1, stream_ptr()))
        return gx, None


class ARange(object):
    """``xp.arange(n)``: a lazy 0..n-1 index.  ``MaskLogits`` recognises it; anything else gets a real tensor."""

    def __init__(self, n, device):
        self.n, self.device = int(n), device

    def __len__(self):
        return self.n

    def tensor(self):
        return torch.arange(self.n, device=self.device)


class XP(object):
    """The ``xp`` (array module) argument of ``mask_loss_fun(roi_cls_mask, gt_roi_mask, xp, gt_roi_label)``."""

    def __init__(self, device):
        self.device = device

    def arange(self, n):
        return ARange(n, self.device)


class MaskLogits(torch.Tensor):
    """(R, C, H, W) mask / keypoint logits as ``roi_cls_mask``: ``x[xp.arange(R), idx]`` picks channel idx[r] of row r
    (HIP select kernel, differentiable); other indexing falls through to torch (views for slices)."""

    def __getitem__(self, key):
        if isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], ARange) and torch.is_tensor(key[1]) and self.dim() >= 3:
            if key[0].n == self.shape[0] and key[1].shape == (self.shape[0],):
                return _SelectChannel.apply(_plain(self), _plain(key[1]))
        if isinstance(key, tuple):
            key = tuple(k.tensor() if isinstance(k, ARange) else k for k in key)
        elif isinstance(key, ARange):
            key = key.tensor()
        return _plain(self)[key]


class _NhwcToNchw(torch.autograd.Function):
    """(R,H,W,Cp) NHWC with Cp >= C padded channels -> (R,C,H,W) contiguous; backward pads the gradient back."""

    @staticmethod
    def forward(ctx, x, C):
        R, H, W, Cp = x.shape
        y = torch.empty((R, C, H, W), dtype=torch.float32, device=x.device)
        check(lib().mrcnn_nhwc_nchw_f32(ptr(x.contiguous()), ptr(y), R, H * W, Cp, C, 0, stream_ptr()))
        ctx.cp = Cp
        return y

    @staticmethod
    def backward(ctx, gy):
        R, C, H, W = gy.shape
        gx = torch.empty((R, H, W, ctx.cp), dtype=torch.float32, device=gy.device)
        check(lib().mrcnn_nhwc_nchw_f32(ptr(gy.contiguous()), ptr(gx), R, H * W, ctx.cp, C, 1, stream_ptr()))
        return gx, None


def nhwc_to_nchw(x, C):
    return _NhwcToNchw.apply(x, C)
