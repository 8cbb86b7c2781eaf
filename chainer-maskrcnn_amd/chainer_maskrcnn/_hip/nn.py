"""Thin torch-tensor wrappers over the C ABI for the dense layers (NHWC, fp32).

Tensors are only device-memory handles; every FLOP runs in csrc/*.hip.  Functions ending in
``_raw`` are 1:1 calls; the ``torch.autograd.Function`` classes wire forward/backward pairs so
the model code can use autograd as the tape (the role Chainer's Variable graph plays in the
reference).
"""
import torch

from chainer_maskrcnn import _hip
from chainer_maskrcnn._hip import lib, check, ptr, stream_ptr

_ws_cache = {}

# bench.py instrumentation: when PROFILE is a list, every convolution launch is bracketed by HIP events on the
# launch stream and recorded as (kind, algorithmic MACs, start, end).  LOGICAL = (cin, cout) un-padded channel
# counts of the layer being run (set by nn.core.Conv), so padded channels do not count as work.
PROFILE = None
LOGICAL = None


class _prof(object):
    def __init__(self, kind, n_out_pix, KH, KW, cin, cout, geom=None):
        self.on = PROFILE is not None
        if self.on:
            pass_ = {'fwd': 0, 'bwd_data': 1, 'bwd_filter': 2}[kind]
            executed = lib().mrcnn_conv2d_executed_macs(*geom, pass_) if geom is not None else n_out_pix * KH * KW * cin * cout
            if LOGICAL is not None:
                cin, cout = LOGICAL
            self.rec = [kind, n_out_pix * KH * KW * cin * cout, torch.cuda.Event(enable_timing=True),
                        torch.cuda.Event(enable_timing=True), (n_out_pix, KH, cin, cout), executed, geom,
                        winograd_pass_tiles(),      # the per-pass tiles in force for THIS call (layers may bracket their own)
                        split_operands()]           # ... and the GEMM arithmetic per pass (the backbone's forward pass may differ)

    def __enter__(self):
        if self.on:
            self.rec[2].record()

    def __exit__(self, *a):
        if self.on:
            self.rec[3].record()
            PROFILE.append(tuple(self.rec))


_side = {}


def side_stream(device):
    """Second HIP stream per device: weight-gradient GEMMs run here, concurrently with the data-gradient chain on
    the main stream (they fill each other's tail waves; both only read gy)."""
    key = (device.type, device.index)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def workspace(nbytes, device, stream=None):
    """Grow-only scratch buffer per (device, stream - the current one unless given): the caller-owned workspace of the C ABI."""
    if device.type != 'cuda':
        sid = 0
    elif stream is None:
        sid = _hip.raw_stream(device.index)
    else:
        sid = stream.cuda_stream
    key = (device.type, device.index, sid)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def winograd_pass_tiles():
    """(forward, backward-data, backward-filter) Winograd tile settings of the process (mrcnn_conv2d_set_winograd_pass_tiles)."""
    import ctypes
    t = (ctypes.c_int * 3)()
    check(lib().mrcnn_conv2d_get_winograd_pass_tiles(t))
    return (t[0], t[1], t[2])


def split_operands():
    """(forward, backward-data, backward-filter) GEMM arithmetic of the process (mrcnn_conv2d_set_split_operands)."""
    import ctypes
    t = (ctypes.c_int * 3)()
    check(lib().mrcnn_conv2d_get_split_operands(t))
    return (t[0], t[1], t[2])


def set_winograd_pass_tiles(fwd, bwd_data, bwd_filter):
    check(lib().mrcnn_conv2d_set_winograd_pass_tiles(int(fwd), int(bwd_data), int(bwd_filter)))


def conv_out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


def conv2d_fwd_raw(x, w, b, stride, pad, relu, keep_v=False):
    """keep_v: returns (y, v) - v = the Winograd-transformed input of layers that take that path (else None), for the
    filter-gradient call of the same layer (conv2d_bwd_filter_raw(..., wino_v=v))."""
    _hip.require_cuda(x, w)
    N, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    assert x.is_contiguous() and w.is_contiguous()
    y = torch.empty((N, conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad), Cout),
                    dtype=torch.float32, device=x.device)
    nb = lib().mrcnn_conv2d_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = workspace(nb, x.device) if nb else None
    v = None
    if keep_v:
        vb = lib().mrcnn_conv2d_winograd_v_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
        if vb:
            v = torch.empty((vb // 4,), dtype=torch.float32, device=x.device)
    with _prof('fwd', N * y.shape[1] * y.shape[2], KH, KW, Cin, Cout, (N, H, W, Cin, Cout, KH, KW, stride, pad)):
        check(lib().mrcnn_conv2d_fwd_f32(ptr(x), ptr(w), ptr(b), ptr(y), N, H, W, Cin, Cout, KH, KW,
                                         stride, pad, int(relu), ptr(v), ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()))
    return (y, v) if keep_v else y


def conv2d_fwd_bnstats_raw(x, w, stride, pad, keep_v=False):
    """Forward convolution (no bias, no ReLU) that also produces the BatchNorm statistics partials of its output (GEMM
    epilogue, or the output transform of a Winograd layer): returns (y, v, part (rows, 2, Cout)) - or None when this geometry
    has no fused-statistics launch (split-K / tail split: mrcnn_conv2d_bnstats_rows == 0); the caller then takes the plain
    path.  v: the Winograd-transformed input when keep_v (as conv2d_fwd_raw)."""
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    rows = lib().mrcnn_conv2d_bnstats_rows(N, H, W, Cin, Cout, KH, KW, stride, pad)
    if rows == 0:
        return None
    assert x.is_contiguous() and w.is_contiguous()
    y = torch.empty((N, conv_out(H, KH, stride, pad), conv_out(W, KW, stride, pad), Cout), dtype=torch.float32, device=x.device)
    part = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
    nb = lib().mrcnn_conv2d_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = workspace(nb, x.device) if nb else None
    v = None
    if keep_v:
        vb = lib().mrcnn_conv2d_winograd_v_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
        if vb:
            v = torch.empty((vb // 4,), dtype=torch.float32, device=x.device)
    with _prof('fwd', N * y.shape[1] * y.shape[2], KH, KW, Cin, Cout, (N, H, W, Cin, Cout, KH, KW, stride, pad)):
        check(lib().mrcnn_conv2d_fwd_bnstats_f32(ptr(x), ptr(w), ptr(y), N, H, W, Cin, Cout, KH, KW, stride, pad, ptr(part), ptr(v),
                                                 ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()))
    return y, v, part


def winograd_w_bytes(x_shape, w_shape, stride, pad):
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = w_shape
    return lib().mrcnn_conv2d_winograd_w_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)


def conv2d_bwd_data_raw(gy, w, x_shape, stride, pad, out=None, relu_x=None, emit_w=False, gb=None, gb_accumulate=False):
    """out given => gx is accumulated into it (fan-out points of the graph), else allocated.
    relu_x = the layer's input when it came out of a ReLU: gx is masked with (relu_x > 0) in the epilogue.
    emit_w (Winograd layers only): returns (gx, wino_w) - the filter-gradient operand computed from the same read of gy
    (pass it to conv2d_bwd_filter_raw); gb: the bias gradient is written there from that read too."""
    N, H, W, Cin = x_shape
    Cout, KH, KW, _ = w.shape
    assert gy.is_contiguous()
    acc = out is not None
    gx = out if acc else torch.empty(x_shape, dtype=torch.float32, device=gy.device)
    nb = lib().mrcnn_conv2d_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = workspace(nb, gy.device) if nb else None
    wt = None
    if emit_w:
        wb = lib().mrcnn_conv2d_winograd_w_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
        assert wb > 0, 'emit_w: the layer does not take the Winograd path'
        wt = torch.empty((wb // 4,), dtype=torch.float32, device=gy.device)
    with _prof('bwd_data', N * gy.shape[1] * gy.shape[2], KH, KW, Cin, Cout, (N, H, W, Cin, Cout, KH, KW, stride, pad)):
        check(lib().mrcnn_conv2d_bwd_data_f32(ptr(gy), ptr(w), ptr(gx), ptr(relu_x), N, H, W, Cin, Cout, KH, KW,
                                              stride, pad, int(acc), ptr(wt), ptr(gb) if emit_w else None, int(gb_accumulate),
                                              ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()))
    return (gx, wt) if emit_w else gx


def conv2d_bwd_filter_raw(x, gy, w_shape, stride, pad, want_bias, gw=None, gb=None, accumulate=None, wino_v=None,
                          wino_w=None):
    """gw / gb given: written in place (accumulate=True adds - layers applied several times); else allocated.
    wino_v: the transformed input kept by conv2d_fwd_raw(..., keep_v=True) of the same layer."""
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_shape
    assert gy.is_contiguous() and x.is_contiguous()
    acc = (gw is not None) if accumulate is None else bool(accumulate)
    if gw is None:
        gw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
        gb = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    nbytes = lib().mrcnn_conv2d_bwd_filter_workspace_bytes(N, H, W, Cin, Cout, KH, KW, stride, pad)
    ws = workspace(nbytes, x.device)
    with _prof('bwd_filter', N * gy.shape[1] * gy.shape[2], KH, KW, Cin, Cout, (N, H, W, Cin, Cout, KH, KW, stride, pad)):
        check(lib().mrcnn_conv2d_bwd_filter_f32(ptr(x), ptr(gy), ptr(gw), ptr(gb) if want_bias else None, N, H, W, Cin,
                                                Cout, KH, KW, stride, pad, int(acc), ptr(wino_v), ptr(wino_w), ptr(ws), ws.numel(),
                                                stream_ptr()))
    return gw, gb
