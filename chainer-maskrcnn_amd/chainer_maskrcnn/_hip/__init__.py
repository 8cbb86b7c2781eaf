"""ctypes binding of libmrcnn_hip.so (include/mrcnn_hip.h).

This is the whole host<->device boundary: plain pointers and sizes, the caller's HIP
stream, integer return codes.  It is the stub a maintainer of the reference would add
 - here over torch tensors, which serve only as device-memory handles.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'csrc', 'libmrcnn_hip.so'))
# measurement only (tools/ab_lib.sh): another BUILD of the same library, for A/B runs of compile-time kernel variants on one box
if os.environ.get('MRCNN_HIP_LIB_AB'):
    LIB_PATH = os.path.abspath(os.environ['MRCNN_HIP_LIB_AB'])

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
_P = ctypes.POINTER

HEADER_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', '..', 'include', 'mrcnn_hip.h'))

# Host-array arguments (arrays of device pointers / per-level ints and floats) need typed pointers;
# every other pointer crosses the boundary as a plain address.
_MANUAL = {
    'mrcnn_last_error': (ctypes.c_char_p, []),
    'mrcnn_roi_align_fpn_fwd_f32': (c_int, [_P(c_void_p), _P(c_int), _P(c_int), _P(c_float), c_int,
                                            c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                            c_int, c_void_p, c_void_p]),
    'mrcnn_roi_align_fpn_fwd_ws_f32': (c_int, [_P(c_void_p), _P(c_int), _P(c_int), _P(c_float), c_int,
                                               c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                               c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'mrcnn_roi_align_fpn_bwd_f32': (c_int, [c_void_p, _P(c_void_p), _P(c_int), _P(c_int), _P(c_float),
                                            c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                            c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'mrcnn_rpn_pack_levels_f32': (c_int, [_P(c_void_p), _P(c_int), c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'mrcnn_rpn_unpack_grad_levels_f32': (c_int, [c_void_p, c_void_p, _P(c_void_p), _P(c_int), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'mrcnn_roi_align_fpn_bwd_workspace_bytes': (c_size_t, [_P(c_int), _P(c_int), c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
}
_CTYPE = {'int': c_int, 'float': c_float, 'size_t': c_size_t, 'long long': ctypes.c_longlong,
          'int32_t': ctypes.c_int32, 'unsigned': ctypes.c_uint, 'unsigned long long': ctypes.c_ulonglong}


def _parse_header(path):
    """name -> (restype, argtypes) for every function declared in include/mrcnn_hip.h."""
    import re
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    sigs = {}
    for m in re.finditer(r'\b(long long|int|size_t|const char \*)\s*(mrcnn_[a-z0-9_]+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        if name in _MANUAL:
            sigs[name] = _MANUAL[name]
            continue
        argtypes = []
        for a in [x.strip() for x in args.split(',')]:
            if a in ('void', ''):
                continue
            if '*' in a:
                argtypes.append(c_void_p)
            else:
                ty = ' '.join(a.replace('const ', '').split()[:-1])
                argtypes.append(_CTYPE[ty])
        sigs[name] = ({'int': c_int, 'size_t': c_size_t, 'long long': ctypes.c_longlong}[ret], argtypes)
    return sigs


# name -> (restype, argtypes): every symbol declared in include/mrcnn_hip.h
SIGNATURES = _parse_header(HEADER_PATH)

_lib = None
ABI_VERSION = 10         # MRCNN_ABI_VERSION of include/mrcnn_hip.h this binding was written against


class MrcnnHipError(RuntimeError):
    pass


def lib():
    """Load the HIP library once.  Fails loudly: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MrcnnHipError(
                'libmrcnn_hip.so not found at %s - build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or `make -C %s`'
                % (LIB_PATH, os.path.dirname(LIB_PATH)))
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if l.mrcnn_abi_version() != ABI_VERSION:
            raise MrcnnHipError('ABI version mismatch: library %d, binding %d' % (l.mrcnn_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().mrcnn_last_error().decode('utf-8', 'replace')
        raise MrcnnHipError('libmrcnn_hip call failed (code %d): %s' % (rc, msg))


_raw_stream = None


def raw_stream(device_index=None):
    """hipStream_t of torch's current stream on the (current) device as an int - the stream every launch of this layer goes to.  Through
    torch's C entry point (0.3 us) instead of building a torch.cuda.Stream object per call (7 us x ~200 calls per step)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)
    return _raw_stream(torch._C._cuda_getDevice() if device_index is None else device_index)


def stream_ptr():
    return ctypes.c_void_p(raw_stream())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MrcnnHipError('this op runs only on a HIP device (got a %s tensor); '
                                'there is no CPU fallback' % t.device)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


LAYOUT_NCHW, LAYOUT_NHWC = 0, 1


# ---- host structs of the composite entry points (include/mrcnn_hip.h, ABI v9) --------------------------------------------------------
_F4 = c_void_p * 4


class Bottleneck(ctypes.Structure):
    """mrcnn_bottleneck_t: geometry, parameter and gradient pointers of one ResNet bottleneck."""
    _fields_ = [('N', ctypes.c_int32), ('H', ctypes.c_int32), ('W', ctypes.c_int32), ('cin', ctypes.c_int32), ('mid', ctypes.c_int32),
                ('cout', ctypes.c_int32), ('stride', ctypes.c_int32), ('project', ctypes.c_int32), ('fwd_split', ctypes.c_int32),
                ('eps', c_float), ('decay', c_float), ('w', _F4), ('gamma', _F4), ('beta', _F4), ('run_mean', _F4), ('run_var', _F4),
                ('gw', _F4), ('ggamma', _F4), ('gbeta', _F4)]


BN_SLOTS = 23


class BottleneckPlan(ctypes.Structure):
    """mrcnn_bottleneck_plan_t: the forward arena's layout for the convolution settings in force when it was made."""
    _fields_ = [('arena_bytes', ctypes.c_uint64), ('ws_bytes', ctypes.c_uint64), ('off', ctypes.c_uint64 * BN_SLOTS),
                ('v_bytes', ctypes.c_uint64 * 4), ('part_rows', ctypes.c_int32 * 4)]
