"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU) and exports every
symbol include/mrcnn_hip.h declares; argument errors are reported without touching a device."""
import os
import re

import pytest

from chainer_maskrcnn import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'mrcnn_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mrcnn_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _hip.lib()
    names = _declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), 'libmrcnn_hip.so lacks %s' % n
        assert n in _hip.SIGNATURES, 'ctypes binding lacks %s' % n
    assert set(_hip.SIGNATURES) == set(names)
    assert lib.mrcnn_abi_version() == 3


def test_argument_errors_do_not_need_a_device():
    lib = _hip.lib()
    rc = lib.mrcnn_roi_align_fwd_f32(None, 7, 1, 8, 5, 5, None, 1, 7, 7, 0.25, 2, None, None)
    assert rc == -1
    assert b'layout' in lib.mrcnn_last_error()
    with pytest.raises(_hip.MrcnnHipError):
        _hip.check(rc)


def test_product_path_has_no_cpu_fallback():
    import torch
    from chainer_maskrcnn.functions.roi_align.roi_align_2d import roi_align_2d
    with pytest.raises(_hip.MrcnnHipError):
        roi_align_2d(torch.zeros(1, 4, 5, 5), torch.zeros(1, 5), 7, 7, 0.25)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'chainer-maskrcnn_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), fn
