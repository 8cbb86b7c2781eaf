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
    import __graft_entry__
    __graft_entry__.build()            # the driver's "does it build" entry (make is a no-op when the library is current)
    lib = _hip.lib()
    names = _declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), 'libmrcnn_hip.so lacks %s' % n
        assert n in _hip.SIGNATURES, 'ctypes binding lacks %s' % n
    assert set(_hip.SIGNATURES) == set(names)
    assert lib.mrcnn_abi_version() == _hip.ABI_VERSION == 8


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


def test_host_side_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers on the CPU build): the library's host code - argument validation, workspace and
    launch planning - built with AddressSanitizer (device code as usual; GPU ASan does not exist on the target pool) and
    driven by tests/native/abi_host_check.cpp: every planning query over a shape sweep, every compute entry point with
    null buffers / bad sizes.  No GPU is touched: each call must fail cleanly before launching."""
    import shutil
    import subprocess
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    import torch
    if torch.cuda.device_count() > 0:       # (does not initialise the device)
        pytest.skip('host-only check: the driver passes null buffers on purpose and must not run where a GPU could execute a launch')
    csrc = os.path.join(ROOT, 'chainer-maskrcnn_amd', 'csrc')
    subprocess.check_call(['make', '-C', csrc, 'asan', '-j8'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rt = subprocess.check_output(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so']).decode().strip()
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0',
               LD_LIBRARY_PATH=os.path.dirname(rt) + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
    r = subprocess.run([os.path.join(csrc, 'asan', 'abi_host_check')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and 'AddressSanitizer' not in out and ' 0 failure(s)' in out, out[-3000:]


def test_gemm_arithmetic_names_are_the_same_everywhere():
    """The training arithmetic is chosen by name in three places (train.py --gemm-arithmetic, bench.py --gemm-arithmetic, the chain's
    GEMM_ARITHMETIC table): the names and the default must agree, the float32-MFMA fallback must exist, and only the float32-ACCURATE
    schemes (library split modes 0 and 3) may be selectable there - the narrower two-plane splits (modes 1, 2) are never a training default."""
    import re
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import GEMM_ARITHMETIC, DEFAULT_GEMM_ARITHMETIC
    assert DEFAULT_GEMM_ARITHMETIC in GEMM_ARITHMETIC and 'f32' in GEMM_ARITHMETIC
    assert GEMM_ARITHMETIC['f32'][0] == (0, 0, 0)
    assert all(set(modes) <= {0, 3} for modes, _ in GEMM_ARITHMETIC.values())
    for path in ('train.py', 'bench.py'):
        src = open(os.path.join(ROOT, path)).read()
        m = re.search(r"'--gemm-arithmetic',\s*default=([^,]+),\s*choices=\[([^\]]+)\]", src)
        assert m, path
        choices = set(re.findall(r"'([^']+)'", m.group(2)))
        assert choices == set(GEMM_ARITHMETIC), (path, choices)
        default = m.group(1).strip().strip("'")
        assert default in ('None', DEFAULT_GEMM_ARITHMETIC), (path, default)      # bench.py: None = the shipped training default


def test_shipped_arithmetic_brackets_the_backbone_forward_calls():
    """'bf16x6_behind_backbone' = library split modes (3, 3, 3) with the FORWARD call of every extractor/resnet convolution bracketed back to
    the float32 MFMA by the host layer (nn/core.py: _layer_tiles).  The split-mode setting is host state of the library (no device needed):
    inside the bracket of a backbone layer the forward mode reads 0 and the backward modes stay 3; a layer behind the backbone keeps 3; the
    setting is restored on exit; 'f32' and 'bf16x6' do not bracket anything."""
    from chainer_maskrcnn._hip import nn as hnn
    from chainer_maskrcnn.nn import core
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import select_gemm_arithmetic

    class L(object):            # what _layer_tiles reads of a Conv
        fwd_tile = None

        def __init__(self, name):
            self.name = name
    backbone, behind = L('extractor/resnet/res3/a/conv2'), L('extractor/lat_p3')
    try:
        select_gemm_arithmetic('bf16x6_behind_backbone')
        assert tuple(hnn.split_operands()) == (3, 3, 3) and core.FWD_EMULATION_IN_BACKBONE is False
        with core._layer_tiles(backbone):
            assert tuple(hnn.split_operands()) == (0, 3, 3)
        assert tuple(hnn.split_operands()) == (3, 3, 3)
        with core._layer_tiles(behind):
            assert tuple(hnn.split_operands()) == (3, 3, 3)
        for name, want in (('bf16x6', (3, 3, 3)), ('bf16x6_backward', (0, 3, 3)), ('f32', (0, 0, 0))):
            select_gemm_arithmetic(name)
            assert core.FWD_EMULATION_IN_BACKBONE is True
            for layer in (backbone, behind):
                with core._layer_tiles(layer):
                    assert tuple(hnn.split_operands()) == want, (name, layer.name)
    finally:
        select_gemm_arithmetic('f32')
        core.FWD_EMULATION_IN_BACKBONE = core.FWD_EMULATION_BEHIND_BACKBONE = True
